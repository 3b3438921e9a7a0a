// runtime.cpp — device selection, allocators and error reporting of the C-ABI (include/g4s.h, "runtime").
#include "common.hpp"
#include <mutex>

namespace g4s {

char *last_error_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Stream-ordered scratch from a library-private memory pool per device whose release threshold is "never": the default pool hands
// its pages back to the driver at the next synchronisation, so a solver that allocates scratch per call would pay a real
// allocation (hundreds of µs) every time — measured as 1.6 ms per outer Stokes iteration.
namespace {
constexpr int kMaxDevices = 16;
hipMemPool_t g_pools[kMaxDevices] = {};
std::mutex g_pool_mutex;
} // namespace

int scratch_alloc(void **p, size_t bytes, hipStream_t s)
{
    int dev = 0;
    G4S_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return set_error(G4S_ERR_INVALID, "device %d outside the scratch pool table", dev);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (!g_pools[dev]) {
            hipMemPoolProps props{};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            G4S_HIP_TRY(hipMemPoolCreate(&g_pools[dev], &props));
            uint64_t keep = UINT64_MAX;
            G4S_HIP_TRY(hipMemPoolSetAttribute(g_pools[dev], hipMemPoolAttrReleaseThreshold, &keep));
        }
    }
    hipError_t e = hipMallocFromPoolAsync(p, bytes ? bytes : 1, g_pools[dev], s);
    if (e == hipErrorOutOfMemory) return set_error(G4S_ERR_NOMEM, "scratch allocation of %zu bytes: out of memory", bytes);
    G4S_HIP_TRY(e);
    return G4S_OK;
}

void scratch_free(void *p, hipStream_t s)
{
    if (p) (void)hipFreeAsync(p, s);
}

int scratch_shutdown()
{
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &pool : g_pools)
        if (pool) { G4S_HIP_TRY(hipMemPoolDestroy(pool)); pool = nullptr; }
    return G4S_OK;
}

} // namespace g4s

G4S_API const char *g4s_version(void) { return "g4s-hip 0.1 (gfx950)"; }
G4S_API const char *g4s_last_error(void) { return g4s::last_error_buf(); }

G4S_API g4s_status g4s_device_count(int *count)
{
    G4S_REQUIRE(count, "count is NULL");
    *count = 0;
    G4S_HIP_TRY(hipGetDeviceCount(count));
    return G4S_OK;
}

G4S_API g4s_status g4s_set_device(int device)
{
    G4S_HIP_TRY(hipSetDevice(device));
    return G4S_OK;
}

G4S_API g4s_status g4s_device_synchronize(void)
{
    G4S_HIP_TRY(hipDeviceSynchronize());
    return G4S_OK;
}

G4S_API g4s_status g4s_shutdown(void) { return g4s::scratch_shutdown(); }

// Host allocator paired with every callee-allocated host output (mm/inc/utility.h:126-153 pairs my_malloc/my_free).
G4S_API void *g4s_malloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
G4S_API void g4s_free(void *p) { std::free(p); }

G4S_API g4s_status g4s_dev_alloc(void **dptr, size_t bytes)
{
    G4S_REQUIRE(dptr, "dptr is NULL");
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return g4s::set_error(G4S_ERR_NOMEM, "hipMalloc(%zu) out of memory", bytes);
    G4S_HIP_TRY(e);
    return G4S_OK;
}

G4S_API g4s_status g4s_dev_free(void *dptr)
{
    if (dptr) G4S_HIP_TRY(hipFree(dptr));
    return G4S_OK;
}

G4S_API g4s_status g4s_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes) G4S_HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return G4S_OK;
}

G4S_API g4s_status g4s_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes) G4S_HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return G4S_OK;
}
