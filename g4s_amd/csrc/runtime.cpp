// runtime.cpp — device selection, allocators and error reporting of the C-ABI (include/g4s.h, "runtime").
#include "common.hpp"

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

G4S_API g4s_status g4s_shutdown(void) { return G4S_OK; }

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
