// runtime.cpp — device selection, allocators and error reporting of the C-ABI (include/g4s.h, "runtime").
#include "common.hpp"
#include <algorithm>
#include "readback.hpp"
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

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

// Per-call arena (round 4). One SpGEMM call makes ≈ 150 small transient allocations (scan tile sums, sort counters, row lists, per-row tables); through the
// stream-ordered pool their hipFreeAsync calls alone cost 3–4.5 ms per call on this stack (one 16 MB free: 3.2 ms — profiles/r04_spgemm_host_phases.txt).
// Inside an ArenaScope, scratch_alloc hands out consecutive pieces of a few large chunks taken from the big-block cache and scratch_free is a no-op; when the
// outermost scope of the thread ends (the call's stream synchronised by then) the chunks go back to the cache, so from the second call on no driver
// allocation or free happens at all. Nothing is reused inside a call: the transients of the largest product add up to a few hundred MB.
namespace {
struct ArenaChunk { char *p; size_t bytes, used; };
struct Arena { std::vector<ArenaChunk> chunks; int depth = 0; };
thread_local Arena t_arena;
constexpr size_t kArenaChunk = (size_t)1 << 30, kArenaAlign = 256;   // (1 GiB since round 5: every small transient of a plan build or a product in ONE block that has been used before — the first host copy into a fresh 256 MB block measured 8.6 ms inside g4s_csr_create)
} // namespace

void arena_enter() { ++t_arena.depth; }
void arena_leave(hipStream_t s, bool idle)
{
    if (--t_arena.depth > 0) return;
    if (t_arena.chunks.empty()) return;
    if (!idle) (void)hipStreamSynchronize(s);                      // an error path left early: nothing may still read the chunks when they are handed out again
    for (auto &c : t_arena.chunks) (void)big_free(c.p, true);
    t_arena.chunks.clear();
}
static int arena_alloc(void **p, size_t bytes)
{
    bytes = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
    if (bytes == 0) bytes = kArenaAlign;
    for (auto &c : t_arena.chunks)
        if (c.bytes - c.used >= bytes) { *p = c.p + c.used; c.used += bytes; return G4S_OK; }
    void *q = nullptr;
    static const size_t chunk = [] { const char *e = getenv("G4S_ARENA_CHUNK_MB"); return e ? (size_t)atoll(e) << 20 : kArenaChunk; }();
    const size_t want = bytes > chunk ? bytes : chunk;
    G4S_TRY(big_alloc(&q, want));
    t_arena.chunks.push_back(ArenaChunk{static_cast<char *>(q), want, bytes});
    *p = q;
    return G4S_OK;
}
static bool arena_owns(const void *p)
{
    for (const auto &c : t_arena.chunks)
        if (p >= c.p && p < c.p + c.bytes) return true;
    return false;
}

int scratch_alloc(void **p, size_t bytes, hipStream_t s)
{
    if (t_arena.depth > 0) return arena_alloc(p, bytes);
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

static bool big_contains(const void *p);
void scratch_free(void *p, hipStream_t s)
{
    if (!p || arena_owns(p)) return;
    // A piece of an arena that is released AFTER its arena has ended, or on another thread, is an interior pointer of a big block (live or back in the cache), not
    // a pool allocation: handing it to hipFreeAsync fails and leaves a HIP error behind for the next hipGetLastError. Its memory went back with its chunk.
    if (big_contains(p)) return;
    if (hipFreeAsync(p, s) != hipSuccess) (void)hipGetLastError();
}

// Large device blocks (product outputs, per-row bitmaps: tens of MB to tens of GB). On this stack a fresh allocation of that size —
// hipMalloc and hipMallocFromPoolAsync alike — takes anything from under a millisecond to several seconds (measured: 0.5 ms vs
// 4.1 s for the same 15.5 GB request one call apart), so freed blocks are kept in a small free list and handed out again when a
// request fits one within 25 % (what a caching allocator does). big_free synchronises the device first, as hipFree would.
namespace {
struct BigBlock { void *p; size_t bytes; int device; };
std::mutex g_big_mutex;
std::unordered_map<void *, BigBlock> g_big_live;
std::vector<BigBlock> g_big_cache;
size_t g_big_cached_bytes = 0;
// Cached (freed, kept) blocks are bounded by count and by bytes: G4S_CACHE_MAX_BYTES, default 48 GiB of the 288 GB — enough for the
// outputs of the largest SpGEMM the int32 index type allows (cval 16 GB + ccol 8 GB + column scratch 8 GB) to be recycled call after call.
size_t big_cache_limit()
{
    static const size_t lim = [] { const char *e = getenv("G4S_CACHE_MAX_BYTES"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)48 << 30; }();
    return lim;
}
} // namespace
static bool big_contains(const void *p)
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    const char *c = static_cast<const char *>(p);
    for (const auto &kv : g_big_live)
        if (c >= static_cast<const char *>(kv.second.p) && c < static_cast<const char *>(kv.second.p) + kv.second.bytes) return true;
    for (const auto &b : g_big_cache)
        if (c >= static_cast<const char *>(b.p) && c < static_cast<const char *>(b.p) + b.bytes) return true;
    return false;
}
namespace {
int current_device()
{
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
// hipFree needs the owning device current only for the synchronisation it implies; the pointer itself identifies the allocation
void big_drop_cache_locked()
{
    for (auto &b : g_big_cache) (void)hipFree(b.p);
    g_big_cache.clear();
    g_big_cached_bytes = 0;
}
} // namespace

int big_alloc(void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lock(g_big_mutex);
        int best = -1;
        for (int i = 0; i < (int)g_big_cache.size(); ++i)
            if (g_big_cache[i].device == dev && g_big_cache[i].bytes >= bytes && g_big_cache[i].bytes - bytes <= bytes / 4 &&
                (best < 0 || g_big_cache[i].bytes < g_big_cache[best].bytes))
                best = i;
        if (best >= 0) {
            *p = g_big_cache[best].p;
            g_big_live[*p] = g_big_cache[best];
            g_big_cached_bytes -= g_big_cache[best].bytes;
            g_big_cache.erase(g_big_cache.begin() + best);
            return G4S_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {                                // give the cached blocks (and the SpGEMM column scratch) back and try once more
        (void)hipGetLastError();
        release_cached_device_memory();
        e = hipMalloc(p, bytes);
    }
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return set_error(G4S_ERR_NOMEM, "device allocation of %zu bytes: out of memory", bytes); }
    G4S_HIP_TRY(e);
    std::lock_guard<std::mutex> lock(g_big_mutex);
    g_big_live[*p] = BigBlock{*p, bytes, dev};
    return G4S_OK;
}

bool big_free(void *p, bool idle)
{
    if (!p) return true;
    BigBlock blk{};
    {
        std::lock_guard<std::mutex> lock(g_big_mutex);
        auto it = g_big_live.find(p);
        if (it == g_big_live.end()) return false;
        blk = it->second;
        g_big_live.erase(it);
    }
    // nothing in flight may still touch the block when it is handed out again: synchronise the device that OWNS it
    // (idle: the caller has already synchronised the one stream the block was used on — a device-wide synchronisation costs ≈ 0.2 ms per block on
    // this stack even when nothing runs, and a SpGEMM call releases a dozen blocks)
    if (!idle) {
        const int cur = current_device();
        if (cur != blk.device) (void)hipSetDevice(blk.device);
        (void)hipDeviceSynchronize();
        if (cur != blk.device) (void)hipSetDevice(cur);
    }
    std::lock_guard<std::mutex> lock(g_big_mutex);
    if (g_big_cache.size() >= 32 || g_big_cached_bytes + blk.bytes > big_cache_limit()) { (void)hipFree(p); return true; }
    g_big_cache.push_back(blk);
    g_big_cached_bytes += blk.bytes;
    return true;
}

void big_release_all()
{
    std::lock_guard<std::mutex> lock(g_big_mutex);
    big_drop_cache_locked();
}

void release_cached_device_memory()
{
    spgemm_release_cache();
    big_release_all();
}

// hipMalloc for the library's own plans and workspaces: on out-of-memory the caches above are dropped and the request repeated, so a
// plan build right after a large SpGEMM does not fail while tens of GB sit in the free list.
hipError_t device_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        release_cached_device_memory();
        e = hipMalloc(p, bytes ? bytes : 1);
    }
    return e;
}

int scratch_shutdown()
{
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (auto &pool : g_pools)
        if (pool) { G4S_HIP_TRY(hipMemPoolDestroy(pool)); pool = nullptr; }
    return G4S_OK;
}

// ---- small reads through pinned memory (readback.hpp)
namespace {
struct PendingRead { void *dst; size_t off, bytes; hipStream_t s; };
struct ReadBlock {
    static constexpr size_t kBytes = 4096;
    char *p = nullptr;                                             // pinned, portable; lives as long as the process (a thread's block is not returned: freeing pinned
    size_t used = 0;                                               // memory from a thread_local destructor can run behind the runtime's own shutdown)
    std::vector<PendingRead> pending;
};
thread_local ReadBlock t_reads;
void settle_reads(hipStream_t s, bool deliver)                     // the reads enqueued on s: handed out (or dropped); the block is reused once nothing is noted
{
    ReadBlock &b = t_reads;
    size_t keep = 0;
    for (const PendingRead &r : b.pending) {
        if (r.s != s) { b.pending[keep++] = r; continue; }
        if (deliver) std::memcpy(r.dst, b.p + r.off, r.bytes);
    }
    b.pending.resize(keep);
    if (!keep) b.used = 0;
}
} // namespace
hipError_t read_small(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    ReadBlock &b = t_reads;
    if (!b.p && hipHostMalloc(reinterpret_cast<void **>(&b.p), ReadBlock::kBytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; }
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (!b.p || b.used + need > ReadBlock::kBytes) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);   // (the plain way: correct, only slower)
    const hipError_t e = hipMemcpyAsync(b.p + b.used, src, bytes, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    b.pending.push_back(PendingRead{dst, b.used, bytes, s});
    b.used += need;
    return hipSuccess;
}
hipError_t reads_sync(hipStream_t s)
{
    const hipError_t e = hipStreamSynchronize(s);
    settle_reads(s, e == hipSuccess);
    return e;
}
hipError_t reads_sync_event(hipEvent_t ev, hipStream_t s)
{
    const hipError_t e = hipEventSynchronize(ev);
    settle_reads(s, e == hipSuccess);
    return e;
}
void reads_forget(const void *lo, const void *hi)
{
    ReadBlock &b = t_reads;
    size_t keep = 0;
    for (const PendingRead &r : b.pending)
        if (!(static_cast<const char *>(r.dst) >= static_cast<const char *>(lo) && static_cast<const char *>(r.dst) < static_cast<const char *>(hi))) b.pending[keep++] = r;
    b.pending.resize(keep);
    if (!keep) b.used = 0;
}

} // namespace g4s

G4S_API const char *g4s_version(void) { return "g4s-hip 0.1 (gfx950)"; }
#ifndef G4S_SPMV_KERNEL_HASH
#define G4S_SPMV_KERNEL_HASH "unknown"
#endif
#ifndef G4S_BUILD_VARIANT
#define G4S_BUILD_VARIANT ""
#endif
// What this library was built from, for tools that pair stored measurements with a build (bench.py's roofline.traffic): the SHA-256 of the SpMV kernel sources
// (tools/kernel_hash.py, taken by the Makefile when runtime.cpp is compiled) and the name of an A/B variant build (tools/build_variant.sh; empty for the regular one).
// g4s_warm_up (include/g4s.h): one small matrix through every SpMV path — host arrays, so that the upload path is loaded too
G4S_API g4s_status g4s_warm_up(void)
{
    const int n = 1 << 16;                                          // (past the single-workgroup sorts and scans of short lists: the build of a large matrix launches the general ones)
    // rows 64 … 191 hold 128 entries, rows 0 … 63 point 128 times at them (16 K products a row: the SpGEMM's long classes), rows 192 … 255 forty times (its
    // rank path from 4 096 products), rows 256 … 319 twelve times (the mid-size classes); every other row six scattered entries (the short-row kernels)
    std::vector<int32_t> rp(n + 1), ci;
    std::vector<double> va;
    ci.reserve((size_t)n * 6 + 40000); va.reserve((size_t)n * 6 + 40000);
    unsigned long long st = 88172645463325252ull;                   // xorshift: scattered columns, sorted inside a row
    std::vector<int32_t> c(128);
    for (int r = 0; r < n; ++r) {
        rp[r] = (int32_t)ci.size();
        const int per = r < 192 ? 128 : r < 256 ? 40 : r < 320 ? 12 : 6;
        const unsigned lo = r >= 64 && r < 192 ? 0u : r < 320 ? 64u : 0u, span = r >= 64 && r < 192 ? (unsigned)n : r < 320 ? 128u : (unsigned)n;
        for (int k = 0; k < per; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; c[k] = (int32_t)(lo + st % span); }
        std::sort(c.begin(), c.begin() + per);
        for (int k = 0; k < per; ++k) if (k == 0 || c[k] != c[k - 1]) { ci.push_back(c[k]); va.push_back(1.0); }
    }
    rp[n] = (int32_t)ci.size();
    double *x = nullptr, *y = nullptr;
    G4S_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&x), sizeof(double) * n));
    if (hipMalloc(reinterpret_cast<void **>(&y), sizeof(double) * n) != hipSuccess) { (void)hipFree(x); return g4s::set_error(G4S_ERR_HIP, "g4s_warm_up: hipMalloc failed"); }
    int rc = hipMemset(x, 0, sizeof(double) * n) == hipSuccess ? G4S_OK : g4s::set_error(G4S_ERR_HIP, "g4s_warm_up: hipMemset failed");
    for (unsigned flags : {0u, (unsigned)G4S_SPMV_BLOCKED, (unsigned)G4S_SPMV_STREAM}) {
        if (rc != G4S_OK) break;
        g4s_csr_t A = nullptr;
        rc = g4s_csr_create(&A, n, n, (int64_t)ci.size(), rp.data(), ci.data(), va.data(), G4S_HOST_POINTERS | flags);
        if (rc == G4S_OK) rc = g4s_spmv(A, x, y, 1.0, 0.0, nullptr);
        if (rc == G4S_OK && hipDeviceSynchronize() != hipSuccess) rc = g4s::set_error(G4S_ERR_HIP, "g4s_warm_up: synchronisation failed");
        (void)g4s_csr_destroy(A);
    }
    (void)hipFree(x); (void)hipFree(y);
    if (rc == G4S_OK) {                                             // … and the square of the same matrix through g4s_spgemm_csr_i32_f64: every row class of the product
        int32_t *crpt = nullptr, *ccol = nullptr;
        double *cval = nullptr;
        int64_t cnnz = 0;
        rc = g4s_spgemm_csr_i32_f64(rp.data(), ci.data(), va.data(), rp.data(), ci.data(), va.data(), &crpt, &ccol, &cval, n, n, n, &cnnz, nullptr, G4S_HOST_POINTERS | G4S_SORT_OUTPUT);
        g4s_free(crpt); g4s_free(ccol); g4s_free(cval);
    }
    return rc;
}

G4S_API const char *g4s_build_info(void) { return "spmv_kernel_sources_sha256=" G4S_SPMV_KERNEL_HASH ";variant=" G4S_BUILD_VARIANT; }
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

G4S_API g4s_status g4s_trim(void)
{
    g4s::release_cached_device_memory();
    return G4S_OK;
}

G4S_API g4s_status g4s_shutdown(void)
{
    g4s::release_cached_device_memory();
    return g4s::scratch_shutdown();
}

// Host allocator paired with every callee-allocated host output (mm/inc/utility.h:126-153 pairs my_malloc/my_free).
G4S_API void *g4s_malloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
G4S_API void g4s_free(void *p) { std::free(p); }

// Device allocations handed to the caller (g4s_dev_alloc, the callee-allocated outputs of g4s_spgemm_csr_i32_f64 with
// G4S_DEVICE_POINTERS) are blocks of the caching allocator above; g4s_dev_free returns them to it (and still accepts a plain
// hipMalloc pointer). g4s_shutdown releases what is cached.
G4S_API g4s_status g4s_dev_alloc(void **dptr, size_t bytes)
{
    G4S_REQUIRE(dptr, "dptr is NULL");
    return g4s::big_alloc(dptr, bytes);
}

G4S_API g4s_status g4s_dev_free(void *dptr)
{
    if (!dptr || g4s::big_free(dptr)) return G4S_OK;               // a block of the library's own allocator
    G4S_HIP_TRY(hipFree(dptr));                                    // a plain hipMalloc pointer
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
