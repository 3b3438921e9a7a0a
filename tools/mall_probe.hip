// mall_probe.hip — microbenchmarks behind the round-2 SpMV redesign:
//   A. is data that one kernel WROTE still on chip (Infinity Cache) when the next kernel reads it, and up to what size?
//   B. does re-writing the same 32 MB region cost less than writing fresh memory (would a write-back cache absorb it)?
//   C. does a streamed read of T bytes in between evict it, with plain and with nontemporal loads?
//   D. L2 -> LDS staging rate when every workgroup walks the tiles of a small table (1 / 3.5 / 8 MB)
//   E. LDS fp64 atomic-add rate on random addresses of an 8 K-entry tile
// Build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o gpurun_out/mall_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256) void write_kernel(double2_t *p, size_t n2, double v)
{
    const double2_t w = {v, v + 1.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(w, p + i); else p[i] = w;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void read_kernel(const double2_t *p, size_t n2, double *out)
{
    double acc = 0.0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n2; i += 4 * st) {
        double2_t a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + st); c = __builtin_nontemporal_load(p + i + 2 * st); d = __builtin_nontemporal_load(p + i + 3 * st); }
        else { a = p[i]; b = p[i + st]; c = p[i + 2 * st]; d = p[i + 3 * st]; }
        acc += a[0] + a[1] + b[0] + b[1] + c[0] + c[1] + d[0] + d[1];
    }
    for (; i < n2; i += st) { double2_t a = p[i]; acc += a[0] + a[1]; }
    if (acc == 123.456) out[0] = acc;
}

// D: every workgroup (1024 threads) stages 64 KiB tiles of a table into LDS, tile after tile, `rounds` tiles; start tile = blockIdx-dependent or 0
__global__ __launch_bounds__(1024) void stage_kernel(const double2_t *table, int ntiles, int rounds, int rotate, double *out)
{
    extern __shared__ double2_t lds[];
    double acc = 0.0;
    int t = rotate ? (int)(blockIdx.x % ntiles) : 0;
    for (int r = 0; r < rounds; ++r) {
        const double2_t *src = table + (size_t)t * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[k * 1024 + threadIdx.x] = src[k * 1024 + threadIdx.x];
        __syncthreads();
        acc += lds[(threadIdx.x * 37 + r) & 4095][0];
        __syncthreads();
        if (++t == ntiles) t = 0;
    }
    if (acc == 123.456) out[0] = acc;
}

// E: LDS fp64 atomic adds on random addresses
__global__ __launch_bounds__(1024) void lds_atomic_kernel(int iters, int span, double *out)
{
    extern __shared__ double ys[];
    for (int i = threadIdx.x; i < 8192; i += 1024) ys[i] = 0.0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned a = (s >> 10) % (unsigned)span;
        atomicAdd(&ys[a], 1.0);
    }
    __syncthreads();
    if (ys[threadIdx.x] == 123.456) out[0] = ys[threadIdx.x];
}

static hipEvent_t e0, e1;
template <typename F> float timed(F f, int reps = 1)
{
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t MB = 1 << 20;
    double2_t *buf, *big; double *out;
    CK(hipMalloc(&buf, 1024 * MB)); CK(hipMalloc(&big, 2048 * MB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, 1024 * MB)); CK(hipMemset(big, 0, 2048 * MB));
    const int G = 2048;
    auto W = [&](size_t bytes, int nt) { if (nt) hipLaunchKernelGGL(write_kernel<1>, dim3(G), dim3(256), 0, 0, buf, bytes / 16, 1.0); else hipLaunchKernelGGL(write_kernel<0>, dim3(G), dim3(256), 0, 0, buf, bytes / 16, 1.0); };
    auto R = [&](size_t bytes, int nt) { if (nt) hipLaunchKernelGGL(read_kernel<1>, dim3(G), dim3(256), 0, 0, buf, bytes / 16, out); else hipLaunchKernelGGL(read_kernel<0>, dim3(G), dim3(256), 0, 0, buf, bytes / 16, out); };
    auto FLUSH = [&](size_t bytes, int nt) { if (nt) hipLaunchKernelGGL(read_kernel<1>, dim3(G), dim3(256), 0, 0, big, bytes / 16, out); else hipLaunchKernelGGL(read_kernel<0>, dim3(G), dim3(256), 0, 0, big, bytes / 16, out); };
    // warm-up
    for (int i = 0; i < 3; ++i) { W(256 * MB, 0); R(256 * MB, 0); }
    CK(hipDeviceSynchronize());

    printf("A. write S then read S (GB/s of the read; 'cold' = after a 2 GB flush read)\n");
    for (size_t S : {16, 32, 64, 96, 128, 192, 256, 384, 768}) {
        float best_w = 1e9, best_r = 1e9, best_rr = 1e9, best_c = 1e9, best_wnt = 1e9, best_rnt = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            FLUSH(2048 * MB, 0);
            float w = timed([&] { W(S * MB, 0); });
            float r = timed([&] { R(S * MB, 0); });
            float rr = timed([&] { R(S * MB, 0); });
            FLUSH(2048 * MB, 0);
            float c = timed([&] { R(S * MB, 0); });
            FLUSH(2048 * MB, 0);
            float wnt = timed([&] { W(S * MB, 1); });
            float rnt = timed([&] { R(S * MB, 1); });
            if (w < best_w) best_w = w; if (r < best_r) best_r = r; if (rr < best_rr) best_rr = rr; if (c < best_c) best_c = c;
            if (wnt < best_wnt) best_wnt = wnt; if (rnt < best_rnt) best_rnt = rnt;
        }
        auto gbs = [&](float ms) { return S * MB / ms / 1e6; };
        printf("  S=%4zu MB: write %6.0f | read-after-write %6.0f | re-read %6.0f | cold read %6.0f | nt-write %6.0f | nt-read-after-nt-write %6.0f  GB/s\n", S, gbs(best_w), gbs(best_r),
               gbs(best_rr), gbs(best_c), gbs(best_wnt), gbs(best_rnt));
    }

    printf("B. 8 x write of the same 32 MB vs 1 x write of 256 MB (ms)\n");
    for (int nt = 0; nt < 2; ++nt) {
        float a = 1e9, b = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            FLUSH(2048 * MB, 0);
            float t1 = timed([&] { for (int i = 0; i < 8; ++i) W(32 * MB, nt); });
            FLUSH(2048 * MB, 0);
            float t2 = timed([&] { W(256 * MB, nt); });
            if (t1 < a) a = t1; if (t2 < b) b = t2;
        }
        printf("  %s stores: 8 x 32 MB %.4f ms (%.0f GB/s), 1 x 256 MB %.4f ms (%.0f GB/s)\n", nt ? "nt" : "plain", a, 256 * MB / a / 1e6, b, 256 * MB / b / 1e6);
    }

    printf("C. write 64 MB, stream-read T MB of another buffer, read the 64 MB back (GB/s of that read)\n");
    for (size_t T : {0, 64, 128, 256, 512, 1024}) for (int nt = 0; nt < 2; ++nt) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            FLUSH(2048 * MB, 0);
            W(64 * MB, 0);
            if (T) FLUSH(T * MB, nt);
            float r = timed([&] { R(64 * MB, 0); });
            if (r < best) best = r;
        }
        printf("  T=%4zu MB %-5s: %6.0f GB/s\n", T, nt ? "nt" : "plain", 64 * MB / best / 1e6);
    }

    printf("D. L2 -> LDS staging of 64 KiB tiles, 1024-thread workgroups, one per CU x 256 (aggregate GB/s)\n");
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int tiles : {16, 56, 128, 512}) for (int rotate = 0; rotate < 2; ++rotate) for (int wgs : {256, 512}) {
        const int rounds = 256;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            float t = timed([&] { hipLaunchKernelGGL(stage_kernel, dim3(wgs), dim3(1024), 65536, 0, buf, tiles, rounds, rotate, out); });
            if (t < best) best = t;
        }
        printf("  table %5.1f MB, %s, %d WGs: %.3f ms, %7.0f GB/s aggregate, %6.1f GB/s per CU\n", tiles * 65536 / 1e6, rotate ? "rotated start" : "lockstep     ", wgs, best,
               (double)wgs * rounds * 65536 / best / 1e6, (double)wgs * rounds * 65536 / best / 1e6 / 256);
    }

    printf("E. LDS fp64 atomic add, random addresses (Gatomic/s chip-wide)\n");
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_atomic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int span : {8192, 64, 1}) {
        const int iters = 2048;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            float t = timed([&] { hipLaunchKernelGGL(lds_atomic_kernel, dim3(256), dim3(1024), 65536, 0, iters, span, out); });
            if (t < best) best = t;
        }
        printf("  span %5d: %.3f ms, %7.1f Gatomic/s\n", span, best, 256.0 * 1024 * iters / best / 1e6);
    }
    return 0;
}
