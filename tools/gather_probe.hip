// gather_probe.hip — microbenchmark: cost of random 8-byte gathers from an 80 MB table under different cache policies.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o gpurun_out/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__host__ __device__ inline uint64_t mix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void fill_idx(int *idx, int64_t n, int table, int mode)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = mix64(i);
        if (mode == 0) idx[i] = (int)(h % (uint64_t)table);                       // uniform
        else { // power-law-ish: R-MAT column marginal, 24 bits with P(bit)=0.24, rejected >= table
            for (uint64_t a = 0;; ++a) { uint64_t c = 0; uint64_t w = 0; for (int l = 0; l < 24; ++l) { if ((l & 3) == 0) w = mix64(h + a * 77 + (l >> 2)); uint32_t r = w & 0xFFFF; w >>= 16; c = (c << 1) | (r < 15729 ? 1 : 0); } if (c < (uint64_t)table) { idx[i] = (int)c; break; } }
        }
    }
}

template <int POLICY, int UNROLL>
__global__ __launch_bounds__(256) void gather_kernel(const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, int64_t n)
{
    double acc = 0.0;
    const int64_t base = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    int c[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { int64_t k = base + u * 256; c[u] = __builtin_nontemporal_load(idx + (k < n ? k : n - 1)); }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        double v;
        if (POLICY == 0) v = x[c[u]];
        else if (POLICY == 1) v = __builtin_nontemporal_load(x + c[u]);
        else if (POLICY == 2) v = __hip_atomic_load(x + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (POLICY == 3) v = __hip_atomic_load(x + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else { float f = ((const float *)x)[2 * (int64_t)c[u]]; v = f; }
        acc += v;
    }
    if (acc == 123.456) out[0] = acc;
}

template <int POLICY>
void run(const char *name, const int *idx, const double *x, double *out, int64_t n)
{
    constexpr int U = 8;
    int grid = (int)((n + 256 * U - 1) / (256 * U));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather_kernel<POLICY, U>), dim3(grid), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gather_kernel<POLICY, U>), dim3(grid), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    printf("  %-28s %8.4f ms  %7.2f Ggather/s  (128B/line => %6.0f GB/s, 64B => %6.0f GB/s)\n", name, ms, n / ms / 1e6, n * 128.0 / ms / 1e6, n * 64.0 / ms / 1e6);
}

int main(int argc, char **argv)
{
    const int64_t n = 100000000;
    for (int table : {10000000, 1000000, 400000}) for (int mode : {0, 1}) {
        if (mode == 1 && table != 10000000) continue;
        int *idx; double *x, *out;
        CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, (size_t)table * 8)); CK(hipMalloc(&out, 8));
        CK(hipMemset(x, 0, (size_t)table * 8));
        hipLaunchKernelGGL(fill_idx, dim3(8192), dim3(256), 0, 0, idx, n, table, mode);
        CK(hipDeviceSynchronize());
        printf("table %d doubles (%.1f MB), %s indices, %ld gathers\n", table, table * 8 / 1e6, mode ? "rmat-marginal" : "uniform", (long)n);
        run<0>("plain", idx, x, out, n);
        run<1>("nt", idx, x, out, n);
        run<2>("agent-scope (sc1)", idx, x, out, n);
        run<3>("system-scope (sc0 sc1)", idx, x, out, n);
        run<4>("4-byte float load", idx, x, out, n);
        CK(hipFree(idx)); CK(hipFree(x)); CK(hipFree(out));
    }
    return 0;
}
