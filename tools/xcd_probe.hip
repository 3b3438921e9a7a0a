// xcd_probe.hip — which XCD does workgroup b of a 1-D launch run on? The XCD-aware walks of the SpMV kernels (csrc/spmv.hip: the diagonal path's plane-sliced
// walk, the CSR kernel's per-XCD row runs) assume "XCD = b mod 8" (round-robin dispatch). The kernel records HW_REG_XCC_ID per workgroup; the host prints how
// consistent that is for launches of the shapes those kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o gpurun_out/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void probe_kernel(int *xcc, int spin)
{
    if (threadIdx.x == 0) {
        // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 32)
        const unsigned v = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
        xcc[blockIdx.x] = (int)(v & 0xF);
    }
    // keep the workgroup alive for a while, as a real kernel would be: later workgroups are placed wherever a slot frees up
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
}

int main()
{
    for (int threads : {256, 1024}) {
        for (int blocks : {2048, 160000}) {
            for (int spin : {0, 20000}) {
                int *d = nullptr;
                CK(hipMalloc(&d, sizeof(int) * blocks));
                hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(threads), 0, 0, d, spin);
                CK(hipDeviceSynchronize());
                std::vector<int> h(blocks);
                CK(hipMemcpy(h.data(), d, sizeof(int) * blocks, hipMemcpyDeviceToHost));
                long long same = 0;
                int map8[8];
                for (int r = 0; r < 8; ++r) map8[r] = h[r];
                long long hist[8][8] = {};
                for (int b = 0; b < blocks; ++b) { same += h[b] == map8[b % 8]; hist[b % 8][h[b] & 7]++; }
                printf("threads %4d blocks %6d spin %5d: first 16 xcc =", threads, blocks, spin);
                for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
                printf(" | share of workgroups on the XCD of their residue class: %.4f\n", (double)same / blocks);
                if (same != blocks) {
                    for (int r = 0; r < 8; ++r) { printf("   b%%8=%d ->", r); for (int x = 0; x < 8; ++x) printf(" %6lld", hist[r][x]); printf("\n"); }
                }
                CK(hipFree(d));
            }
        }
    }
    return 0;
}
