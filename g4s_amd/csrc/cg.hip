// cg.hip — device-resident Jacobi-preconditioned conjugate gradient around the mat-vec.
//
// Follows conj_grad, citcoms/lib/General_matrix_functions.c:307-424, in its update order (z = BI∘r; β; p; Ap; α; d, r; residual)
// with global_vdot = plain dot product (single process: Global_operations.c:534-562 with no skipped halo equations) and
// assemble_del2_u(..., strip_bcs = 1) = mat-vec followed by zeroing the boundary equations (BC_util.c:89-102).
// The reference's own CUDA attempt copied p and Ap across PCIe every iteration and kept the dot products on the host
// (citcoms/lib/cgrad_kernel.cu:419-467); here every vector stays in HBM and each iteration is the mat-vec plus three fused
// vector kernels. Dot products are two-level (fixed grid → partials → serial sum in a fixed order): reproducible.
// Scalars (dot products, α, β) never leave the device; only the residual (8 bytes) goes to the host loop that owns the
// termination test of the source.
#include "common.hpp"
#include <algorithm>
#include <cmath>

// opaque handle types of the two operators, defined in graph.hip / spmv.hip
extern "C" g4s_status g4s_elem_op_apply(g4s_elem_op_t op, const double *u_dev, double *Au_dev, void *stream);
extern "C" g4s_status g4s_spmv(g4s_csr_t A, const double *x_dev, double *y_dev, double alpha, double beta, void *stream);

namespace {

constexpr int kDotBlocks = 256;   // partial sums per dot product
constexpr int kThreads = 256;

struct Scalars { double r1z1, r0z0; };

__device__ __forceinline__ double block_sum(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return s;
}

__device__ __forceinline__ double sum_partials(const double *__restrict__ part)
{
    // every thread adds the kDotBlocks partials in the same order: identical value everywhere, no broadcast needed
    double s = 0.0;
    for (int i = 0; i < kDotBlocks; ++i) s += part[i];
    return s;
}

// r1 = F; d0 = 0; partial r1·r1
__global__ __launch_bounds__(kThreads) void cg_init_kernel(int n, const double *__restrict__ F, double *__restrict__ r1, double *__restrict__ d0,
                                                            double *__restrict__ part)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kDotBlocks * kThreads) {
        const double f = F[i];
        r1[i] = f; d0[i] = 0.0;
        acc += f * f;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// z = BI∘r1; partial r1·z
__global__ __launch_bounds__(kThreads) void cg_precond_kernel(int n, const double *__restrict__ BI, const double *__restrict__ r1,
                                                               double *__restrict__ z, double *__restrict__ part)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kDotBlocks * kThreads) {
        const double zi = BI[i] * r1[i];
        z[i] = zi;
        acc += r1[i] * zi;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// dotr1z1 = Σ part; p2 = z (first) | z + (dotr1z1/dotr0z0)·p1; dotr0z0 := dotr1z1   (General_matrix_functions.c:365-379)
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(int n, int first, const double *__restrict__ part, Scalars *__restrict__ sc,
                                                                 const double *__restrict__ z, const double *__restrict__ p1, double *__restrict__ p2)
{
    const double r1z1 = sum_partials(part);
    const double beta = first ? 0.0 : r1z1 / sc->r0z0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) p2[i] = first ? z[i] : z[i] + beta * p1[i];
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) sc->r1z1 = r1z1;   // r0z0 is overwritten by cg_update_kernel, after every block has read it
}

// boundary rows of Ap := 0 (strip_bcs_from_residual); partial p2·Ap
__global__ __launch_bounds__(kThreads) void cg_strip_kernel(int n_zero, const int *__restrict__ zero_resid, double *__restrict__ v)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_zero) v[zero_resid[i]] = 0.0;
}

// boundary rows of Ap := 0 (strip_bcs_from_residual, through a byte mask built once per solve) fused with the partial p2·Ap
__global__ __launch_bounds__(kThreads) void cg_pAp_kernel(int n, const unsigned char *__restrict__ bc_mask, const double *__restrict__ p2,
                                                           double *__restrict__ Ap, double *__restrict__ part)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kDotBlocks * kThreads) {
        double a = Ap[i];
        if (bc_mask && bc_mask[i]) { a = 0.0; Ap[i] = 0.0; }
        acc += p2[i] * a;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void cg_mask_kernel(int n_zero, const int *__restrict__ zero_resid, unsigned char *__restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) mask[zero_resid[i]] = 1;
}

// alpha = dotprod == 0 ? 1e-3 : dotr1z1/dotprod; d0 += alpha·p2; r2 = r1 − alpha·Ap; partial r2·r2   (:383-394)
__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, const double *__restrict__ part_pAp, Scalars *__restrict__ sc,
                                                              const double *__restrict__ p2, const double *__restrict__ Ap,
                                                              const double *__restrict__ r1, double *__restrict__ r2, double *__restrict__ d0,
                                                              double *__restrict__ part_rr)
{
    __shared__ double sh[4];
    const double pAp = sum_partials(part_pAp);
    const double r1z1 = sc->r1z1;
    const double alpha = (pAp == 0.0) ? 1.0e-3 : r1z1 / pAp;
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kDotBlocks * kThreads) {
        d0[i] += alpha * p2[i];
        const double r = r1[i] - alpha * Ap[i];
        r2[i] = r;
        acc += r * r;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->r0z0 = r1z1;   // dotr0z0 := dotr1z1 for the next iteration's β (nobody reads r0z0 in this kernel)
}

__global__ __launch_bounds__(kThreads) void elem_inverse_diagonal_finish_kernel(int n, double *__restrict__ BI)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) BI[i] = BI[i] != 0.0 ? 1.0 / BI[i] : 0.0;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e != hipSuccess) return g4s::set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        return G4S_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

} // namespace

// defined in graph.hip (needs the operator's node→term map)
extern "C" g4s_status g4s_elem_op_diagonal_sum(g4s_elem_op_t op, double *diag_dev, void *stream);
int g4s_elem_op_neq(g4s_elem_op_t op);

G4S_API g4s_status g4s_elem_op_inverse_diagonal(g4s_elem_op_t op, double *BI_dev, void *stream)
{
    G4S_REQUIRE(op && BI_dev, "NULL argument");
    int neq = 0;
    G4S_TRY(g4s_elem_op_diagonal_sum(op, BI_dev, stream));
    neq = g4s_elem_op_neq(op);
    if (neq > 0) hipLaunchKernelGGL(elem_inverse_diagonal_finish_kernel, dim3((neq + kThreads - 1) / kThreads), dim3(kThreads), 0, g4s::as_stream(stream), neq, BI_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_conj_grad(g4s_elem_op_t op, g4s_csr_t A, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                                 const double *F, double *d0, double acc, int32_t *cycles, double *residual_out, void *stream)
{
    G4S_REQUIRE((op != nullptr) != (A != nullptr), "exactly one of op / A must be given");
    G4S_REQUIRE(neq > 0 && BI && F && d0 && cycles, "bad argument");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid), "zero_resid is NULL");
    hipStream_t s = g4s::as_stream(stream);
    const size_t nb = sizeof(double) * (size_t)neq;
    DevBuf r1b, r2b, zb, p1b, p2b, Apb, part, scal;
    G4S_TRY(r1b.alloc(nb)); G4S_TRY(r2b.alloc(nb)); G4S_TRY(zb.alloc(nb)); G4S_TRY(p1b.alloc(nb)); G4S_TRY(p2b.alloc(nb)); G4S_TRY(Apb.alloc(nb));
    G4S_TRY(part.alloc(sizeof(double) * 3 * kDotBlocks));
    G4S_TRY(scal.alloc(sizeof(Scalars)));
    double *r1 = r1b.as<double>(), *r2 = r2b.as<double>(), *z = zb.as<double>(), *p1 = p1b.as<double>(), *p2 = p2b.as<double>(), *Ap = Apb.as<double>();
    double *part_a = part.as<double>(), *part_b = part_a + kDotBlocks, *part_c = part_b + kDotBlocks;
    Scalars *sc = scal.as<Scalars>();
    G4S_HIP_TRY(hipMemsetAsync(sc, 0, sizeof(Scalars), s));

    DevBuf maskb;
    unsigned char *bc_mask = nullptr;
    if (n_zero) {
        G4S_TRY(maskb.alloc((size_t)neq));
        bc_mask = maskb.as<unsigned char>();
        G4S_HIP_TRY(hipMemsetAsync(bc_mask, 0, (size_t)neq, s));
        hipLaunchKernelGGL(cg_mask_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid, bc_mask);
    }
    // the residual: the kDotBlocks partial sums come to the host (2 KiB) and are added in block order — the order a device-side
    // finish kernel would use, without its launch
    double h_part[kDotBlocks];
    auto fetch_rr = [&](double *rr) -> int {
        G4S_HIP_TRY(hipMemcpyAsync(h_part, part_c, sizeof(h_part), hipMemcpyDeviceToHost, s));
        G4S_HIP_TRY(hipStreamSynchronize(s));
        double t = 0.0;
        for (int i = 0; i < kDotBlocks; ++i) t += h_part[i];
        *rr = t;
        return G4S_OK;
    };
    hipLaunchKernelGGL(cg_init_kernel, dim3(kDotBlocks), dim3(kThreads), 0, s, neq, F, r1, d0, part_c);
    double rr = 0.0;
    G4S_TRY(fetch_rr(&rr));
    double residual = std::sqrt(rr);
    const int steps = *cycles;
    int count = 0;
    while ((residual > acc && count < steps) || count == 0) {
        hipLaunchKernelGGL(cg_precond_kernel, dim3(kDotBlocks), dim3(kThreads), 0, s, neq, BI, r1, z, part_a);
        hipLaunchKernelGGL(cg_direction_kernel, dim3(kDotBlocks), dim3(kThreads), 0, s, neq, count == 0 ? 1 : 0, part_a, sc, z, p1, p2);
        if (op) G4S_TRY(g4s_elem_op_apply(op, p2, Ap, s));
        else G4S_TRY(g4s_spmv(A, p2, Ap, 1.0, 0.0, s));
        hipLaunchKernelGGL(cg_pAp_kernel, dim3(kDotBlocks), dim3(kThreads), 0, s, neq, bc_mask, p2, Ap, part_b);
        hipLaunchKernelGGL(cg_update_kernel, dim3(kDotBlocks), dim3(kThreads), 0, s, neq, part_b, sc, p2, Ap, r1, r2, d0, part_c);
        G4S_TRY(fetch_rr(&rr));
        residual = std::sqrt(rr);
        std::swap(r1, r2);      // the pointer rotation of General_matrix_functions.c:398-402
        std::swap(p1, p2);
        ++count;
    }
    *cycles = count;
    if (n_zero) hipLaunchKernelGGL(cg_strip_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid, d0);   // :409
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipStreamSynchronize(s));
    if (residual_out) *residual_out = residual;
    return G4S_OK;
}
