// dense_compare.hip — the reference's dense comparison drivers on the device:
//   mm/src/cblas_dxxmm.c:57-111   matrix_multiply_dsymm / _dtrmm / _dgemm   (column-major dim×dim, MKL CBLAS level 3)
//   mv/mv.c:6-27                  matrix_multiply_dsymv / _dtrmv / _sspmv (cblas_dspmv) / _dgemv   (MKL CBLAS level 2)
// They are what the reference times NEXT to its sparse kernels, not the optimisation target. Level 3 runs on the library's fp64 MFMA
// GEMM (graph.hip, g4s_dense_rows_times_matrix): a column-major product C = A·B is the row-major product Cᵀ = Bᵀ·Aᵀ on the same
// memory, so no transposition is ever made; the symmetric / triangular forms first expand the referenced triangle of A into a full
// matrix (one pass over A). Level 2 is HBM-bound: one pass over the matrix with unit-stride loads.
#include "common.hpp"

namespace {

// out (column-major, full) from the UPPER triangle of the column-major a: mode 0 = symmetric completion, 1 = upper triangle, rest zero
__global__ void dense_expand_upper_kernel(int n, int mode, const double *__restrict__ a, double *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const int i = (int)(idx % n), j = (int)(idx / n);               // element (row i, column j) at i + j·n
    out[idx] = i <= j ? a[idx] : (mode == 0 ? a[(long long)j + (long long)i * n] : 0.0);
}
// out (column-major, full symmetric) from the packed upper triangle ap: (i,j), i <= j, at i + j(j+1)/2
__global__ void dense_unpack_upper_kernel(int n, const double *__restrict__ ap, double *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    const long long i = idx % n, j = idx / n;
    out[idx] = i <= j ? ap[i + j * (j + 1) / 2] : ap[j + i * (i + 1) / 2];
}
// y = A·x, A column-major: 64 rows per workgroup (unit stride across lanes), the columns split over the 4 waves, LDS reduction
__global__ __launch_bounds__(256) void dense_gemv_n_kernel(int n, const double *__restrict__ a, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (i < n)
        for (int j = w; j < n; j += 4) s += a[(long long)i + (long long)j * n] * x[j];
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) y[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
// y = Aᵀ·x, A column-major: one wavefront per column (unit stride along the column), shuffle reduction
__global__ __launch_bounds__(256) void dense_gemv_t_kernel(int n, const double *__restrict__ a, const double *__restrict__ x, double *__restrict__ y)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += a[(long long)i + (long long)j * n] * x[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[j] = s;
}

struct Dev {
    double *p = nullptr;
    bool own = false;
    ~Dev() { if (own && p) (void)hipFree(p); }
};
// device view of a host or device array of n doubles (upload when host; allocate-only when src is NULL)
int dev_in(Dev &d, const double *src, size_t n, bool device_ptrs, bool copy)
{
    if (device_ptrs && src) { d.p = const_cast<double *>(src); d.own = false; return G4S_OK; }
    if (g4s::device_malloc((void **)&d.p, sizeof(double) * (n ? n : 1)) != hipSuccess) return g4s::set_error(G4S_ERR_NOMEM, "device allocation of %zu doubles failed", n);
    d.own = true;
    if (copy && src && n) G4S_HIP_TRY(hipMemcpy(d.p, src, sizeof(double) * n, hipMemcpyHostToDevice));
    return G4S_OK;
}
int dev_out(const Dev &d, double *dst, size_t n, bool device_ptrs)
{
    G4S_HIP_TRY(hipDeviceSynchronize());
    if (!device_ptrs && n) G4S_HIP_TRY(hipMemcpy(dst, d.p, sizeof(double) * n, hipMemcpyDeviceToHost));
    return G4S_OK;
}
inline int grid_nn(int n) { return (int)(((long long)n * n + 255) / 256); }

} // namespace

G4S_API g4s_status g4s_dense_mm(int32_t kind, int32_t dim, const double *A, double *B, double *C, unsigned flags)
{
    G4S_REQUIRE(dim >= 0, "negative dimension");
    G4S_REQUIRE(kind == G4S_DENSE_DGEMM || kind == G4S_DENSE_DSYMM || kind == G4S_DENSE_DTRMM, "unknown kind");
    if (dim == 0) return G4S_OK;
    G4S_REQUIRE(A && B && (C || kind == G4S_DENSE_DTRMM), "NULL argument");
    G4S_REQUIRE((long long)dim * dim < (1ll << 31), "dim too large");
    const bool dp = (flags & G4S_DEVICE_POINTERS) != 0;
    const size_t nn = (size_t)dim * dim;
    Dev a, b, c, t;
    G4S_TRY(dev_in(a, A, nn, dp, true));
    G4S_TRY(dev_in(b, B, nn, dp, true));
    if (kind == G4S_DENSE_DGEMM) {
        G4S_TRY(dev_in(c, C, nn, dp, false));
        G4S_TRY(g4s_dense_rows_times_matrix(dim, dim, dim, b.p, a.p, c.p, nullptr));          // Cᵀ = Bᵀ·Aᵀ
        return dev_out(c, C, nn, dp);
    }
    G4S_TRY(dev_in(t, nullptr, nn, false, false));
    hipLaunchKernelGGL(dense_expand_upper_kernel, dim3(grid_nn(dim)), dim3(256), 0, nullptr, dim, kind == G4S_DENSE_DSYMM ? 0 : 1, a.p, t.p);
    G4S_HIP_TRY(hipGetLastError());
    if (kind == G4S_DENSE_DSYMM) {                                                             // C = sym(A)·B  (Left, Upper)
        G4S_TRY(dev_in(c, C, nn, dp, false));
        G4S_TRY(g4s_dense_rows_times_matrix(dim, dim, dim, b.p, t.p, c.p, nullptr));
        return dev_out(c, C, nn, dp);
    }
    // dtrmm, Right / Upper / NoTrans / NonUnit: B := B·U, i.e. row-major (B·U)ᵀ = Uᵀ·Bᵀ = [U memory]·[B memory]
    Dev r;
    G4S_TRY(dev_in(r, nullptr, nn, false, false));
    G4S_TRY(g4s_dense_rows_times_matrix(dim, dim, dim, t.p, b.p, r.p, nullptr));
    if (dp) { G4S_HIP_TRY(hipMemcpyAsync(B, r.p, sizeof(double) * nn, hipMemcpyDeviceToDevice, nullptr)); G4S_HIP_TRY(hipDeviceSynchronize()); return G4S_OK; }
    return dev_out(r, B, nn, false);
}

G4S_API g4s_status g4s_dense_mv(int32_t kind, int32_t dim, const double *A, double *x, double *y, unsigned flags)
{
    G4S_REQUIRE(dim >= 0, "negative dimension");
    G4S_REQUIRE(kind == G4S_DENSE_DGEMV || kind == G4S_DENSE_DSYMV || kind == G4S_DENSE_DTRMV || kind == G4S_DENSE_DSPMV, "unknown kind");
    if (dim == 0) return G4S_OK;
    G4S_REQUIRE(A && x && (y || kind == G4S_DENSE_DTRMV), "NULL argument");
    G4S_REQUIRE((long long)dim * dim < (1ll << 31), "dim too large");
    const bool dp = (flags & G4S_DEVICE_POINTERS) != 0;
    const size_t nn = (size_t)dim * dim, na = kind == G4S_DENSE_DSPMV ? (size_t)dim * (dim + 1) / 2 : nn;
    Dev a, xv, yv, t;
    G4S_TRY(dev_in(a, A, na, dp, true));
    G4S_TRY(dev_in(xv, x, dim, dp, true));
    const double *m = a.p;
    if (kind != G4S_DENSE_DGEMV) {
        G4S_TRY(dev_in(t, nullptr, nn, false, false));
        if (kind == G4S_DENSE_DSPMV) hipLaunchKernelGGL(dense_unpack_upper_kernel, dim3(grid_nn(dim)), dim3(256), 0, nullptr, dim, a.p, t.p);
        else hipLaunchKernelGGL(dense_expand_upper_kernel, dim3(grid_nn(dim)), dim3(256), 0, nullptr, dim, kind == G4S_DENSE_DSYMV ? 0 : 1, a.p, t.p);
        G4S_HIP_TRY(hipGetLastError());
        m = t.p;
    }
    if (kind == G4S_DENSE_DTRMV) {                                                             // x := Uᵀ·x (Upper, Trans, NonUnit)
        Dev r;
        G4S_TRY(dev_in(r, nullptr, dim, false, false));
        hipLaunchKernelGGL(dense_gemv_t_kernel, dim3((dim + 3) / 4), dim3(256), 0, nullptr, dim, m, xv.p, r.p);
        G4S_HIP_TRY(hipGetLastError());
        if (dp) { G4S_HIP_TRY(hipMemcpyAsync(x, r.p, sizeof(double) * dim, hipMemcpyDeviceToDevice, nullptr)); G4S_HIP_TRY(hipDeviceSynchronize()); return G4S_OK; }
        return dev_out(r, x, dim, false);
    }
    G4S_TRY(dev_in(yv, y, dim, dp, false));
    hipLaunchKernelGGL(dense_gemv_n_kernel, dim3((dim + 63) / 64), dim3(256), 0, nullptr, dim, m, xv.p, yv.p);
    G4S_HIP_TRY(hipGetLastError());
    return dev_out(yv, y, dim, dp);
}
