// synth.hip — device generators of the synthetic inputs (include/g4s_synth.h). Twins of oracle/g4s_oracle.c's
// generators: counter-based hashing, so the integer outputs are bit-identical on CPU and GPU.
#include "common.hpp"
#include "g4s_synth.h"

namespace {

__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline double entry_value(uint64_t seed, int64_t i, int64_t j, int64_t n)
{
    const uint64_t h = mix64(seed ^ mix64((uint64_t)(i * n + j) + 0x5851F42D4C957F2Dull));
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

__global__ void rmat_keys_kernel(uint64_t seed, int scale, int64_t n, int64_t e0, int64_t count, int64_t *keys)
{
    const uint32_t TA = 37356u, TB = 49807u, TC = 62259u;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t e = (uint64_t)(e0 + q);
        for (uint64_t attempt = 0;; ++attempt) {
            const uint64_t base = mix64(seed ^ (e * 0x9E3779B97F4A7C15ull)) + attempt * 0xC2B2AE3D27D4EB4Full;
            int64_t row = 0, col = 0;
            uint64_t w = 0;
            for (int lvl = 0; lvl < scale; ++lvl) {
                if ((lvl & 3) == 0) w = mix64(base + (uint64_t)(lvl >> 2));
                const uint32_t r16 = (uint32_t)(w & 0xFFFFu);
                w >>= 16;
                const int rb = r16 >= TB, cb = (r16 >= TA && r16 < TB) || r16 >= TC;
                row = (row << 1) | rb;
                col = (col << 1) | cb;
            }
            if (row < n && col < n) { keys[q] = row * n + col; break; }
        }
    }
}

__global__ void csr_from_keys_kernel(uint64_t seed, int64_t n, int32_t rows, const int64_t *__restrict__ keys, int64_t nnz,
                                     int32_t *__restrict__ rowptr, int32_t *__restrict__ colids, double *__restrict__ values)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= nnz; k += (int64_t)gridDim.x * blockDim.x) {
        // rowptr[r] = first k whose row >= r: fill the gap between the previous entry's row and this one's.
        const int64_t row_prev = k == 0 ? -1 : keys[k - 1] / n;
        const int64_t row_cur = k == nnz ? (int64_t)rows : keys[k] / n;
        for (int64_t r = row_prev + 1; r <= row_cur; ++r) rowptr[r] = (int32_t)k;
        if (k < nnz) {
            const int64_t col = keys[k] - row_cur * n;
            colids[k] = (int32_t)col;
            values[k] = entry_value(seed, row_cur, col, n);
        }
    }
}

__global__ void vector_kernel(uint64_t seed, int64_t i0, int64_t count, double *x)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(seed + 0xD1B54A32D192ED03ull * (uint64_t)(i0 + i + 1));
        x[i] = (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

__global__ void laplacian_rows_kernel(int kind, int nx, int ny, int nz, int64_t r0, int64_t r1, int32_t *counts,
                                      const int32_t *__restrict__ rowptr, int32_t *__restrict__ colids,
                                      double *__restrict__ values, int fill)
{
    const int64_t pl = (int64_t)nx * ny;
    const double diag = kind == 5 ? 4.0 : 6.0;
    for (int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(r / pl), yy = (int)((r % pl) / nx), xx = (int)(r % nx);
        int64_t c[7];
        double v[7];
        int m = 0;
        if (z > 0) { c[m] = r - pl; v[m++] = -1.0; }
        if (yy > 0) { c[m] = r - nx; v[m++] = -1.0; }
        if (xx > 0) { c[m] = r - 1; v[m++] = -1.0; }
        c[m] = r; v[m++] = diag;
        if (xx < nx - 1) { c[m] = r + 1; v[m++] = -1.0; }
        if (yy < ny - 1) { c[m] = r + nx; v[m++] = -1.0; }
        if (z < nz - 1) { c[m] = r + pl; v[m++] = -1.0; }
        if (!fill) counts[r - r0] = m;
        else {
            const int32_t k = rowptr[r - r0];
            for (int t = 0; t < m; ++t) { colids[k + t] = (int32_t)c[t]; values[k + t] = v[t]; }
        }
    }
}

__global__ void banded_kernel(int n, int hb, uint64_t seed, int32_t *rowptr, int32_t *colids, double *values)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x) {
        // k(i) = Σ_{r<i} len(r), len(r) = min(n-1,r+hb) - max(0,r-hb) + 1
        auto start = [&](int64_t r) -> int64_t {
            int64_t tot = r * (2 * (int64_t)hb + 1);
            // rows r' < hb lose (hb - r') on the left
            int64_t a = r < hb ? r : hb;                     // rows 0..a-1 are clipped on the left
            tot -= a * hb - a * (a - 1) / 2;
            // rows r' > n-1-hb lose (r' + hb - (n-1)) on the right
            int64_t first = (int64_t)n - hb;                 // first clipped row index (may be < 0)
            if (first < 0) first = 0;
            if (r > first) {
                // Σ_{r'=first}^{r-1} (r' + hb - n + 1)
                int64_t cnt = r - first;
                tot -= cnt * (hb - n + 1) + (first + r - 1) * cnt / 2;
            }
            return tot;
        };
        const int64_t k = start(i);
        rowptr[i] = (int32_t)k;
        if (i < n) {
            const int64_t lo = i - hb < 0 ? 0 : i - hb, hi = i + hb > n - 1 ? n - 1 : i + hb;
            for (int64_t c = lo; c <= hi; ++c) {
                colids[k + (c - lo)] = (int32_t)c;
                values[k + (c - lo)] = entry_value(seed, i, c, n);
            }
        }
    }
}

inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

} // namespace

G4S_API g4s_status g4s_synth_rmat_keys(uint64_t seed, int32_t scale, int64_t n, int64_t e0, int64_t count,
                                       int64_t *keys_dev, void *stream)
{
    G4S_REQUIRE(scale > 0 && scale <= 31 && n > 0 && n <= ((int64_t)1 << scale) && count >= 0, "bad R-MAT parameters");
    if (count == 0) return G4S_OK;
    G4S_REQUIRE(keys_dev, "keys is NULL");
    hipLaunchKernelGGL(rmat_keys_kernel, dim3(grid_for(count)), dim3(256), 0, g4s::as_stream(stream), seed, scale, n, e0, count, keys_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_synth_csr_from_keys(uint64_t seed, int64_t n, int32_t rows, const int64_t *keys_dev, int64_t nnz,
                                           int32_t *rowptr_dev, int32_t *colids_dev, double *values_dev, void *stream)
{
    G4S_REQUIRE(rowptr_dev && rows >= 0 && nnz >= 0 && nnz <= INT32_MAX, "bad arguments");
    hipLaunchKernelGGL(csr_from_keys_kernel, dim3(grid_for(nnz + 1)), dim3(256), 0, g4s::as_stream(stream), seed, n, rows, keys_dev, nnz,
                       rowptr_dev, colids_dev, values_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_synth_vector(uint64_t seed, int64_t i0, int64_t count, double *x_dev, void *stream)
{
    if (count <= 0) return G4S_OK;
    G4S_REQUIRE(x_dev, "x is NULL");
    hipLaunchKernelGGL(vector_kernel, dim3(grid_for(count)), dim3(256), 0, g4s::as_stream(stream), seed, i0, count, x_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_synth_laplacian_rows(int32_t kind, int32_t nx, int32_t ny, int32_t nz, int64_t r0, int64_t r1,
                                            int32_t *counts_dev, const int32_t *rowptr_dev, int32_t *colids_dev,
                                            double *values_dev, int32_t fill, void *stream)
{
    G4S_REQUIRE((kind == 5 && nz == 1) || kind == 7, "kind must be 5 (nz==1) or 7");
    G4S_REQUIRE(nx > 0 && ny > 0 && nz > 0 && r0 >= 0 && r1 >= r0 && r1 <= (int64_t)nx * ny * nz, "bad grid/row range");
    G4S_REQUIRE((int64_t)nx * ny * nz <= INT32_MAX, "grid exceeds int32 column ids");
    if (r1 == r0) return G4S_OK;
    G4S_REQUIRE(fill ? (rowptr_dev && colids_dev && values_dev) : (counts_dev != nullptr), "NULL output");
    hipLaunchKernelGGL(laplacian_rows_kernel, dim3(grid_for(r1 - r0)), dim3(256), 0, g4s::as_stream(stream), kind, nx, ny, nz, r0, r1,
                       counts_dev, rowptr_dev, colids_dev, values_dev, fill);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_synth_banded(int32_t n, int32_t hb, uint64_t seed, int32_t *rowptr_dev, int32_t *colids_dev,
                                    double *values_dev, void *stream)
{
    G4S_REQUIRE(n > 0 && hb >= 0 && hb < n, "bad banded parameters");
    G4S_REQUIRE(rowptr_dev && colids_dev && values_dev, "NULL output");
    hipLaunchKernelGGL(banded_kernel, dim3(grid_for((int64_t)n + 1)), dim3(256), 0, g4s::as_stream(stream), n, hb, seed, rowptr_dev,
                       colids_dev, values_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

// ---------------------------------------------------------------------------------------------- the library's primitives, testable on their own
#include "prims.hpp"
G4S_API g4s_status g4s_prim_exclusive_scan_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream)
{
    G4S_REQUIRE(n >= 0 && (n == 0 || (in_dev && out_dev)), "bad argument");
    return g4s::prims::exclusive_scan<int>(in_dev, out_dev, n, g4s::as_stream(stream));
}
G4S_API g4s_status g4s_prim_exclusive_scan_i64(const int64_t *in_dev, int64_t *out_dev, int64_t n, void *stream)
{
    G4S_REQUIRE(n >= 0 && (n == 0 || (in_dev && out_dev)), "bad argument");
    return g4s::prims::exclusive_scan<long long>(reinterpret_cast<const long long *>(in_dev), reinterpret_cast<long long *>(out_dev), n, g4s::as_stream(stream));
}
G4S_API g4s_status g4s_prim_sort_pairs_desc_i32(const int32_t *keys_in_dev, const int32_t *vals_in_dev, int32_t *keys_out_dev, int32_t *vals_out_dev,
                                                int32_t n, int32_t key_bits, void *stream)
{
    G4S_REQUIRE(n >= 0 && (n == 0 || (keys_in_dev && vals_in_dev && keys_out_dev && vals_out_dev)), "bad argument");
    if (n == 0) return G4S_OK;
    hipStream_t s = g4s::as_stream(stream);
    void *tmp = nullptr;
    G4S_TRY(g4s::scratch_alloc(&tmp, sizeof(int) * 2 * (size_t)n, s));
    const int st = g4s::prims::sort_pairs_descending(keys_in_dev, vals_in_dev, keys_out_dev, vals_out_dev, static_cast<int *>(tmp), static_cast<int *>(tmp) + n, n, key_bits, s);
    g4s::scratch_free(tmp, s);
    return st;
}
