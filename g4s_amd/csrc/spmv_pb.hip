// spmv_pb.hip — propagation-blocked fp64 SpMV for matrices whose x gathers have no locality (power-law graphs).
//
// Why: on R-MAT 10M/1e8 the row-streaming CSR kernel (spmv.hip) is bound by the x gather, not by the matrix stream — every
// 8-byte gather that misses L2 moves a 128-byte line (PMC: 6.1 GB fetched for 1.4 GB of algorithmic bytes; the gathers alone
// cost 0.89 ms, tools/gather_probe.hip). Blocking turns both random sides into streams:
//   producer  — nonzeros regrouped by 16K-column band. A workgroup loads that band of x into LDS (128 KiB) and streams
//               (local column u16, value f64) → product, which it writes to the slot the consumer will read it from.
//   consumer  — products regrouped by 16K-row band. A workgroup keeps that band of y in LDS, streams (product, local row u16)
//               and accumulates with LDS fp64 atomics, then writes the band of y once.
// HBM traffic is ≈28 B per nonzero (10 read + 8 written by the producer, 10 read by the consumer), all of it sequential,
// instead of ≈62 B per nonzero of 128-byte line fetches.
// Both orders are stable regroupings of the CSR order, so the cell (column band c, row band r) holds the same entries in the
// same order on both sides and one per-cell offset (delta[c][r]) maps a producer slot to its consumer slot.
// The regrouping is built once per matrix (g4s_csr_create) with rocPRIM radix sorts; the values are stored a second time in
// producer order. Sums are accumulated by LDS atomics: equal to the oracle within the fp64 tolerance, not bit for bit, and the
// last bits may differ from run to run.
#include "common.hpp"
#include "spmv_pb.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <memory>
#include <vector>

namespace g4s {

namespace {

#ifndef G4S_PB_BAND_BITS
#define G4S_PB_BAND_BITS 14
#endif
constexpr int kBandBits = G4S_PB_BAND_BITS;
constexpr int kBand = 1 << kBandBits;        // 16384 columns / rows per band: 128 KiB of fp64 in LDS
constexpr int kPbThreads = 1024;
constexpr int kProducerChunk = 1 << 17;      // entries per producer workgroup (x band load amortised over ≥ 2.3 MiB of stream)
constexpr int kConsumerChunk = 1 << 17;      // entries per consumer workgroup of a split (heavy) row band

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        bytes = n;
        return G4S_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- plan construction kernels
__global__ void pb_keys_kernel(int rows, long long nnz, const int *__restrict__ rowptr, const int *__restrict__ colids, int band_key_bits,
                               unsigned *__restrict__ keyP, unsigned *__restrict__ keyC, unsigned *__restrict__ idx)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        // row of entry k: last r with rowptr[r] <= k
        int lo = 0, hi = rows;
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (rowptr[mid] <= k) lo = mid; else hi = mid;
        }
        const unsigned rb = (unsigned)lo >> kBandBits, cb = (unsigned)colids[k] >> kBandBits;
        keyP[k] = (cb << band_key_bits) | rb;
        keyC[k] = (rb << band_key_bits) | cb;
        idx[k] = (unsigned)k;
    }
}

// Padded layout: every band segment starts at a multiple of 4 entries and every cell at a multiple of 2 (on BOTH sides), so an
// aligned pair of entries never straddles a cell and its consumer slot is 16-byte aligned: the producer moves pairs with 16-byte
// loads and stores, the consumer groups of 4. Entry i of the sorted order lives at i + shift[cell]. Pad slots: producer local
// column 0xFFFF (flag → product forced to 0) and value 0; consumer local row 0 and product 0 — harmless to the sums.
__global__ void pb_fill_producer_kernel(long long nnz, const int *__restrict__ colids, const double *__restrict__ values,
                                        const unsigned *__restrict__ permP, const unsigned *__restrict__ keyP_sorted, int band_key_bits,
                                        int nminor, const int *__restrict__ shiftP, unsigned short *__restrict__ p_lcol, double *__restrict__ p_val)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const unsigned k = permP[i];
        const unsigned key = keyP_sorted[i];
        const long long pos = i + shiftP[(long long)(key >> band_key_bits) * nminor + (key & ((1u << band_key_bits) - 1u))];
        p_lcol[pos] = (unsigned short)(colids[k] & (kBand - 1));
        p_val[pos] = values[k];
    }
}

__global__ void pb_fill_consumer_kernel(int rows, long long nnz, const int *__restrict__ rowptr, const unsigned *__restrict__ permC,
                                        const unsigned *__restrict__ keyC_sorted, int band_key_bits, int nminor, const int *__restrict__ shiftC,
                                        unsigned short *__restrict__ c_lrow)
{
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (long long)gridDim.x * blockDim.x) {
        const long long k = permC[j];
        int lo = 0, hi = rows;
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (rowptr[mid] <= k) lo = mid; else hi = mid;
        }
        const unsigned key = keyC_sorted[j];
        c_lrow[j + shiftC[(long long)(key >> band_key_bits) * nminor + (key & ((1u << band_key_bits) - 1u))]] = (unsigned short)(lo & (kBand - 1));
    }
}

// start[q] = first position whose sorted key >= key(q), for every cell q = major·nminor + minor, plus start[ncells] = nnz
__global__ void pb_cell_starts_kernel(long long nnz, const unsigned *__restrict__ sorted_keys, int nmajor, int nminor, int band_key_bits,
                                      int *__restrict__ start)
{
    const long long ncells = (long long)nmajor * nminor;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q <= ncells; q += (long long)gridDim.x * blockDim.x) {
        if (q == ncells) { start[q] = (int)nnz; continue; }
        const unsigned key = ((unsigned)(q / nminor) << band_key_bits) | (unsigned)(q % nminor);
        long long lo = 0, hi = nnz;
        while (lo < hi) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        start[q] = (int)lo;
    }
}

// ---- SpMV kernels
struct ProducerItem { int cband, k0, k1, pad; };
struct ConsumerItem { int rband, k0, k1, split; };

constexpr int kPbUnroll = 2;          // consumer: groups of 4 consecutive entries per thread per iteration
constexpr int kPbProducerUnroll = 4;  // producer: pairs of consecutive entries per thread per iteration
constexpr unsigned kPadFlag = 0x8000u;

typedef unsigned short ushort4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kPbThreads) void pb_producer_kernel(const ProducerItem *__restrict__ items, int cols, int RB,
                                                                  const unsigned short *__restrict__ p_lcol, const int *__restrict__ stP,
                                                                  const double *__restrict__ p_val, const int *__restrict__ delta,
                                                                  const double *__restrict__ x, double *__restrict__ prod)
{
    extern __shared__ double pb_lds[];
    double *xs = pb_lds;                                           // kBand doubles
    int *dl = reinterpret_cast<int *>(pb_lds + kBand);             // RB ints: this band's row of the delta table
    int *st = dl + RB;                                             // RB+1 ints: this band's cell starts (padded producer coordinates)
    const ProducerItem it = items[blockIdx.x];                     // [k0, k1): even bounds
    const int c0 = it.cband << kBandBits;
    const int last_pair = it.k1 - 2;
    constexpr int STEP = 2 * kPbThreads * kPbProducerUnroll;
    int base = it.k0 + 2 * (int)threadIdx.x;
    unsigned lc[kPbProducerUnroll], lc_n[kPbProducerUnroll];
    double2_t v[kPbProducerUnroll], v_n[kPbProducerUnroll];
    // first tile's stream loads go out before the x band is staged: their latency hides under the staging
#pragma unroll
    for (int u = 0; u < kPbProducerUnroll; ++u) {
        const int k = min(base + u * 2 * kPbThreads, last_pair);
        lc[u] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p_lcol + k));
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(p_val + k));
    }
    for (int i = threadIdx.x; i < kBand; i += kPbThreads) xs[i] = (c0 + i < cols) ? x[c0 + i] : 0.0;
    for (int r = threadIdx.x; r < RB; r += kPbThreads) dl[r] = delta[(long long)it.cband * RB + r];
    for (int r = threadIdx.x; r <= RB; r += kPbThreads) st[r] = stP[(long long)it.cband * (RB + 1) + r];
    __syncthreads();
    for (; base < it.k1; base += STEP) {
        const bool more = base + STEP < it.k1;
        if (more) {
#pragma unroll
            for (int u = 0; u < kPbProducerUnroll; ++u) {
                const int k = min(base + STEP + u * 2 * kPbThreads, last_pair);
                lc_n[u] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p_lcol + k));
                v_n[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(p_val + k));
            }
        }
#pragma unroll
        for (int u = 0; u < kPbProducerUnroll; ++u) {
            const int kk = base + u * 2 * kPbThreads;
            if (kk < it.k1) {
                int lo = 0, hi = RB;                               // last r with st[r] <= k (empty cells share a start: the last one wins)
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (st[mid] <= kk) lo = mid; else hi = mid;
                }
                const int dst = kk + dl[lo];
                const unsigned l0 = lc[u] & 0xFFFFu, l1 = lc[u] >> 16;
                double2_t pr;
                pr[0] = (l0 & kPadFlag) ? 0.0 : v[u][0] * xs[l0 & (kBand - 1)];
                pr[1] = (l1 & kPadFlag) ? 0.0 : v[u][1] * xs[l1 & (kBand - 1)];
                *reinterpret_cast<double2_t *>(prod + dst) = pr;
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < kPbProducerUnroll; ++u) { lc[u] = lc_n[u]; v[u] = v_n[u]; }
        }
    }
}

__global__ void pb_scale_rows_kernel(const int *__restrict__ split_bands, int rows, double *__restrict__ y, double beta)
{
    const int r0 = split_bands[blockIdx.y] << kBandBits;
    const int i = r0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows && i < r0 + kBand) y[i] = beta == 0.0 ? 0.0 : beta * y[i];
}

__global__ __launch_bounds__(kPbThreads) void pb_consumer_kernel(const ConsumerItem *__restrict__ items, int rows,
                                                                  const unsigned short *__restrict__ c_lrow, const double *__restrict__ prod,
                                                                  double *__restrict__ y, double alpha, double beta)
{
    extern __shared__ double pb_lds[];
    double *ys = pb_lds;
    const ConsumerItem it = items[blockIdx.x];                     // [k0, k1): multiples of 4; pad slots carry product 0 for row 0
    const int last_group = it.k1 - 4;
    constexpr int STEP = 4 * kPbThreads * kPbUnroll;
    int base = it.k0 + 4 * (int)threadIdx.x;
    ushort4_t lr[kPbUnroll], lr_n[kPbUnroll];
    double2_t pa[kPbUnroll], pb[kPbUnroll], pa_n[kPbUnroll], pb_n[kPbUnroll];
    if (it.k1 > it.k0) {
#pragma unroll
        for (int u = 0; u < kPbUnroll; ++u) {
            const int k = min(base + u * 4 * kPbThreads, last_group);
            lr[u] = __builtin_nontemporal_load(reinterpret_cast<const ushort4_t *>(c_lrow + k));
            pa[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(prod + k));
            pb[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(prod + k + 2));
        }
    }
    for (int i = threadIdx.x; i < kBand; i += kPbThreads) ys[i] = 0.0;
    __syncthreads();
    for (; base < it.k1; base += STEP) {
        const bool more = base + STEP < it.k1;
        if (more) {
#pragma unroll
            for (int u = 0; u < kPbUnroll; ++u) {
                const int k = min(base + STEP + u * 4 * kPbThreads, last_group);
                lr_n[u] = __builtin_nontemporal_load(reinterpret_cast<const ushort4_t *>(c_lrow + k));
                pa_n[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(prod + k));
                pb_n[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(prod + k + 2));
            }
        }
#pragma unroll
        for (int u = 0; u < kPbUnroll; ++u) {
            if (base + u * 4 * kPbThreads < it.k1) {
                const double p[4] = {pa[u][0], pa[u][1], pb[u][0], pb[u][1]};
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&ys[lr[u][j]], p[j]);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < kPbUnroll; ++u) { lr[u] = lr_n[u]; pa[u] = pa_n[u]; pb[u] = pb_n[u]; }
        }
    }
    __syncthreads();
    const int r0 = it.rband << kBandBits;
    for (int i = threadIdx.x; i < kBand; i += kPbThreads) {
        const int r = r0 + i;
        if (r >= rows) break;
        if (it.split) {
            if (ys[i] != 0.0) atomicAdd(&y[r], alpha * ys[i]);     // y was pre-scaled by beta (pb_scale_rows_kernel)
        } else {
            y[r] = beta == 0.0 ? alpha * ys[i] : alpha * ys[i] + beta * y[r];
        }
    }
}

inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

} // namespace

struct PbPlan {
    int rows = 0, cols = 0, CB = 0, RB = 0;
    long long nnz = 0;
    DevBuf p_lcol, p_val, c_lrow, prod, delta, stP, pitems, citems, split_bands;
    int n_pitems = 0, n_citems = 0, n_split = 0;
    size_t lds_producer = 0, lds_consumer = 0;
    long long bytes = 0;
};

int pb_build(PbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values)
{
    *out = nullptr;
    if (nnz <= 0 || rows <= 0 || cols <= 0) return set_error(G4S_ERR_INVALID, "pb_build: empty matrix");
    auto P = new (std::nothrow) PbPlan();
    if (!P) return set_error(G4S_ERR_NOMEM, "host allocation failed");
    std::unique_ptr<PbPlan> guard(P);
    P->rows = rows; P->cols = cols; P->nnz = nnz;
    P->CB = (cols + kBand - 1) >> kBandBits;
    P->RB = (rows + kBand - 1) >> kBandBits;
    int bits = 1;
    while ((1 << bits) < std::max(P->CB, P->RB)) ++bits;
    if (2 * bits > 32) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: too many bands");
    P->lds_producer = sizeof(double) * kBand + sizeof(int) * (2 * (size_t)P->RB + 1);
    P->lds_consumer = sizeof(double) * kBand;
    if (P->lds_producer > 160 * 1024) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: delta row does not fit LDS (rows > 134M)");

    DevBuf keyP, keyC, idx, keyP_s, keyC_s, permP, permC, tmp, startC;
    const size_t n4 = sizeof(unsigned) * (size_t)nnz;
    G4S_TRY(keyP.alloc(n4)); G4S_TRY(keyC.alloc(n4)); G4S_TRY(idx.alloc(n4));
    G4S_TRY(keyP_s.alloc(n4)); G4S_TRY(keyC_s.alloc(n4)); G4S_TRY(permP.alloc(n4)); G4S_TRY(permC.alloc(n4));
    hipLaunchKernelGGL(pb_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, rows, nnz, d_rowptr, d_colids, bits,
                       keyP.as<unsigned>(), keyC.as<unsigned>(), idx.as<unsigned>());
    G4S_HIP_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keyP.as<unsigned>(), keyP_s.as<unsigned>(), idx.as<unsigned>(),
                                                   permP.as<unsigned>(), (int)nnz, 0, 2 * bits, nullptr));
    G4S_TRY(tmp.alloc(tmp_bytes));
    G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keyP.as<unsigned>(), keyP_s.as<unsigned>(), idx.as<unsigned>(),
                                                   permP.as<unsigned>(), (int)nnz, 0, 2 * bits, nullptr));
    G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keyC.as<unsigned>(), keyC_s.as<unsigned>(), idx.as<unsigned>(),
                                                   permC.as<unsigned>(), (int)nnz, 0, 2 * bits, nullptr));
    const long long ncells = (long long)P->CB * P->RB;
    DevBuf startP;
    G4S_TRY(startP.alloc(sizeof(int) * (size_t)(ncells + 1)));
    G4S_TRY(startC.alloc(sizeof(int) * (size_t)(ncells + 1)));
    hipLaunchKernelGGL(pb_cell_starts_kernel, dim3(grid_for(ncells + 1)), dim3(256), 0, nullptr, nnz, keyP_s.as<unsigned>(), P->CB, P->RB, bits, startP.as<int>());
    hipLaunchKernelGGL(pb_cell_starts_kernel, dim3(grid_for(ncells + 1)), dim3(256), 0, nullptr, nnz, keyC_s.as<unsigned>(), P->RB, P->CB, bits, startC.as<int>());
    G4S_HIP_TRY(hipGetLastError());

    // cell boundaries (host) → padded layout: band segments at multiples of 4, cells at multiples of 2, on both sides
    std::vector<int> hP((size_t)ncells + 1), hC((size_t)ncells + 1);
    G4S_HIP_TRY(hipMemcpy(hP.data(), startP.p, sizeof(int) * hP.size(), hipMemcpyDeviceToHost));
    G4S_HIP_TRY(hipMemcpy(hC.data(), startC.p, sizeof(int) * hC.size(), hipMemcpyDeviceToHost));
    const int CB = P->CB, RB = P->RB;
    std::vector<int> padP((size_t)CB * (RB + 1)), padC((size_t)RB * (CB + 1)), shP((size_t)ncells), shC((size_t)ncells), h_delta((size_t)ncells);
    long long totP = 0, totC = 0;
    for (int c = 0; c < CB; ++c) {
        totP = (totP + 3) & ~3ll;
        for (int r = 0; r < RB; ++r) {
            const size_t q = (size_t)c * RB + r;
            padP[(size_t)c * (RB + 1) + r] = (int)totP;
            shP[q] = (int)(totP - hP[q]);
            totP += hP[q + 1] - hP[q];
            totP = (totP + 1) & ~1ll;
        }
        padP[(size_t)c * (RB + 1) + RB] = (int)totP;               // end of the band's last cell (its pad included)
    }
    for (int r = 0; r < RB; ++r) {
        totC = (totC + 3) & ~3ll;
        for (int c = 0; c < CB; ++c) {
            const size_t q = (size_t)r * CB + c;
            padC[(size_t)r * (CB + 1) + c] = (int)totC;
            shC[q] = (int)(totC - hC[q]);
            totC += hC[q + 1] - hC[q];
            totC = (totC + 1) & ~1ll;
        }
        padC[(size_t)r * (CB + 1) + CB] = (int)totC;
    }
    for (int c = 0; c < CB; ++c)
        for (int r = 0; r < RB; ++r) h_delta[(size_t)c * RB + r] = padC[(size_t)r * (CB + 1) + c] - padP[(size_t)c * (RB + 1) + r];
    totP = (totP + 3) & ~3ll; totC = (totC + 3) & ~3ll;
    if (totP + 8 > INT32_MAX || totC + 8 > INT32_MAX) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: padded length exceeds int32");
    DevBuf d_shP, d_shC;
    G4S_TRY(d_shP.alloc(sizeof(int) * shP.size())); G4S_TRY(d_shC.alloc(sizeof(int) * shC.size()));
    G4S_HIP_TRY(hipMemcpy(d_shP.p, shP.data(), sizeof(int) * shP.size(), hipMemcpyHostToDevice));
    G4S_HIP_TRY(hipMemcpy(d_shC.p, shC.data(), sizeof(int) * shC.size(), hipMemcpyHostToDevice));

    G4S_TRY(P->p_lcol.alloc(sizeof(unsigned short) * (size_t)(totP + 8)));
    G4S_TRY(P->p_val.alloc(sizeof(double) * (size_t)(totP + 8)));
    G4S_TRY(P->c_lrow.alloc(sizeof(unsigned short) * (size_t)(totC + 8)));
    G4S_TRY(P->prod.alloc(sizeof(double) * (size_t)(totC + 8)));
    G4S_HIP_TRY(hipMemset(P->p_lcol.p, 0xFF, P->p_lcol.bytes));   // pad flag (bit 15) everywhere; real entries overwrite it
    G4S_HIP_TRY(hipMemset(P->p_val.p, 0, P->p_val.bytes));
    G4S_HIP_TRY(hipMemset(P->c_lrow.p, 0, P->c_lrow.bytes));
    G4S_HIP_TRY(hipMemset(P->prod.p, 0, P->prod.bytes));
    hipLaunchKernelGGL(pb_fill_producer_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, nnz, d_colids, d_values, permP.as<unsigned>(),
                       keyP_s.as<unsigned>(), bits, RB, d_shP.as<int>(), P->p_lcol.as<unsigned short>(), P->p_val.as<double>());
    hipLaunchKernelGGL(pb_fill_consumer_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, rows, nnz, d_rowptr, permC.as<unsigned>(),
                       keyC_s.as<unsigned>(), bits, CB, d_shC.as<int>(), P->c_lrow.as<unsigned short>());
    G4S_HIP_TRY(hipGetLastError());
    G4S_TRY(P->stP.alloc(sizeof(int) * padP.size()));
    G4S_TRY(P->delta.alloc(sizeof(int) * h_delta.size()));
    G4S_HIP_TRY(hipMemcpy(P->stP.p, padP.data(), sizeof(int) * padP.size(), hipMemcpyHostToDevice));
    G4S_HIP_TRY(hipMemcpy(P->delta.p, h_delta.data(), sizeof(int) * h_delta.size(), hipMemcpyHostToDevice));

    const int kPC = (getenv("G4S_PB_PCHUNK") ? atoi(getenv("G4S_PB_PCHUNK")) : kProducerChunk) & ~3;   // tuning knobs (experiments)
    const int kCC = (getenv("G4S_PB_CCHUNK") ? atoi(getenv("G4S_PB_CCHUNK")) : kConsumerChunk) & ~3;
    std::vector<ProducerItem> pit;
    for (int c = 0; c < P->CB; ++c) {
        const int b0 = padP[(size_t)c * (RB + 1)], b1 = padP[(size_t)c * (RB + 1) + RB];                 // b0 multiple of 4, b1 even
        for (int k = b0; k < b1; k += kPC) pit.push_back(ProducerItem{c, k, std::min(b1, k + kPC), 0});
    }
    std::vector<ConsumerItem> cit;
    std::vector<int> split;
    for (int r = 0; r < P->RB; ++r) {
        const int b0 = padC[(size_t)r * (CB + 1)], b1 = (padC[(size_t)r * (CB + 1) + CB] + 3) & ~3;      // pads included: harmless
        if (b1 - b0 <= kCC) cit.push_back(ConsumerItem{r, b0, b1, 0});   // also the empty bands: their rows must still be written
        else {
            split.push_back(r);
            for (int k = b0; k < b1; k += kCC) cit.push_back(ConsumerItem{r, k, std::min(b1, k + kCC), 1});
        }
    }
    // heaviest items first: the tail of each launch is then made of light bands
    std::stable_sort(cit.begin(), cit.end(), [](const ConsumerItem &a, const ConsumerItem &b) { return (a.k1 - a.k0) > (b.k1 - b.k0); });
    std::stable_sort(pit.begin(), pit.end(), [](const ProducerItem &a, const ProducerItem &b) { return (a.k1 - a.k0) > (b.k1 - b.k0); });
    P->n_pitems = (int)pit.size(); P->n_citems = (int)cit.size(); P->n_split = (int)split.size();
    G4S_TRY(P->pitems.alloc(sizeof(ProducerItem) * pit.size()));
    G4S_TRY(P->citems.alloc(sizeof(ConsumerItem) * cit.size()));
    G4S_TRY(P->split_bands.alloc(sizeof(int) * split.size()));
    if (!pit.empty()) G4S_HIP_TRY(hipMemcpy(P->pitems.p, pit.data(), sizeof(ProducerItem) * pit.size(), hipMemcpyHostToDevice));
    if (!cit.empty()) G4S_HIP_TRY(hipMemcpy(P->citems.p, cit.data(), sizeof(ConsumerItem) * cit.size(), hipMemcpyHostToDevice));
    if (!split.empty()) G4S_HIP_TRY(hipMemcpy(P->split_bands.p, split.data(), sizeof(int) * split.size(), hipMemcpyHostToDevice));
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pb_producer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_producer));
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(pb_consumer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_consumer));
    G4S_HIP_TRY(hipDeviceSynchronize());
    P->bytes = (long long)(P->p_lcol.bytes + P->stP.bytes + P->p_val.bytes + P->c_lrow.bytes + P->prod.bytes + P->delta.bytes +
                           P->pitems.bytes + P->citems.bytes + P->split_bands.bytes);
    *out = guard.release();
    return G4S_OK;
}

void pb_destroy(PbPlan *P) { delete P; }

long long pb_bytes(const PbPlan *P) { return P ? P->bytes : 0; }

int pb_spmv(PbPlan *P, const double *x, double *y, double alpha, double beta, hipStream_t s)
{
    if (P->n_split)
        hipLaunchKernelGGL(pb_scale_rows_kernel, dim3(kBand / 256, P->n_split), dim3(256), 0, s, P->split_bands.as<int>(), P->rows, y, beta);
    if (P->n_pitems)
        hipLaunchKernelGGL(pb_producer_kernel, dim3(P->n_pitems), dim3(kPbThreads), P->lds_producer, s, P->pitems.as<ProducerItem>(), P->cols, P->RB,
                           P->p_lcol.as<unsigned short>(), P->stP.as<int>(), P->p_val.as<double>(), P->delta.as<int>(), x, P->prod.as<double>());
    if (P->n_citems)
        hipLaunchKernelGGL(pb_consumer_kernel, dim3(P->n_citems), dim3(kPbThreads), P->lds_consumer, s, P->citems.as<ConsumerItem>(), P->rows,
                           P->c_lrow.as<unsigned short>(), P->prod.as<double>(), y, alpha, beta);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

// Locality probe: over a sample of 2048-nonzero windows, how many distinct 128-byte lines of x does a window touch per nonzero?
// Stencil / banded matrices: ≈0.01–0.1 (neighbouring rows share lines). Power-law graphs: ≈0.8–1. Blocking pays above ≈0.5
// and only when x is far larger than an XCD's 4 MiB L2.
bool pb_should_use(int rows, int cols, long long nnz, const int *d_colids)
{
    (void)rows;
    if ((long long)cols * 8 < (32ll << 20) || nnz < (4ll << 20)) return false;
    const int W = 2048, S = 64;
    std::vector<int> h(W);
    double ratio = 0.0;
    int used = 0;
    for (int sidx = 0; sidx < S; ++sidx) {
        const long long k0 = (nnz - W) / S * sidx;
        if (hipMemcpy(h.data(), d_colids + k0, sizeof(int) * W, hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (auto &c : h) c >>= 4;
        std::sort(h.begin(), h.end());
        ratio += (double)(std::unique(h.begin(), h.end()) - h.begin()) / W;
        ++used;
    }
    return used > 0 && ratio / used > 0.5;
}

} // namespace g4s
