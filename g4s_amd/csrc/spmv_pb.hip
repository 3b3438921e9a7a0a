// spmv_pb.hip — propagation-blocked fp64 SpMV for matrices whose x gathers have no locality (power-law graphs).
//
// Why: on R-MAT 10M/1e8 the row-streaming CSR kernel (spmv.hip) is bound by the x gather, not by the matrix stream — every
// 8-byte gather that misses L2 moves a 128-byte line (PMC: 6.1 GB fetched for 1.4 GB of algorithmic bytes; the gathers alone
// cost 0.89 ms, tools/gather_probe.hip). Blocking turns both random sides into streams:
//   producer  — nonzeros regrouped by 16K-column band. A workgroup loads that band of x into LDS (128 KiB); each lane takes a
//               pair of consecutive entries (local column u16, value f64; unit-stride loads), multiplies, and the four lanes
//               of an 8-entry span sum the products of equal rows (a "micro-run": same row, same cell, same span) before
//               writing them — on R-MAT only ≈0.57 sums per nonzero leave the producer (tools/cell_hist.py: 0.51 distinct
//               (row, column band) pairs per nonzero).
//   consumer  — micro-run sums regrouped by 16K-row band. A workgroup keeps that band of y in LDS, streams (sum f64, local row
//               u16), accumulates with LDS fp64 atomics, and writes the band of y once.
// HBM traffic ≈ 10.6 B/nonzero read + 4.4 written by the producer and 5.5 read by the consumer, all sequential, instead of
// ≈62 B/nonzero of 128-byte line fetches.
// The cell (column band c, row band r) holds its entries in CSR order (a stable regrouping), so the micro-runs of a cell are
// numbered consecutively on both sides and one per-cell offset maps a producer micro-run to its consumer slot.
// The regrouping is built once per matrix (g4s_csr_create) with the library's own radix sort and scan (prims.hpp; round 5 — rocPRIM was the last vendor-library
// call in the product); the values are stored a second time in producer order. Sums are accumulated by LDS atomics: equal to the oracle within the fp64 tolerance, not bit for bit,
// and the last bits may differ from run to run.
#include "common.hpp"
#include "spmv_pb.hpp"
#include "prims.hpp"
#include <algorithm>
#include <memory>
#include <vector>

namespace g4s {

namespace {

constexpr int kBandBits = 14;
constexpr int kBand = 1 << kBandBits;        // 16384 columns / rows per band: 128 KiB of fp64 in LDS
constexpr int kMaxHotBands = 64;             // popularity-ranked column bands in front of the natural ones
constexpr double kHotFactor = 4.0;           // a hot band must hold this many times the nonzeros of an average natural band
constexpr int kPbThreads = 1024;
constexpr int kSpan = 8;                     // entries per span (= 4 lanes × a pair each); cells are padded to a multiple of it
constexpr int kWindow = 32;                  // entries per merge window (= the 16 lanes of a DPP row × a pair each = 4 spans): a micro-run
                                             // is a run of one row inside one cell and one window; column bands start at multiples of it
constexpr int kProducerChunk = 1 << 17;      // entries per producer workgroup (x band load amortised over ≥ 1.3 MiB of stream)
constexpr int kConsumerChunk = 1 << 17;      // micro-runs per consumer workgroup of a split (heavy) row band
constexpr unsigned kPadFlag = 0x8000u;       // local-column flag of a pad slot (its product is forced to 0)

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned short ushort4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

#ifndef G4S_PB_NT
#define G4S_PB_NT 1
#endif
template <typename T>
__device__ __forceinline__ T pb_stream_load(const T *p)
{
#if G4S_PB_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        hipError_t e = g4s::device_malloc(&p, n);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        bytes = n;
        return G4S_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// A transient of the plan build: a piece of the build's arena (runtime.cpp: arena_enter … arena_leave around pb_build) — no driver call to get it and none to
// give it back. (Round 5: the transients were hipMalloc / hipFree pairs, twenty of them per create, every hipFree a device-wide wait; and the first scratch
// request of a process created the stream-ordered pool, 3 ms in the middle of the first create.)
struct TmpBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~TmpBuf() { release(); }
    int alloc(size_t n) { release(); bytes = n; return scratch_alloc(&p, n ? n : 1, nullptr); }
    void release() { if (p) { scratch_free(p, nullptr); p = nullptr; bytes = 0; } }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct BuildArena { BuildArena() { arena_enter(); } ~BuildArena() { arena_leave(nullptr, false); } };

inline int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }
inline int prims_bits(int max_value) { int b = 1; while (b < 31 && (max_value >> b)) ++b; return b; }   // significant bits of the largest key

// ================================================================================================ plan construction kernels
// Hot column bands. In a power-law graph a few thousand columns hold a third of the nonzeros, but their ids are scattered over
// the natural 16 K bands (R-MAT: ids with few 1-bits), so a row's entries in popular columns land in different bands and each
// costs its own partial sum. Columns are therefore ranked by degree; the top H·16 K go to H extra "hot" bands in rank order (a row
// now merges all its entries of one hot band into one micro-run), the rest keep their natural band. On C2 the distinct
// (row, band) pairs per nonzero drop from 0.512 to 0.324 (tools/hot_band_probe.py). colmap[col] = (band' << 14) | local column;
// the x values of the hot columns are gathered into a dense hot_x (H·128 KiB) at the start of every product.
// (Round 5. The first form was a bare atomicAdd(&deg[colids[k]], 1) per nonzero: 9.4 ms of the 51 ms plan on configs[1] — a power-law column distribution puts
// 10^5 increments on each of a few addresses, and same-address atomics serialise in L2. Now every workgroup counts its contiguous share of the entries through
// a small direct-mapped table in LDS: a column that owns its slot is counted there (the popular columns claim theirs within the first few hundred entries), a
// column that finds its slot taken by another goes to HBM as before — those are the unpopular ones, spread over millions of addresses — and the table is
// flushed with one atomic per used slot. Same counts, whatever the interleaving.)
constexpr int kDegSlots = 4096, kDegBlocks = 2048;
__global__ __launch_bounds__(256) void pb_col_degree_kernel(long long nnz, const int *__restrict__ colids, int *__restrict__ deg)
{
    __shared__ int s_key[kDegSlots], s_cnt[kDegSlots];
    for (int i = threadIdx.x; i < kDegSlots; i += 256) { s_key[i] = -1; s_cnt[i] = 0; }
    __syncthreads();
    const long long per = (nnz + gridDim.x - 1) / gridDim.x, k0 = per * blockIdx.x, k1 = k0 + per < nnz ? k0 + per : nnz;
    for (long long k = k0 + threadIdx.x; k < k1; k += 256) {
        const int c = colids[k];
        const int h = (int)(((unsigned)c * 2654435761u) >> 20);      // 12 bits
        int owner = s_key[h];
        if (owner == -1) { const int old = atomicCAS(&s_key[h], -1, c); owner = old == -1 ? c : old; }
        if (owner == c) atomicAdd(&s_cnt[h], 1);
        else atomicAdd(&deg[c], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kDegSlots; i += 256)
        if (s_cnt[i]) atomicAdd(&deg[s_key[i]], s_cnt[i]);
}
// Large matrices (round 5): even with the popular columns counted in LDS the kernel above took 7.9 ms on configs[1] — 10^8 atomics on 10^7 counters are scattered
// 64-byte read-modify-writes, and the chip does about 2·10^10 of those per second whoever collides (MI355X_MICROARCH.md, global atomics: 64 lanes in 64 rows).
// So the column ids are first PARTITIONED by their top 12 bits (three passes of the library's radix sort: bins of at most 4 096 columns, in descending bin
// order), and every 64 K-entry piece of the partitioned list is counted by one workgroup: bin by bin in a 4 096-counter LDS table, flushed with one atomic per
// non-zero counter (a bin's piece of fewer than 1 024 entries goes to HBM directly). A first form with 256 bins and one workgroup per (bin, 8 192-column
// sub-range) ran 17.9 ms: R-MAT puts 11 % of the entries into bin 0, and its eight workgroups walked 11 M entries each.
constexpr int kDegBinBits = 12, kDegPiece = 1 << 16;
__global__ void pb_bin_keys_kernel(long long nnz, const int *__restrict__ colids, int shift, int *__restrict__ key)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) key[k] = colids[k] >> shift;
}
__global__ __launch_bounds__(256) void pb_bin_degree_kernel(int n, const int *__restrict__ bins_desc /* sorted, descending */, const int *__restrict__ cols_sorted, int shift,
                                                            int *__restrict__ deg /* zeroed */)
{
    extern __shared__ int s_cnt[];                                 // 1 << shift counters
    __shared__ int s_end;
    const int width = 1 << shift;
    const int k0 = blockIdx.x * kDegPiece, k1 = min(n, k0 + kDegPiece);
    int k = k0;
    while (k < k1) {                                               // uniform: one bin's part of the piece per turn
        const int bin = bins_desc[k];
        if (threadIdx.x == 0) {                                    // the end of this bin inside the piece: first index whose key is smaller
            int lo = k, hi = k1;
            while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (bins_desc[mid] >= bin) lo = mid + 1; else hi = mid; }   // (lo + hi overflows int past 2^30 entries: C of configs[2] has 1.94e9)
            s_end = lo;
        }
        __syncthreads();
        const int e = s_end, base = bin << shift;
        if (e - k < 1024) {
            for (int i = k + threadIdx.x; i < e; i += 256) atomicAdd(&deg[cols_sorted[i]], 1);
        } else {
            for (int i = threadIdx.x; i < width; i += 256) s_cnt[i] = 0;
            __syncthreads();
            for (int i = k + threadIdx.x; i < e; i += 256) atomicAdd(&s_cnt[cols_sorted[i] - base], 1);
            __syncthreads();
            for (int i = threadIdx.x; i < width; i += 256)
                if (s_cnt[i]) atomicAdd(&deg[base + i], s_cnt[i]);
        }
        __syncthreads();                                           // (s_end and the table are rewritten by the next turn)
        k = e;
    }
}
__global__ __launch_bounds__(256) void pb_max_kernel(int n, const int *__restrict__ v, int *__restrict__ out)
{
    int m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void pb_iota_kernel(int n, int *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
__global__ void pb_colmap_kernel(int cols, int H, const int *__restrict__ order /* columns by descending degree */, unsigned *__restrict__ colmap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cols) return;
    if (H == 0) { colmap[i] = (unsigned)i; return; }
    // natural position first (every thread writes its own column), hot columns are overwritten by pb_colmap_hot_kernel afterwards
    colmap[i] = (((unsigned)i >> kBandBits) + (unsigned)H) << kBandBits | ((unsigned)i & (kBand - 1));
}
__global__ void pb_colmap_hot_kernel(int nhot, const int *__restrict__ order, unsigned *__restrict__ colmap)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nhot) colmap[order[r]] = (unsigned)r;                   // band' = r >> 14 < H, local = r & 16383
}

// The regrouping key of CSR entry k is (column band, row band). The entries arrive in CSR order — their row band never decreases — so a STABLE sort by column band
// alone gives the (column band, row band) order: key[k] = CB − 1 − band for prims' descending sort; rowid[k] = row of entry k; idx[k] = k.
__global__ void pb_keys_kernel(int rows, long long nnz, const int *__restrict__ rowptr, const int *__restrict__ colids, const unsigned *__restrict__ colmap,
                               int CB, int *__restrict__ key, int *__restrict__ rowid, int *__restrict__ idx)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = rows;                       // last r with rowptr[r] <= k
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (rowptr[mid] <= k) lo = mid; else hi = mid;
        }
        key[k] = CB - 1 - (int)(colmap[colids[k]] >> kBandBits);
        rowid[k] = lo;
        idx[k] = (int)k;
    }
}
// the full key of every sorted position: (column band << bits) | row band
__global__ void pb_full_keys_kernel(long long nnz, const int *__restrict__ sorted_ckey, const int *__restrict__ perm, const int *__restrict__ rowid, int CB, int band_key_bits,
                                    unsigned *__restrict__ key_s)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x)
        key_s[i] = ((unsigned)(CB - 1 - sorted_ckey[i]) << band_key_bits) | ((unsigned)rowid[perm[i]] >> kBandBits);
}
// the windows of the locality probe (pb_should_use), gathered into one buffer: window w = W consecutive column ids from position w·stride on
__global__ void pb_sample_kernel(int W, int S, long long stride, const int *__restrict__ colids, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W * S) out[i] = colids[(long long)(i / W) * stride + (i % W)];
}

// start[q] = first sorted position whose key >= key(q), q = c·RB + r; start[ncells] = nnz
__global__ void pb_cell_starts_kernel(long long nnz, const unsigned *__restrict__ sorted_keys, int CB, int RB, int band_key_bits, int *__restrict__ start)
{
    const long long ncells = (long long)CB * RB;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q <= ncells; q += (long long)gridDim.x * blockDim.x) {
        if (q == ncells) { start[q] = (int)nnz; continue; }
        const unsigned key = ((unsigned)(q / RB) << band_key_bits) | (unsigned)(q % RB);
        long long lo = 0, hi = nnz;
        while (lo < hi) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        start[q] = (int)lo;
    }
}

// Padded producer layout: sorted entry i of cell q lives at i + shift[q] (cells start at multiples of kSpan).
__global__ void pb_fill_producer_kernel(long long nnz, const int *__restrict__ colids, const unsigned *__restrict__ colmap, const double *__restrict__ values, const int *__restrict__ rowid,
                                        const unsigned *__restrict__ perm, const unsigned *__restrict__ sorted_keys, int band_key_bits, int RB,
                                        const int *__restrict__ shift, unsigned short *__restrict__ p_lcol, double *__restrict__ p_val,
                                        int *__restrict__ t_row, int *__restrict__ t_cell, int *__restrict__ p_src /* may be NULL: the CSR index behind every slot (g4s_csr_update_values) */)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const unsigned k = perm[i], key = sorted_keys[i];
        const int q = (int)((long long)(key >> band_key_bits) * RB + (key & ((1u << band_key_bits) - 1u)));
        const long long pos = i + shift[q];
        p_lcol[pos] = (unsigned short)(colmap[colids[k]] & (kBand - 1));
        p_val[pos] = values[k];
        if (p_src) p_src[pos] = (int)k;
        t_row[pos] = rowid[k];
        t_cell[pos] = q;
    }
}

// New values into the stored order (g4s_csr_update_values): slot j holds CSR entry p_src[j] (pads: −1, their value stays 0). The gathers are nearly
// sequential — a cell keeps the CSR order of its entries — so this is one pass over the producer stream. (Round 4 tried four slots per thread, the index loads and
// the gathers in flight together: 1.31 against 1.18–1.21 ms — the pass is bound by the lines its gathers touch, ≈ 6 GB for 1e8 entries, not by their latency.)
__global__ void pb_update_values_kernel(long long slots, const int *__restrict__ p_src, const double *__restrict__ values, double *__restrict__ p_val)
{
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < slots; j += (long long)gridDim.x * blockDim.x) {
        const int k = __builtin_nontemporal_load(p_src + j);        // the two streams pass through; the cache is left to the gathered values
        if (k >= 0) __builtin_nontemporal_store(values[k], p_val + j);
    }
}

// Micro-run heads of a span: a real entry whose row differs from the previous entry's (or that opens the span). t_row < 0 marks pads.
__global__ void pb_span_heads_kernel(long long nspans, const int *__restrict__ t_row, const int *__restrict__ t_cell,
                                     unsigned char *__restrict__ masks, int *__restrict__ counts)
{
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < nspans; s += (long long)gridDim.x * blockDim.x) {
        unsigned m = 0;
        // a run continues from the previous span when that span is in the same window and the same cell (pads exist only at the
        // end of a cell, so the previous span of a cell is full and its last entry is real)
        int prev = -2;
        if ((s * kSpan) % kWindow != 0 && t_row[s * kSpan] >= 0 && t_row[s * kSpan - 1] >= 0 && t_cell[s * kSpan] == t_cell[s * kSpan - 1])
            prev = t_row[s * kSpan - 1];
        for (int j = 0; j < kSpan; ++j) {
            const int r = t_row[s * kSpan + j];
            if (r >= 0 && r != prev) m |= 1u << j;
            if (r >= 0) prev = r;
        }
        masks[s] = (unsigned char)m;
        counts[s] = __popc(m);
    }
}

__global__ void pb_gather_kernel(long long n, const int *__restrict__ index, const int *__restrict__ src, int *__restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[index[i]];
}

// mbase[s] += delta[cell of span s]: from the span's first micro-run index to its consumer slot.
__global__ void pb_span_slots_kernel(long long nspans, const int *__restrict__ t_cell, const int *__restrict__ delta, int *__restrict__ mbase)
{
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < nspans; s += (long long)gridDim.x * blockDim.x)
        mbase[s] += delta[t_cell[s * kSpan]];
}

// Local row of every micro-run, at its consumer slot.
__global__ void pb_fill_consumer_kernel(long long nspans, const int *__restrict__ t_row, const int *__restrict__ t_cell,
                                        const unsigned char *__restrict__ masks, const int *__restrict__ mbase,
                                        unsigned short *__restrict__ c_lrow)
{
    for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < nspans; s += (long long)gridDim.x * blockDim.x) {
        const unsigned m = masks[s];
        if (!m) continue;
        int dst = mbase[s];   // already a consumer slot (pb_span_slots_kernel)
        for (int j = 0; j < kSpan; ++j)
            if ((m >> j) & 1u) c_lrow[dst++] = (unsigned short)(t_row[s * kSpan + j] & (kBand - 1));
    }
}

// ================================================================================================ SpMV kernels
struct ProducerItem { int cband, s0, s1, pad; };     // spans [s0, s1) of one column band
struct ConsumerItem { int rband, k0, k1, split; };   // micro-run slots [k0, k1) of one row band (multiples of 4)

// Lane i of a 16-lane DPP row reads lane i+SHIFT of the same row (row_shl: a VALU move, no LDS crossbar like ds_bpermute); lanes
// whose source falls past the row read 0.
template <int SHIFT>
__device__ __forceinline__ int row_down_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x100 + SHIFT, 0xF, 0xF, true);
}
template <int SHIFT>
__device__ __forceinline__ double row_down_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = row_down_i<SHIFT>((int)(b & 0xFFFFFFFFll)), hi = row_down_i<SHIFT>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Producer: one lane per PAIR of consecutive entries (unit-stride 4-byte / 16-byte loads), four lanes per 8-entry span, sixteen
// per 32-entry window. The products of a window are summed per micro-run with a backward segmented scan across the window's
// sixteen lanes (four DPP row shifts), and each lane stores the sums of the micro-runs that START in its pair (0, 1 or 2 stores).
#ifndef G4S_PB_PAIR_UNROLL
#define G4S_PB_PAIR_UNROLL 4
#endif
constexpr int kPairUnroll = G4S_PB_PAIR_UNROLL;

__global__ __launch_bounds__(kPbThreads) void pb_producer_kernel(const ProducerItem *__restrict__ items, int cols, int RB, int H, const double *__restrict__ hot_x,
                                                                  const unsigned short *__restrict__ p_lcol, const double *__restrict__ p_val,
                                                                  const unsigned char *__restrict__ masks, const int *__restrict__ mbase /* consumer slot of each span's first micro-run */,
                                                                  const double *__restrict__ x, double *__restrict__ prod)
{
    extern __shared__ double pb_lds[];
    double *xs = pb_lds;                                           // kBand doubles
    (void)RB;
    const ProducerItem it = items[blockIdx.x];
    const bool hot = it.cband < H;                                 // hot bands read the gathered copy; the others a natural 16 K slice of x
    const int c0 = hot ? it.cband << kBandBits : (it.cband - H) << kBandBits;
    const double *xsrc = hot ? hot_x : x;
    const int xlimit = hot ? H << kBandBits : cols;
    const int p_end = it.s1 * 4, p_last = p_end - 1;               // pair indices: pair p = entries 2p, 2p+1; span = p >> 2
    constexpr int STEP = kPbThreads * kPairUnroll;
    int base = it.s0 * 4 + (int)threadIdx.x;
    unsigned lc[kPairUnroll], lc_n[kPairUnroll], mk[kPairUnroll], mk_n[kPairUnroll];
    int mb[kPairUnroll], mb_n[kPairUnroll];
    double2_t v[kPairUnroll], v_n[kPairUnroll];
    // first tile's stream loads go out before the x band is staged: their latency hides under the staging
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
        const long long p = min(base + u * kPbThreads, p_last);
        lc[u] = pb_stream_load(reinterpret_cast<const unsigned *>(p_lcol) + p);
        v[u] = pb_stream_load(reinterpret_cast<const double2_t *>(p_val) + p);
        mk[u] = masks[p >> 2];
        mb[u] = mbase[p >> 2];
    }
#ifdef G4S_PB_STAGE_LOOP   /* A/B only (tools/build_variant.sh): the staging loop as it was until round 4 */
    for (int i = threadIdx.x; i < kBand; i += kPbThreads) xs[i] = (c0 + i < xlimit) ? xsrc[c0 + i] : 0.0;
#else
    {
        // The band of x: ALL of a thread's loads first, then the LDS writes. As a plain loop this compiled to 16 × (load, wait, write) — sixteen round trips in a
        // row, ≈ 6 µs per work item with nothing to overlap them (one workgroup per CU), found in round 4 in the ISA. The index is clamped instead of predicated
        // so that the loads carry no branch. (Then tried: one persistent workgroup per CU walking the items with the NEXT item's descriptor and band of x loaded into
        // registers under the current item's main loop — 0.416 against 0.400 ms on one box, 256 or 512 workgroups: the static order loses more balance than the hidden
        // ≈ 3 µs per item buy.)
        constexpr int kPerThread = kBand / kPbThreads;
        static_assert(kBand % kPbThreads == 0, "band staging");
        double xv[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = c0 + (int)threadIdx.x + k * kPbThreads;
            xv[k] = xsrc[min(i, xlimit - 1)];
        }
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = (int)threadIdx.x + k * kPbThreads;
            xs[i] = (c0 + i < xlimit) ? xv[k] : 0.0;
        }
    }
#endif
    __syncthreads();
    const int q = threadIdx.x & 3;                                 // position of this lane's pair inside its span
    for (; base - (int)threadIdx.x < p_end; base += STEP) {        // uniform trip count per workgroup: every lane takes part in the shuffles
        const bool more = base - (int)threadIdx.x + STEP < p_end;
        if (more) {
#pragma unroll
            for (int u = 0; u < kPairUnroll; ++u) {
                const long long p = min(base + STEP + u * kPbThreads, p_last);
                lc_n[u] = pb_stream_load(reinterpret_cast<const unsigned *>(p_lcol) + p);
                v_n[u] = pb_stream_load(reinterpret_cast<const double2_t *>(p_val) + p);
                mk_n[u] = masks[p >> 2];
                mb_n[u] = mbase[p >> 2];
            }
        }
        // (round 4 tried all of a tile's x reads from LDS up front and unconditional instead of one under each pad test: 0.3843 / 0.3797 / 0.3850 ms against
        // 0.3812 / 0.3796 / 0.3810 as written here, alternating processes on one box — the waves of the other SIMDs cover those waits already)
#pragma unroll
        for (int u = 0; u < kPairUnroll; ++u) {
            const int p = base + u * kPbThreads;
            const bool live = p < p_end;                           // whole spans are live or not (p_end is a multiple of 4)
            const unsigned l0 = lc[u] & 0xFFFFu, l1 = lc[u] >> 16;
            const double p0 = (!live || (l0 & kPadFlag)) ? 0.0 : v[u][0] * xs[l0 & (kBand - 1)];
            const double p1 = (!live || (l1 & kPadFlag)) ? 0.0 : v[u][1] * xs[l1 & (kBand - 1)];
            const unsigned m = live ? mk[u] : 0u;
            const bool h0 = (m >> (2 * q)) & 1u, h1 = (m >> (2 * q + 1)) & 1u;
            // open prefix: the part of this pair that continues a micro-run begun in an earlier lane of the span
            const double op = h0 ? 0.0 : (h1 ? p0 : p0 + p1);
            const bool closed = h0 | h1;
            // S_j = op_j + (closed_j ? 0 : S_{j+1}) over the 16 lanes of the window, by doubling; ext = S of the next lane = what the
            // following lanes add to the micro-run that contains this pair's last entry
            double S = op;
            int f = (int)closed;
#define G4S_PB_SCAN_STEP(D) { const double sv = row_down_d<D>(S); const int sf = row_down_i<D>(f); if (!f) S += sv; f |= sf; }
            G4S_PB_SCAN_STEP(1) G4S_PB_SCAN_STEP(2) G4S_PB_SCAN_STEP(4) G4S_PB_SCAN_STEP(8)
#undef G4S_PB_SCAN_STEP
            const double ext = row_down_d<1>(S);
            if (live && closed) {
                const int d0 = mb[u];
                if (h0) prod[d0 + __popc(m & ((1u << (2 * q)) - 1u))] = h1 ? p0 : p0 + p1 + ext;
                if (h1) prod[d0 + __popc(m & ((1u << (2 * q + 1)) - 1u))] = p1 + ext;
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < kPairUnroll; ++u) { lc[u] = lc_n[u]; v[u] = v_n[u]; mk[u] = mk_n[u]; mb[u] = mb_n[u]; }
        }
    }
}

// One launch ahead of the producer: blocks [0, n_split·64) pre-scale y for the row bands whose sums arrive from several consumer
// workgroups (those add into y with atomics), the remaining blocks gather the x values of the hot columns into hot_x.
__global__ void pb_prepare_kernel(int n_split, const int *__restrict__ split_bands, int rows, double *__restrict__ y, double beta,
                                  int nhot, const int *__restrict__ hot_cols, const double *__restrict__ x, double *__restrict__ hot_x)
{
    constexpr int kBlocksPerBand = kBand / 256;
    const int b = blockIdx.x;
    if (b < n_split * kBlocksPerBand) {
        const int r0 = split_bands[b / kBlocksPerBand] << kBandBits;
        const int i = r0 + (b % kBlocksPerBand) * 256 + (int)threadIdx.x;
        if (i < rows) y[i] = beta == 0.0 ? 0.0 : beta * y[i];
    } else {
        const int r = (b - n_split * kBlocksPerBand) * 256 + (int)threadIdx.x;
        if (r < nhot) hot_x[r] = x[hot_cols[r]];
    }
}

#ifndef G4S_PB_CONS_UNROLL
#define G4S_PB_CONS_UNROLL 2
#endif
constexpr int kPbUnroll = G4S_PB_CONS_UNROLL;   // consumer: groups of 4 consecutive slots per thread per iteration

__global__ __launch_bounds__(kPbThreads) void pb_consumer_kernel(const ConsumerItem *__restrict__ items, int rows,
                                                                  const unsigned short *__restrict__ c_lrow, const double *__restrict__ prod,
                                                                  double *__restrict__ y, double alpha, double beta)
{
    extern __shared__ double pb_lds[];
    double *ys = pb_lds;
    const ConsumerItem it = items[blockIdx.x];                     // [k0, k1): multiples of 4; pad slots carry 0 for local row 0
    const int last_group = it.k1 - 4;
    constexpr int STEP = 4 * kPbThreads * kPbUnroll;
    int base = it.k0 + 4 * (int)threadIdx.x;
    ushort4_t lr[kPbUnroll], lr_n[kPbUnroll];
    double2_t pa[kPbUnroll], pb[kPbUnroll], pa_n[kPbUnroll], pb_n[kPbUnroll];
    if (it.k1 > it.k0) {
#pragma unroll
        for (int u = 0; u < kPbUnroll; ++u) {
            const int k = min(base + u * 4 * kPbThreads, last_group);
            lr[u] = pb_stream_load(reinterpret_cast<const ushort4_t *>(c_lrow + k));
            pa[u] = pb_stream_load(reinterpret_cast<const double2_t *>(prod + k));
            pb[u] = pb_stream_load(reinterpret_cast<const double2_t *>(prod + k + 2));
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
                lr_n[u] = pb_stream_load(reinterpret_cast<const ushort4_t *>(c_lrow + k));
                pa_n[u] = pb_stream_load(reinterpret_cast<const double2_t *>(prod + k));
                pb_n[u] = pb_stream_load(reinterpret_cast<const double2_t *>(prod + k + 2));
            }
        }
#pragma unroll
        for (int u = 0; u < kPbUnroll; ++u) {
            const bool active = base + u * 4 * kPbThreads < it.k1;     // varies per lane in the tail wave of a band: the cross-lane part below runs
                                                                        // for the whole wave, lanes past the end contribute zeros
            double p[4] = {pa[u][0], pa[u][1], pb[u][0], pb[u][1]};
            // Consecutive slots of one row (a hub row cut into many micro-runs inside a hot cell) would hit one LDS address from
            // every lane of the wave, and ds_add_f64 serialises same-address lanes. Equal neighbours are summed in registers
            // first; a wave whose 256 slots all belong to one row reduces across lanes and issues a single atomic.
            const unsigned r0_ = lr[u][0], r3_ = lr[u][3];
            const bool lane_uniform = r0_ == r3_ && lr[u][1] == r0_ && lr[u][2] == r0_;
            const unsigned first_row = (unsigned)__builtin_amdgcn_readfirstlane((int)r0_);
            if (__all(active && lane_uniform && r0_ == first_row)) {     // wave-uniform branch, every lane active: the shuffles read live registers
                double t = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
                if ((threadIdx.x & 63) == 0) atomicAdd(&ys[first_row], t);
            } else if (active) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (lr[u][j] == lr[u][j + 1]) { p[j + 1] += p[j]; p[j] = 0.0; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p[j] != 0.0) atomicAdd(&ys[lr[u][j]], p[j]);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < kPbUnroll; ++u) { lr[u] = lr_n[u]; pa[u] = pa_n[u]; pb[u] = pb_n[u]; }
        }
    }
    __syncthreads();
    const int r0 = it.rband << kBandBits;
    // All of a thread's sums out of LDS first, and — beta != 0 — all its old y values in flight together: as a loop with a break every row waited for its own
    // LDS read, and for its own round trip to y when beta != 0 (sixteen in a row per work item). Neutral for the bench line (beta = 0: 0.3910 against 0.3919 ms).
    constexpr int kRowsPerThread = kBand / kPbThreads;
    double yv[kRowsPerThread], yo[kRowsPerThread];
    const bool read_y = !it.split && beta != 0.0;                  // (uniform)
#pragma unroll
    for (int k = 0; k < kRowsPerThread; ++k) {
        yv[k] = ys[(int)threadIdx.x + k * kPbThreads];
        const int r = min(r0 + (int)threadIdx.x + k * kPbThreads, rows - 1);
        yo[k] = read_y ? y[r] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kRowsPerThread; ++k) {
        const int r = r0 + (int)threadIdx.x + k * kPbThreads;
        if (r < rows) {
            if (it.split) {
                if (yv[k] != 0.0) atomicAdd(&y[r], alpha * yv[k]);  // y was pre-scaled by beta (pb_prepare_kernel)
            } else {
                y[r] = beta == 0.0 ? alpha * yv[k] : alpha * yv[k] + beta * yo[k];
            }
        }
    }
}

} // namespace

struct PbPlan {
    int rows = 0, cols = 0, CB = 0, RB = 0;
    long long nnz = 0, micro_runs = 0;
    DevBuf p_lcol, p_val, masks, mbase, c_lrow, prod, delta, pitems, citems, split_bands, hot_cols, hot_x;
    DevBuf p_src;                                                   // G4S_SPMV_UPDATABLE: CSR index of every producer slot
    long long slots = 0;
    int n_pitems = 0, n_citems = 0, n_split = 0, H = 0;
    size_t lds_producer = 0, lds_consumer = 0;
    long long bytes = 0;
};

int pb_build(PbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values, bool keep_value_map)
{
    *out = nullptr;
    if (nnz <= 0 || rows <= 0 || cols <= 0) return set_error(G4S_ERR_INVALID, "pb_build: empty matrix");
    auto P = new (std::nothrow) PbPlan();
    if (!P) return set_error(G4S_ERR_NOMEM, "host allocation failed");
    std::unique_ptr<PbPlan> guard(P);
    P->rows = rows; P->cols = cols; P->nnz = nnz;
    BuildArena arena;                                              // the build's transients (TmpBuf) live in it; declared before them, so it ends after them
    // 0. hot column bands: rank the columns by degree, take the leading 16 K-column groups that are much denser than a natural band
    const int CBnat = (cols + kBand - 1) >> kBandBits;
    TmpBuf colmap, deg, deg_s, order_in, order, tmp0;
    G4S_TRY(colmap.alloc(sizeof(unsigned) * (size_t)cols));
    int H = 0;
    {
        const char *e = getenv("G4S_PB_HOT_BANDS");
        const int want = e ? atoi(e) : -1;                          // -1 = decide from the degree distribution
        const int Hmax = std::min(kMaxHotBands, cols / (2 * kBand));
        if (want != 0 && Hmax > 0) {
            G4S_TRY(deg.alloc(sizeof(int) * (size_t)cols)); G4S_TRY(deg_s.alloc(sizeof(int) * (size_t)cols));
            G4S_TRY(order_in.alloc(sizeof(int) * (size_t)cols)); G4S_TRY(order.alloc(sizeof(int) * (size_t)cols));
            const int cbits = prims_bits(cols - 1), shift = std::max(0, cbits - kDegBinBits);
            G4S_HIP_TRY(hipMemsetAsync(deg.p, 0, deg.bytes, nullptr));
            if (cols >= (1 << 20) && nnz >= (1ll << 24) && nnz < (1ll << 31) && shift <= 13) {
                // partition by the top 12 bits of the column id, then count piece by piece in LDS (pb_bin_degree_kernel)
                TmpBuf bkey, bkey_s, bcol_s, tk, tv;
                const size_t nb = sizeof(int) * (size_t)nnz;
                G4S_TRY(bkey.alloc(nb)); G4S_TRY(bkey_s.alloc(nb)); G4S_TRY(bcol_s.alloc(nb)); G4S_TRY(tk.alloc(nb)); G4S_TRY(tv.alloc(nb));
                hipLaunchKernelGGL(pb_bin_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, nnz, d_colids, shift, bkey.as<int>());
                G4S_TRY(prims::sort_pairs_descending(bkey.as<int>(), d_colids, bkey_s.as<int>(), bcol_s.as<int>(), tk.as<int>(), tv.as<int>(), (int)nnz, prims_bits((cols - 1) >> shift), nullptr));
                hipLaunchKernelGGL(pb_bin_degree_kernel, dim3((unsigned)((nnz + kDegPiece - 1) / kDegPiece)), dim3(256), sizeof(int) << shift, nullptr, (int)nnz, bkey_s.as<int>(), bcol_s.as<int>(), shift,
                                   deg.as<int>());
            } else {
                hipLaunchKernelGGL(pb_col_degree_kernel, dim3((unsigned)std::min<long long>(kDegBlocks, (nnz + 255) / 256)), dim3(256), 0, nullptr, nnz, d_colids, deg.as<int>());
            }
            hipLaunchKernelGGL(pb_iota_kernel, dim3((cols + 255) / 256), dim3(256), 0, nullptr, cols, order_in.as<int>());
            G4S_HIP_TRY(hipGetLastError());
            int *d_max = nullptr, h_max = 0;                        // the largest degree bounds the key bits of the ranking sort
            G4S_TRY(tmp0.alloc(sizeof(int) * (2 * (size_t)cols + 1)));
            d_max = tmp0.as<int>() + 2 * (size_t)cols;
            G4S_HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int), nullptr));
            hipLaunchKernelGGL(pb_max_kernel, dim3(std::min(grid_for(cols), 1024)), dim3(256), 0, nullptr, cols, deg.as<int>(), d_max);
            G4S_HIP_TRY(hipMemcpy(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost));
            G4S_TRY(prims::sort_pairs_descending(deg.as<int>(), order_in.as<int>(), deg_s.as<int>(), order.as<int>(), tmp0.as<int>(), tmp0.as<int>() + cols, cols, prims_bits(h_max), nullptr));
            std::vector<int> top((size_t)Hmax * kBand);
            G4S_HIP_TRY(hipMemcpy(top.data(), deg_s.p, sizeof(int) * top.size(), hipMemcpyDeviceToHost));
            if (want > 0) H = std::min(want, Hmax);
            else {
                // band h of the ranking is worth its 128 KiB gather while it holds kHotFactor times the nonzeros of an average natural band
                const double avg = (double)nnz / CBnat;
                for (int h = 0; h < Hmax; ++h) {
                    long long in_band = 0;
                    for (int i = 0; i < kBand; ++i) in_band += top[(size_t)h * kBand + i];
                    if ((double)in_band < kHotFactor * avg) break;
                    H = h + 1;
                }
            }
        }
    }
    hipLaunchKernelGGL(pb_colmap_kernel, dim3((cols + 255) / 256), dim3(256), 0, nullptr, cols, H, order.as<int>(), colmap.as<unsigned>());
    if (H) {
        const int nhot = H * kBand;
        hipLaunchKernelGGL(pb_colmap_hot_kernel, dim3((nhot + 255) / 256), dim3(256), 0, nullptr, nhot, order.as<int>(), colmap.as<unsigned>());
        G4S_TRY(P->hot_cols.alloc(sizeof(int) * (size_t)nhot));
        G4S_TRY(P->hot_x.alloc(sizeof(double) * (size_t)nhot));
        G4S_HIP_TRY(hipMemcpy(P->hot_cols.p, order.p, sizeof(int) * (size_t)nhot, hipMemcpyDeviceToDevice));
    }
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipDeviceSynchronize());
    deg.release(); deg_s.release(); order_in.release(); order.release(); tmp0.release();
    P->H = H;
    const int CB = P->CB = CBnat + H;
    const int RB = P->RB = (rows + kBand - 1) >> kBandBits;
    int bits = 1;
    while ((1 << bits) < std::max(CB, RB)) ++bits;
    if (2 * bits > 32) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: too many bands");
    P->lds_producer = sizeof(double) * kBand;
    P->lds_consumer = sizeof(double) * kBand;
    const long long ncells = (long long)CB * RB;

    // 1. regroup: stable sort of the CSR entries by (column band, row band)
    TmpBuf key, key_s, idx, perm, rowid, startP;
    const size_t n4 = sizeof(unsigned) * (size_t)nnz;
    G4S_TRY(key.alloc(n4)); G4S_TRY(key_s.alloc(n4)); G4S_TRY(idx.alloc(n4)); G4S_TRY(perm.alloc(n4)); G4S_TRY(rowid.alloc(n4));
    if (nnz >= (1ll << 31)) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: more than 2^31 - 1 nonzeros");
    hipLaunchKernelGGL(pb_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, rows, nnz, d_rowptr, d_colids, colmap.as<unsigned>(), CB, key.as<int>(),
                       rowid.as<int>(), idx.as<int>());
    G4S_HIP_TRY(hipGetLastError());
    {
        TmpBuf ckey_s, tk, tv;                                      // sorted band keys, the sort's ping-pong partners
        G4S_TRY(ckey_s.alloc(n4)); G4S_TRY(tk.alloc(n4)); G4S_TRY(tv.alloc(n4));
        G4S_TRY(prims::sort_pairs_descending(key.as<int>(), idx.as<int>(), ckey_s.as<int>(), perm.as<int>(), tk.as<int>(), tv.as<int>(), (int)nnz, prims_bits(CB - 1), nullptr));
        hipLaunchKernelGGL(pb_full_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, nnz, ckey_s.as<int>(), perm.as<int>(), rowid.as<int>(), CB, bits, key_s.as<unsigned>());
        G4S_HIP_TRY(hipGetLastError());
        G4S_HIP_TRY(hipDeviceSynchronize());
    }
    key.release(); idx.release();
    G4S_TRY(startP.alloc(sizeof(int) * (size_t)(ncells + 1)));
    hipLaunchKernelGGL(pb_cell_starts_kernel, dim3(grid_for(ncells + 1)), dim3(256), 0, nullptr, nnz, key_s.as<unsigned>(), CB, RB, bits, startP.as<int>());
    G4S_HIP_TRY(hipGetLastError());

    // 2. padded producer layout (host): every cell starts at a multiple of kSpan
    std::vector<int> hP((size_t)ncells + 1);
    G4S_HIP_TRY(hipMemcpy(hP.data(), startP.p, sizeof(int) * hP.size(), hipMemcpyDeviceToHost));
    std::vector<int> padP((size_t)ncells + 1), shP((size_t)ncells);
    long long totP = 0;
    for (long long q = 0; q < ncells; ++q) {
        if (q % RB == 0) totP = (totP + kWindow - 1) & ~(long long)(kWindow - 1);   // a column band starts on a window boundary
        padP[q] = (int)totP;
        shP[q] = (int)(totP - hP[q]);
        totP += hP[q + 1] - hP[q];
        totP = (totP + kSpan - 1) & ~(long long)(kSpan - 1);
    }
    totP = (totP + kWindow - 1) & ~(long long)(kWindow - 1);
    padP[ncells] = (int)totP;
    if (totP + 64 > INT32_MAX) return set_error(G4S_ERR_UNSUPPORTED, "pb_build: padded length exceeds int32");
    const long long nspans = totP / kSpan;
    TmpBuf d_shP, t_row, t_cell, counts;
    G4S_TRY(d_shP.alloc(sizeof(int) * shP.size()));
    G4S_HIP_TRY(hipMemcpy(d_shP.p, shP.data(), sizeof(int) * shP.size(), hipMemcpyHostToDevice));
    G4S_TRY(P->p_lcol.alloc(sizeof(unsigned short) * (size_t)(totP + 64)));
    G4S_TRY(P->p_val.alloc(sizeof(double) * (size_t)(totP + 64)));
    G4S_TRY(t_row.alloc(sizeof(int) * (size_t)(totP + 64)));
    G4S_TRY(t_cell.alloc(sizeof(int) * (size_t)(totP + 64)));
    G4S_HIP_TRY(hipMemset(P->p_lcol.p, 0xFF, P->p_lcol.bytes));   // pad flag everywhere; real entries overwrite it
    G4S_HIP_TRY(hipMemset(P->p_val.p, 0, P->p_val.bytes));
    G4S_HIP_TRY(hipMemset(t_row.p, 0xFF, t_row.bytes));           // −1 = pad
    G4S_HIP_TRY(hipMemset(t_cell.p, 0, t_cell.bytes));
    P->slots = totP;
    if (keep_value_map) {
        G4S_TRY(P->p_src.alloc(sizeof(int) * (size_t)(totP + 64)));
        G4S_HIP_TRY(hipMemset(P->p_src.p, 0xFF, P->p_src.bytes));   // −1 = pad
    }
    hipLaunchKernelGGL(pb_fill_producer_kernel, dim3(grid_for(nnz)), dim3(256), 0, nullptr, nnz, d_colids, colmap.as<unsigned>(), d_values, rowid.as<int>(), perm.as<unsigned>(),
                       key_s.as<unsigned>(), bits, RB, d_shP.as<int>(), P->p_lcol.as<unsigned short>(), P->p_val.as<double>(), t_row.as<int>(), t_cell.as<int>(),
                       keep_value_map ? P->p_src.as<int>() : (int *)nullptr);
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipDeviceSynchronize());
    key_s.release(); perm.release(); rowid.release(); startP.release(); colmap.release();

    // 3. micro-runs: head mask per span, exclusive scan → index of each span's first micro-run
    G4S_TRY(P->masks.alloc((size_t)nspans + 64));
    G4S_TRY(counts.alloc(sizeof(int) * ((size_t)nspans + 1)));
    G4S_TRY(P->mbase.alloc(sizeof(int) * ((size_t)nspans + 64)));
    G4S_HIP_TRY(hipMemset(counts.p, 0, counts.bytes));
    hipLaunchKernelGGL(pb_span_heads_kernel, dim3(grid_for(nspans)), dim3(256), 0, nullptr, nspans, t_row.as<int>(), t_cell.as<int>(), P->masks.as<unsigned char>(), counts.as<int>());
    G4S_HIP_TRY(hipGetLastError());
    G4S_TRY(prims::exclusive_scan(counts.as<int>(), P->mbase.as<int>(), nspans + 1, nullptr));

    // 4. micro-run index at every cell start → consumer layout (host): row band segments at multiples of 4
    std::vector<int> span_of_cell((size_t)ncells + 1), mstart((size_t)ncells + 1);
    for (long long q = 0; q <= ncells; ++q) span_of_cell[q] = padP[q] / kSpan;
    TmpBuf d_soc, d_mstart;
    G4S_TRY(d_soc.alloc(sizeof(int) * span_of_cell.size())); G4S_TRY(d_mstart.alloc(sizeof(int) * mstart.size()));
    G4S_HIP_TRY(hipMemcpy(d_soc.p, span_of_cell.data(), sizeof(int) * span_of_cell.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pb_gather_kernel, dim3(grid_for(ncells + 1)), dim3(256), 0, nullptr, ncells + 1, d_soc.as<int>(), P->mbase.as<int>(), d_mstart.as<int>());
    G4S_HIP_TRY(hipMemcpy(mstart.data(), d_mstart.p, sizeof(int) * mstart.size(), hipMemcpyDeviceToHost));
    P->micro_runs = mstart[ncells];
    std::vector<int> h_delta((size_t)ncells), bandC((size_t)RB + 1);
    long long totC = 0;
    for (int r = 0; r < RB; ++r) {
        totC = (totC + 3) & ~3ll;
        bandC[r] = (int)totC;
        for (int c = 0; c < CB; ++c) {
            const size_t q = (size_t)c * RB + r;
            h_delta[q] = (int)(totC - mstart[q]);
            totC += mstart[q + 1] - mstart[q];
        }
    }
    totC = (totC + 3) & ~3ll;
    bandC[RB] = (int)totC;
    G4S_TRY(P->delta.alloc(sizeof(int) * h_delta.size()));
    G4S_HIP_TRY(hipMemcpy(P->delta.p, h_delta.data(), sizeof(int) * h_delta.size(), hipMemcpyHostToDevice));
    G4S_TRY(P->c_lrow.alloc(sizeof(unsigned short) * (size_t)(totC + 64)));
    G4S_TRY(P->prod.alloc(sizeof(double) * (size_t)(totC + 64)));
    G4S_HIP_TRY(hipMemset(P->c_lrow.p, 0, P->c_lrow.bytes));
    G4S_HIP_TRY(hipMemset(P->prod.p, 0, P->prod.bytes));          // pad slots stay 0 for ever: the producer never writes them
    hipLaunchKernelGGL(pb_span_slots_kernel, dim3(grid_for(nspans)), dim3(256), 0, nullptr, nspans, t_cell.as<int>(), P->delta.as<int>(), P->mbase.as<int>());
    hipLaunchKernelGGL(pb_fill_consumer_kernel, dim3(grid_for(nspans)), dim3(256), 0, nullptr, nspans, t_row.as<int>(), t_cell.as<int>(),
                       P->masks.as<unsigned char>(), P->mbase.as<int>(), P->c_lrow.as<unsigned short>());
    G4S_HIP_TRY(hipGetLastError());
    // 5. work items, heaviest first (the tail of each launch is then made of light bands)
    // Work-item sizes follow the matrix: the full C2 matrix runs best at 128 K entries / 128 K micro-runs per item (sweeps in
    // profiles/README.md); a row slab of it (the per-rank matrix of the multi-GPU bench, an eighth of the nonzeros) would then have
    // ~90 heavy producer items and ~20–80 consumer items for 256 CUs, and measured 20–25 % faster at 32 K / 16 K.
    auto pow2_at_most = [](long long v) { long long p = 1; while (p * 2 <= v) p *= 2; return p; };
    // (round 4, the 8-way slabs again, tools/dist_probe.py: 64 K / 32 K for 10–13 M entries measured 4 % faster than 32 K / 16 K — a producer item pays ≈ 6 µs
    // for staging its band of x whatever its length — and 128 K / 64 K 20 % slower: too few items for 256 CUs)
    const long long auto_pc = std::min<long long>(kProducerChunk, std::max<long long>(32768, pow2_at_most(totP / 128)));
    const long long auto_cc = std::min<long long>(kConsumerChunk, std::max<long long>(16384, pow2_at_most(P->micro_runs / 128)));
    // (Round 5, VERDICT r4 item 4: items cut so that the heavy ones make a WHOLE number of rounds over the CUs — total / 256 entries per producer item for a 12 M-entry
    // slab of configs[1] instead of 64 K — measured SLOWER: 73.0 against 65.3 µs per product, tools/small_sizes.py rmat12m. The 413 items of the power-of-two rule
    // are not one and a half rounds of equal items: the hot bands' items are long, the natural bands' short, and longer items only lengthen the round's tail.)
    const int kPC = (std::max<long long>(kSpan * kPbThreads, getenv("G4S_PB_PCHUNK") ? atoll(getenv("G4S_PB_PCHUNK")) : auto_pc) / kWindow) * (kWindow / kSpan);   // spans per item, whole windows
    const int kCC = (int)std::max<long long>(4, (getenv("G4S_PB_CCHUNK") ? atoll(getenv("G4S_PB_CCHUNK")) : auto_cc) & ~3ll);
    std::vector<ProducerItem> pit;
    for (int c = 0; c < CB; ++c) {
        const int s0 = padP[(size_t)c * RB] / kSpan, s1 = padP[(size_t)(c + 1) * RB] / kSpan;
        for (int s = s0; s < s1; s += kPC) pit.push_back(ProducerItem{c, s, std::min(s1, s + kPC), 0});
    }
    std::vector<ConsumerItem> cit;
    std::vector<int> split;
    for (int r = 0; r < RB; ++r) {
        const int b0 = bandC[r], b1 = bandC[r + 1];   // both multiples of 4; slots past the band's true end are pads
        if (b1 - b0 <= kCC) cit.push_back(ConsumerItem{r, b0, b1, 0});   // also the empty bands: their rows must still be written
        else {
            split.push_back(r);
            for (int k = b0; k < b1; k += kCC) cit.push_back(ConsumerItem{r, k, std::min(b1, k + kCC), 1});
        }
    }
    std::stable_sort(cit.begin(), cit.end(), [](const ConsumerItem &a, const ConsumerItem &b) { return (a.k1 - a.k0) > (b.k1 - b.k0); });
    std::stable_sort(pit.begin(), pit.end(), [](const ProducerItem &a, const ProducerItem &b) { return (a.s1 - a.s0) > (b.s1 - b.s0); });
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
    P->bytes = (long long)(P->p_src.bytes + P->p_lcol.bytes + P->p_val.bytes + P->masks.bytes + P->mbase.bytes + P->c_lrow.bytes + P->prod.bytes + P->delta.bytes +
                           P->pitems.bytes + P->citems.bytes + P->split_bands.bytes + P->hot_cols.bytes + P->hot_x.bytes);
    if (getenv("G4S_DEBUG"))
        fprintf(stderr, "g4s blocked SpMV plan: %d x %d bands (%d hot), nnz %lld, padded %lld, micro-runs %lld (%.3f per nonzero), %d producer / %d consumer items, %d split bands, %.2f GB\n",
                CB, RB, H, nnz, totP, P->micro_runs, (double)P->micro_runs / (double)nnz, P->n_pitems, P->n_citems, P->n_split, P->bytes / 1e9);
    *out = guard.release();
    return G4S_OK;
}

void pb_destroy(PbPlan *P) { delete P; }

long long pb_bytes(const PbPlan *P) { return P ? P->bytes : 0; }
bool pb_has_value_map(const PbPlan *P) { return P && P->p_src.p != nullptr; }

// new values (CSR order, device) into the plan's regrouped copy; needs the value map (G4S_SPMV_UPDATABLE at create)
int pb_update_values(PbPlan *P, const double *d_values, hipStream_t s)
{
    if (!P->p_src.p) return set_error(G4S_ERR_UNSUPPORTED, "the blocked plan was built without its value map");
    hipLaunchKernelGGL(pb_update_values_kernel, dim3(grid_for(P->slots)), dim3(256), 0, s, P->slots, P->p_src.as<int>(), d_values, P->p_val.as<double>());
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

int pb_spmv(PbPlan *P, const double *x, double *y, double alpha, double beta, hipStream_t s)
{
    if (P->n_split || P->H)
        hipLaunchKernelGGL(pb_prepare_kernel, dim3((P->n_split + P->H) * (kBand / 256)), dim3(256), 0, s, P->n_split, P->split_bands.as<int>(), P->rows, y, beta,
                           P->H * kBand, P->hot_cols.as<int>(), x, P->hot_x.as<double>());
    if (P->n_pitems)
        hipLaunchKernelGGL(pb_producer_kernel, dim3(P->n_pitems), dim3(kPbThreads), P->lds_producer, s, P->pitems.as<ProducerItem>(), P->cols, P->RB, P->H, P->hot_x.as<double>(),
                           P->p_lcol.as<unsigned short>(), P->p_val.as<double>(), P->masks.as<unsigned char>(), P->mbase.as<int>(), x, P->prod.as<double>());
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
    if ((long long)cols * 8 < (8ll << 20) || nnz < (4ll << 20)) return false;   // x within ~2 L2s: the gathers mostly hit (a 15 MB x still ran 2.3× faster blocked)
    const int W = 2048, S = 16;                                     // (round 5: 16 windows instead of 64 — the ratio is ≈ 0.9 for power-law graphs and ≈ 0.05 for stencils, and the 64 host-side sorts were 2 ms of every create)
    // (one gather kernel and one copy: the 64 separate 8 KB copies of the first form cost a millisecond of round trips)
    BuildArena arena;
    TmpBuf d_sample;
    if (d_sample.alloc(sizeof(int) * (size_t)W * S) != G4S_OK) return false;
    hipLaunchKernelGGL(pb_sample_kernel, dim3((W * S + 255) / 256), dim3(256), 0, nullptr, W, S, (nnz - W) / S, d_colids, d_sample.as<int>());
    std::vector<int> all((size_t)W * S);
    if (hipMemcpy(all.data(), d_sample.p, sizeof(int) * all.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    double ratio = 0.0;
    for (int sidx = 0; sidx < S; ++sidx) {
        int *h = all.data() + (size_t)sidx * W;
        for (int i = 0; i < W; ++i) h[i] >>= 4;
        std::sort(h, h + W);
        ratio += (double)(std::unique(h, h + W) - h) / W;
    }
    return ratio / S > 0.5;
}

} // namespace g4s
