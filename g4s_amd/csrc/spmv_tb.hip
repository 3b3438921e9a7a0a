// spmv_tb.hip — tile-blocked fp64 SpMV for matrices whose x gathers have no locality (power-law graphs). Round-2 EXPERIMENT next to the
// propagation-blocked path (spmv_pb.hip, the default): opt in with G4S_SPMV_IMPL=tb. Parity-green on every SpMV test, but measured SLOWER on
// configs[1] (0.61–0.66 ms against 0.40 ms): the R-MAT cells are too sparse — ≈55 000 cells of ≈1 400 entries, each costing a barrier and a
// 24 KiB x-tile staging, and gfx950's in-order vmcnt ties the short-latency x-tile loads to the HBM stream in front of them.
//
// What limited spmv_pb: every partial sum (0.35 per nonzero on R-MAT 10M) crossed HBM twice — 2.0 GB moved for 1.385 GB of
// algorithmic bytes. On-chip, a product needs x[col] and y[row] in the SAME LDS: that takes a 2-D cell dense enough to pay for
// staging its slice of x. Measured structure of configs[1] (tools/c2_structure.py): 59 % of the rows and 58 % of the columns are
// empty, the 256 K most popular columns hold 73 % of the nonzeros, and with the empty rows squeezed out a cell of 8 K rows × 4 K
// popular columns holds 500–50 000 entries. So:
//   * rows: the non-empty rows are numbered consecutively ("compact rows") and cut into y tiles of at most 8 192 compact rows with
//     about equal work; a tile's slice of y lives in LDS (64 KiB) for the whole product and is written to HBM exactly once,
//     together with the zeros of the empty rows in its range of natural rows.
//   * columns: ranked by degree; the leading ranks form x tiles of 4 096 columns (32 KiB). hot_x = x in rank order is gathered at
//     the start of every product (a few MB, L2-resident afterwards).
//   * a cell (y tile, x tile) with at least kMinCell entries is HOT: its entries are stored tile-major / x-tile-major as (local
//     column u16, local row u16, value f64) = 12 B, and tb_tile_kernel multiplies them against the staged x tile and adds into the
//     y tile with LDS atomics — no partial sum leaves the CU. The x tiles of a workgroup's cells are staged from L2 two cells ahead
//     (registers → LDS double buffer), the entry stream is prefetched one batch ahead across cell boundaries.
//   * all other entries are COLD and take the propagation route: tb_cold_kernel walks them column band by column band (16 K natural
//     columns of x in LDS), writes one product per entry with unit-stride 16-byte stores, and the tile kernel reads the products of
//     its tile back — (column band, tile) chunks, contiguous — and adds them into the same LDS y tile. 28 B per cold entry.
// HBM bytes per product on configs[1]: see DESIGN.md §4.1 (measured with the PMC passes of tools/prof_pmc.sh).
// fp64 sums are accumulated by LDS atomics: within the 1e-10 tolerance of the oracle, not bit for bit, last bits may differ from run
// to run. The plan (regrouped copy of the matrix) is built once per matrix in g4s_csr_create with rocPRIM sorts and scans.
#include "common.hpp"
#include "spmv_pb.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <memory>
#include <vector>

namespace g4s {

namespace {

constexpr int kXTile = 3072;                 // columns per hot x tile: 24 KiB of fp64, three of them in LDS
#ifndef G4S_TB_DEPTH
#define G4S_TB_DEPTH 4
#endif
constexpr int kDepth = G4S_TB_DEPTH;                    // batches of the hot entry stream in flight per thread
constexpr int kYTile = 8192;                 // compact rows per y tile: 64 KiB of fp64
constexpr int kCBandBits = 14;
constexpr int kCBand = 1 << kCBandBits;      // natural columns per cold column band: 128 KiB of fp64 in the cold kernel's LDS
constexpr int kTbThreads = 1024;
constexpr int kBatch = 2 * kTbThreads;       // hot entries per batch: one pair per thread
constexpr int kPad = 8;                      // cells are padded to a multiple of 8 entries (16-byte aligned pairs, whole 64-byte product groups)
constexpr int kMaxXTiles = 256;              // at most 768 K ranked columns
constexpr int kColdItem = 1 << 16;           // cold entries per producer work item
constexpr unsigned kPadCol = 0x8000u;        // local-column flag of a pad slot: its product is forced to 0
constexpr int kRowChunk = 64;                // y tiles start at multiples of 64 natural rows (one word of the non-empty-row bitmap)

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ T tb_stream_load(const T *p) { return __builtin_nontemporal_load(p); }

struct TbBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~TbBuf() { if (p) (void)hipFree(p); }
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
    template <typename T> int upload(const std::vector<T> &v)
    {
        G4S_TRY(alloc(sizeof(T) * v.size()));
        if (!v.empty()) G4S_HIP_TRY(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
        return G4S_OK;
    }
};

inline int tb_grid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

struct TileDesc { int b0, b1, q0, q1, r0, r1, c0, xt0, xt1, split, pad1, pad2; };   // one work item of the tile kernel: hot batches [b0,b1), cold chunks [q0,q1) of a y tile
                                                                                    // (natural rows [r0,r1), first compact row c0), x tiles of its first two cells; split: the tile's work is shared by several items
struct BatchDesc { int e0, n, first, xt2; };                        // hot entries [e0, e0+n) of ONE cell (n <= kBatch, a multiple of kPad); first batch of its cell?; x tile of the cell two cells ahead
struct ColdChunk { int e0, n; };                                    // cold entries [e0, e0+n) of one (column band, tile) cell, n a multiple of kPad
struct ColdItem { int cband, e0, e1, pad; };                        // cold entries [e0, e1) of one column band

// ================================================================================================ plan construction kernels
__global__ void tb_col_degree_kernel(long long nnz, const int *__restrict__ colids, int *__restrict__ deg)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) atomicAdd(&deg[colids[k]], 1);
}
__global__ void tb_iota_kernel(int n, int *__restrict__ v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
// colmap[col] = rank for the nhot most popular columns, 0x80000000 | col for the others
__global__ void tb_colmap_kernel(int cols, unsigned *__restrict__ colmap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cols) colmap[i] = 0x80000000u | (unsigned)i;
}
__global__ void tb_colmap_hot_kernel(int nhot, const int *__restrict__ order, unsigned *__restrict__ colmap)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nhot) colmap[order[r]] = (unsigned)r;
}
// rowid[k] = row of CSR entry k (binary search in rowptr)
__global__ void tb_rowid_kernel(int rows, long long nnz, const int *__restrict__ rowptr, int *__restrict__ rowid)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = rows;                       // last r with rowptr[r] <= k
        while (hi - lo > 1) {
            const int mid = lo + ((hi - lo) >> 1);
            if (rowptr[mid] <= k) lo = mid; else hi = mid;
        }
        rowid[k] = lo;
    }
}
// entries per candidate hot cell (tile, x tile)
__global__ void tb_cell_count_kernel(long long nnz, const int *__restrict__ rowid, const int *__restrict__ colids, const unsigned *__restrict__ colmap,
                                     const int *__restrict__ tile_of_chunk, int XT, int *__restrict__ count)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        const unsigned cm = colmap[colids[k]];
        if (cm & 0x80000000u) continue;
        atomicAdd(&count[(long long)tile_of_chunk[rowid[k] / kRowChunk] * XT + (int)(cm / kXTile)], 1);
    }
}
// key = hot cell id (tile·XT + x tile) for the entries of hot cells, n_hot_cells + (column band·NT + tile) for the others
__global__ void tb_keys_kernel(long long nnz, const int *__restrict__ rowid, const int *__restrict__ colids, const unsigned *__restrict__ colmap,
                               const int *__restrict__ tile_of_chunk, int XT, int NT, const int *__restrict__ cell_count, int min_cell,
                               unsigned *__restrict__ key, unsigned *__restrict__ idx)
{
    const unsigned cold0 = (unsigned)NT * (unsigned)XT;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        const int c = colids[k];
        const unsigned cm = colmap[c];
        const int t = tile_of_chunk[rowid[k] / kRowChunk];
        unsigned kk;
        if (!(cm & 0x80000000u) && cell_count[(long long)t * XT + (int)(cm / kXTile)] >= min_cell) kk = (unsigned)t * (unsigned)XT + (cm / kXTile);
        else kk = cold0 + ((unsigned)c >> kCBandBits) * (unsigned)NT + (unsigned)t;
        key[k] = kk;
        idx[k] = (unsigned)k;
    }
}
// start[q] = first sorted position whose key >= q, q in [0, ncells]
__global__ void tb_cell_starts_kernel(long long nnz, const unsigned *__restrict__ sorted_keys, long long ncells, int *__restrict__ start)
{
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q <= ncells; q += (long long)gridDim.x * blockDim.x) {
        long long lo = 0, hi = nnz;
        while (lo < hi) {
            const long long mid = lo + ((hi - lo) >> 1);
            if (sorted_keys[mid] < (unsigned)q) lo = mid + 1; else hi = mid;
        }
        start[q] = (int)lo;
    }
}
// Scatter the sorted entries into the padded hot / cold layouts. shift[q] = padded start of cell q − its sorted start.
__global__ void tb_fill_kernel(long long nnz, const unsigned *__restrict__ sorted_keys, const unsigned *__restrict__ perm, const int *__restrict__ rowid,
                               const int *__restrict__ colids, const double *__restrict__ values, const unsigned *__restrict__ colmap,
                               const unsigned long long *__restrict__ rowbits, const int *__restrict__ rowpre, const int *__restrict__ tile_c0,
                               int XT, int NT, const int *__restrict__ shift,
                               unsigned short *__restrict__ h_meta /* per pair: lcol0, lcol1, lrow0, lrow1 */, double *__restrict__ h_val,
                               unsigned short *__restrict__ c_lcol, unsigned short *__restrict__ c_lrow, double *__restrict__ c_val)
{
    const unsigned cold0 = (unsigned)NT * (unsigned)XT;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const unsigned q = sorted_keys[i], k = perm[i];
        const int row = rowid[k], col = colids[k];
        const bool hot = q < cold0;
        const int t = hot ? (int)(q / (unsigned)XT) : (int)((q - cold0) % (unsigned)NT);
        const int w = row / kRowChunk;
        const int lrow = rowpre[w] + __popcll(rowbits[w] & ((1ull << (row % kRowChunk)) - 1ull)) - tile_c0[t];
        const long long pos = i + shift[q];
        if (hot) {
            h_meta[(pos >> 1) * 4 + (pos & 1)] = (unsigned short)(colmap[col] % kXTile);
            h_meta[(pos >> 1) * 4 + 2 + (pos & 1)] = (unsigned short)lrow;
            h_val[pos] = values[k];
        } else {
            c_lcol[pos] = (unsigned short)(col & (kCBand - 1));
            c_lrow[pos] = (unsigned short)lrow;
            c_val[pos] = values[k];
        }
    }
}

// ================================================================================================ SpMV kernels
// One launch ahead of the products: blocks [0, n_split) pre-scale y for the rows of split y tiles (their work items add into y with
// atomics), the remaining blocks gather x in rank order into hot_x.
__global__ void tb_prepare_kernel(int n_split, const int2 *__restrict__ split_blocks, double *__restrict__ y, double beta,
                                  int nhot, const int *__restrict__ hot_cols, const double *__restrict__ x, double *__restrict__ hot_x)
{
    const int b = blockIdx.x;
    if (b < n_split) {
        const int2 rg = split_blocks[b];
        const int r = rg.x + (int)threadIdx.x;
        if (r < rg.y) y[r] = beta == 0.0 ? 0.0 : beta * y[r];
    } else {
        const int r = (b - n_split) * 256 + (int)threadIdx.x;
        if (r < nhot) hot_x[r] = x[hot_cols[r]];
    }
}

// Cold entries, one column band of x in LDS: prod[e] = val[e] · x[col[e]], a pair per lane, unit-stride 16-byte stores.
__global__ __launch_bounds__(kTbThreads) void tb_cold_kernel(const ColdItem *__restrict__ items, int cols, const unsigned short *__restrict__ c_lcol,
                                                              const double *__restrict__ c_val, const double *__restrict__ x, double *__restrict__ prod)
{
    extern __shared__ double tb_lds[];
    double *xs = tb_lds;                                           // kCBand doubles
    const ColdItem it = items[blockIdx.x];
    const int c0 = it.cband << kCBandBits;
    constexpr int U = 4;
    const long long p_end = it.e1 / 2, p_last = p_end - 1;         // pair indices
    long long base = it.e0 / 2 + (int)threadIdx.x;
    unsigned lc[U], lc_n[U];
    double2_t v[U], v_n[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long p = min(base + (long long)u * kTbThreads, p_last);
        lc[u] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lcol) + p);
        v[u] = tb_stream_load(reinterpret_cast<const double2_t *>(c_val) + p);
    }
    for (int i = threadIdx.x; i < kCBand; i += kTbThreads) xs[i] = (c0 + i < cols) ? x[c0 + i] : 0.0;
    __syncthreads();
    constexpr long long STEP = (long long)kTbThreads * U;
    for (; base - (int)threadIdx.x < p_end; base += STEP) {
        const bool more = base - (int)threadIdx.x + STEP < p_end;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long p = min(base + STEP + (long long)u * kTbThreads, p_last);
                lc_n[u] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lcol) + p);
                v_n[u] = tb_stream_load(reinterpret_cast<const double2_t *>(c_val) + p);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = base + (long long)u * kTbThreads;
            if (p < p_end) {
                const unsigned l0 = lc[u] & 0xFFFFu, l1 = lc[u] >> 16;
                double2_t o;
                o[0] = (l0 & kPadCol) ? 0.0 : v[u][0] * xs[l0 & (kCBand - 1)];
                o[1] = (l1 & kPadCol) ? 0.0 : v[u][1] * xs[l1 & (kCBand - 1)];
                reinterpret_cast<double2_t *>(prod)[p] = o;
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) { lc[u] = lc_n[u]; v[u] = v_n[u]; }
        }
    }
}

// Lane i of a 16-lane DPP row reads lane i+SHIFT (row_shl) / lane i−1 (row_shr:1) of the same row; lanes whose source falls outside read 0.
template <int SHIFT>
__device__ __forceinline__ int tb_row_down_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + SHIFT, 0xF, 0xF, true); }
template <int SHIFT>
__device__ __forceinline__ double tb_row_down_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = tb_row_down_i<SHIFT>((int)(b & 0xFFFFFFFFll)), hi = tb_row_down_i<SHIFT>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int tb_row_up1_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }

// One pair of hot entries per lane: products against the staged x tile, sums of equal-row neighbours inside the 32-entry window of a
// DPP row (backward segmented scan, four row shifts), one LDS atomic per run that starts in the lane's pair.
__device__ __forceinline__ void tb_accumulate_pair(bool live, unsigned lc, unsigned lr, double2_t v, const double *__restrict__ xb, double *__restrict__ ys)
{
    const unsigned l0 = lc & 0xFFFFu, l1 = lc >> 16;
    const unsigned r0 = lr & 0xFFFFu, r1 = lr >> 16;
    const double p0 = (!live || (l0 & kPadCol)) ? 0.0 : v[0] * xb[l0 & 0xFFFu];
    const double p1 = (!live || (l1 & kPadCol)) ? 0.0 : v[1] * xb[l1 & 0xFFFu];
    // head = the entry opens a run: its row differs from the previous entry's, or it is the first entry of the window
    const unsigned prev = (unsigned)tb_row_up1_i((int)r1);
    const bool h0 = !live || (threadIdx.x & 15) == 0 || prev != r0;
    const bool h1 = live && r1 != r0;
    const double op = h0 ? 0.0 : (h1 ? p0 : p0 + p1);             // the part of this pair that continues a run begun in an earlier lane
    const bool closed = h0 | h1;
    double S = op;
    int f = (int)closed;
#define G4S_TB_SCAN_STEP(D) { const double sv = tb_row_down_d<D>(S); const int sf = tb_row_down_i<D>(f); if (!f) S += sv; f |= sf; }
    G4S_TB_SCAN_STEP(1) G4S_TB_SCAN_STEP(2) G4S_TB_SCAN_STEP(4) G4S_TB_SCAN_STEP(8)
#undef G4S_TB_SCAN_STEP
    const double ext = tb_row_down_d<1>(S);                        // what the following lanes add to the run holding this pair's last entry
    if (live) {
        if (h0) { const double s = h1 ? p0 : p0 + p1 + ext; if (s != 0.0) atomicAdd(&ys[r0], s); }
        if (h1) { const double s = p1 + ext; if (s != 0.0) atomicAdd(&ys[r1], s); }
    }
}

__global__ __launch_bounds__(kTbThreads) void tb_tile_kernel(const TileDesc *__restrict__ tiles, const BatchDesc *__restrict__ batches, const ColdChunk *__restrict__ chunks,
                                                              const uint2_t *__restrict__ h_meta, const double *__restrict__ h_val, const double *__restrict__ hot_x,
                                                              const unsigned short *__restrict__ c_lrow, const double *__restrict__ prod,
                                                              const unsigned long long *__restrict__ rowbits, const int *__restrict__ rowpre,
                                                              double *__restrict__ y, double alpha, double beta, int dbg)
{
    extern __shared__ double tb_lds[];
    double *ys = tb_lds;                                           // kYTile doubles
    double *xbuf = tb_lds + kYTile;                                // three x tiles: cell i reads buffer i % 3
    const TileDesc T = tiles[blockIdx.x];
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kYTile; i += kTbThreads) ys[i] = 0.0;

    // ---- prologue: the first kDepth batches of the entry stream and the x tiles of the first two cells go out before the barrier
    constexpr int D = kDepth;
    uint2_t meta[D];
    double2_t val[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        meta[u] = uint2_t{0u, 0u}; val[u] = double2_t{0.0, 0.0};
        if (T.b0 + u < T.b1) {
            const BatchDesc d = batches[T.b0 + u];
            const long long p = ((long long)d.e0 + min(2 * tid, d.n - 2)) >> 1;
            meta[u] = tb_stream_load(h_meta + p);
            val[u] = tb_stream_load(reinterpret_cast<const double2_t *>(h_val) + p);
        }
    }
    double xr0 = 0.0, xr1 = 0.0, xr2 = 0.0;                        // the x tile of the NEXT cell on its way from L2 to LDS
    if (T.xt0 >= 0) {
        const double *src = hot_x + (size_t)T.xt0 * kXTile;
        xbuf[tid] = src[tid]; xbuf[tid + kTbThreads] = src[tid + kTbThreads]; xbuf[tid + 2 * kTbThreads] = src[tid + 2 * kTbThreads];
    }
    if (T.xt1 >= 0) {
        const double *src = hot_x + (size_t)T.xt1 * kXTile;
        xr0 = src[tid]; xr1 = src[tid + kTbThreads]; xr2 = src[tid + 2 * kTbThreads];
    }
    __syncthreads();

    // ---- cold products of this tile: (column band, tile) chunks of ~50 entries; 16 lanes per chunk, two chunks in flight per lane group
    if (!(dbg & 1)) {
        const int grp = tid >> 4, l = tid & 15;
        constexpr int G = kTbThreads / 16, CU = 2;
        for (int q = T.q0 + grp; q < T.q1; q += G * CU) {
            ColdChunk ch[CU];
            unsigned rr[CU][2]; double2_t pp[CU][2];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int qq = q + u * G;
                ch[u] = qq < T.q1 ? chunks[qq] : ColdChunk{0, 0};
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const long long p = ((long long)ch[u].e0 + min(2 * l + 32 * h, max(ch[u].n - 2, 0))) >> 1;
                    rr[u][h] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lrow) + p);
                    pp[u][h] = tb_stream_load(reinterpret_cast<const double2_t *>(prod) + p);
                }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (2 * l + 32 * h < ch[u].n) {
                        const unsigned r0 = rr[u][h] & 0xFFFFu, r1 = rr[u][h] >> 16;
                        double a = pp[u][h][0], b = pp[u][h][1];
                        if (r0 == r1) { b += a; a = 0.0; }
                        if (a != 0.0) atomicAdd(&ys[r0], a);
                        if (b != 0.0) atomicAdd(&ys[r1], b);
                    }
                for (int i = 2 * l + 64; i < ch[u].n; i += 32) {           // chunks longer than 64 entries (a cell just too sparse to be hot)
                    const long long p = ((long long)ch[u].e0 + i) >> 1;
                    const unsigned r = reinterpret_cast<const unsigned *>(c_lrow)[p];
                    const double2_t d = reinterpret_cast<const double2_t *>(prod)[p];
                    if (d[0] != 0.0) atomicAdd(&ys[r & 0xFFFFu], d[0]);
                    if (d[1] != 0.0) atomicAdd(&ys[r >> 16], d[1]);
                }
            }
        }
    }

    // ---- hot cells. The entry stream runs kDepth batches ahead in registers, regardless of cell boundaries. At the first batch of cell i
    // the x tile of cell i+1 (requested one cell earlier) moves from registers to buffer (i+1) % 3 — last read by cell i−2, which every
    // wave left before the previous barrier — then one barrier, then the x tile of cell i+2 is requested.
    int ci = -1;
    for (int b = T.b0; b < T.b1; b += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int bb = b + u;
            if (bb < T.b1) {
                const BatchDesc d = batches[bb];
                if (d.first || bb == T.b0) {
                    ++ci;
                    if (!(dbg & 4)) {
                        double *dst = xbuf + ((ci + 1) % 3) * kXTile;
                        dst[tid] = xr0; dst[tid + kTbThreads] = xr1; dst[tid + 2 * kTbThreads] = xr2;
                    }
                    if (!(dbg & 8)) __syncthreads();
                    if (d.xt2 >= 0 && !(dbg & 4)) {
                        const double *src = hot_x + (size_t)d.xt2 * kXTile;
                        xr0 = src[tid]; xr1 = src[tid + kTbThreads]; xr2 = src[tid + 2 * kTbThreads];
                    }
                }
                if (!(dbg & 2)) tb_accumulate_pair(2 * tid < d.n, meta[u][0], meta[u][1], val[u], xbuf + (ci % 3) * kXTile, ys);
                else if (val[u][0] == 1.2345e-300 && meta[u][0] == meta[u][1]) ys[0] = val[u][1];
                if (bb + D < T.b1) {
                    const BatchDesc dn = batches[bb + D];
                    const long long p = ((long long)dn.e0 + min(2 * tid, dn.n - 2)) >> 1;
                    meta[u] = tb_stream_load(h_meta + p);
                    val[u] = tb_stream_load(reinterpret_cast<const double2_t *>(h_val) + p);
                }
            }
        }
    }
    __syncthreads();

    // ---- the tile's range of natural rows: non-empty rows take their sum from LDS, empty rows get beta·y
    for (int r = T.r0 + tid; r < T.r1; r += kTbThreads) {
        const int w = r / kRowChunk;
        const unsigned long long bits = rowbits[w];
        const unsigned long long bit = 1ull << (r % kRowChunk);
        double s = 0.0;
        if (bits & bit) s = alpha * ys[rowpre[w] + __popcll(bits & (bit - 1ull)) - T.c0];
        if (T.split) { if (s != 0.0) atomicAdd(&y[r], s); }           // y was pre-scaled by beta (tb_prepare_kernel)
        else y[r] = beta == 0.0 ? s : s + beta * y[r];
    }
}

} // namespace

struct TbPlan {
    int rows = 0, cols = 0, NT = 0, XT = 0, CB = 0;
    long long nnz = 0, hot_entries = 0, cold_entries = 0, hot_padded = 0, cold_padded = 0;
    int n_cells = 0, n_batches = 0, n_chunks = 0, n_items = 0, n_work_items = 0, n_split_blocks = 0;
    TbBuf split_blocks, hot_cols, hot_x, h_meta, h_val, c_lcol, c_lrow, c_val, prod, tiles, batches, chunks, items, rowbits, rowpre;
    size_t lds_tile = 0, lds_cold = 0;
    int dbg = 0;   // G4S_TB_DBG: timing-only ablations (1 no cold reads, 2 no accumulation, 4 no x staging, 8 no cell barriers) — results are wrong
    long long bytes = 0;
};

int tb_build(TbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values)
{
    *out = nullptr;
    if (nnz <= 0 || rows <= 0 || cols <= 0) return set_error(G4S_ERR_INVALID, "tb_build: empty matrix");
    auto P = new (std::nothrow) TbPlan();
    if (!P) return set_error(G4S_ERR_NOMEM, "host allocation failed");
    std::unique_ptr<TbPlan> guard(P);
    P->rows = rows; P->cols = cols; P->nnz = nnz;
    auto env_int = [](const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; };

    // ---- 1. rows (host): non-empty rows → compact numbering, y tiles of <= kYTile compact rows with about equal work
    std::vector<int> h_rowptr((size_t)rows + 1);
    G4S_HIP_TRY(hipMemcpy(h_rowptr.data(), d_rowptr, sizeof(int) * h_rowptr.size(), hipMemcpyDeviceToHost));
    const int nchunks = (rows + kRowChunk - 1) / kRowChunk;
    std::vector<unsigned long long> h_bits((size_t)nchunks, 0ull);
    std::vector<int> h_pre((size_t)nchunks + 1, 0), h_tile_of_chunk((size_t)nchunks, 0);
    for (int w = 0; w < nchunks; ++w) {
        unsigned long long b = 0;
        const int r_end = std::min(rows, (w + 1) * kRowChunk);
        for (int r = w * kRowChunk; r < r_end; ++r)
            if (h_rowptr[r + 1] > h_rowptr[r]) b |= 1ull << (r - w * kRowChunk);
        h_bits[w] = b;
        h_pre[w + 1] = h_pre[w] + __builtin_popcountll(b);
    }
    const long long target_tiles = std::max<long long>(1, env_int("G4S_TB_TILES", 256));
    const long long quota = std::max<long long>(1, (nnz + target_tiles - 1) / target_tiles);
    const int ytile = (int)std::min<long long>(kYTile, std::max<long long>(kRowChunk, env_int("G4S_TB_YTILE", kYTile)));
    std::vector<int> tile_chunk0;                                   // first 64-row chunk of every tile
    {
        int comp = 0; long long work = 0;
        tile_chunk0.push_back(0);
        for (int w = 0; w < nchunks; ++w) {
            const int cnt = h_pre[w + 1] - h_pre[w];
            const int r_end = std::min(rows, (w + 1) * kRowChunk);
            const long long wk = (long long)h_rowptr[r_end] - h_rowptr[w * kRowChunk];
            if (w > tile_chunk0.back() && (comp + cnt > ytile || (work >= quota && comp > 0))) { tile_chunk0.push_back(w); comp = 0; work = 0; }
            comp += cnt; work += wk;
        }
    }
    const int NT = P->NT = (int)tile_chunk0.size();
    tile_chunk0.push_back(nchunks);
    std::vector<int> h_tile_c0((size_t)NT);
    for (int t = 0; t < NT; ++t) {
        h_tile_c0[t] = h_pre[tile_chunk0[t]];
        for (int w = tile_chunk0[t]; w < tile_chunk0[t + 1]; ++w) h_tile_of_chunk[w] = t;
    }
    TbBuf d_tile_of_chunk, d_tile_c0;
    G4S_TRY(P->rowbits.upload(h_bits));
    h_pre.pop_back();
    G4S_TRY(P->rowpre.upload(h_pre));
    G4S_TRY(d_tile_of_chunk.upload(h_tile_of_chunk));
    G4S_TRY(d_tile_c0.upload(h_tile_c0));

    // ---- 2. columns: rank by degree; the leading XT·4096 ranks are candidates for hot x tiles
    int XT = (int)std::min<long long>(kMaxXTiles, cols / kXTile);
    {
        const long long want = env_int("G4S_TB_XTILES", -1);
        if (want >= 0) XT = (int)std::min<long long>(XT, want);
    }
    TbBuf colmap, deg, deg_s, order_in, order, tmp0;
    G4S_TRY(colmap.alloc(sizeof(unsigned) * (size_t)cols));
    hipLaunchKernelGGL(tb_colmap_kernel, dim3((cols + 255) / 256), dim3(256), 0, nullptr, cols, colmap.as<unsigned>());
    if (XT > 0) {
        G4S_TRY(deg.alloc(sizeof(int) * (size_t)cols)); G4S_TRY(deg_s.alloc(sizeof(int) * (size_t)cols));
        G4S_TRY(order_in.alloc(sizeof(int) * (size_t)cols)); G4S_TRY(order.alloc(sizeof(int) * (size_t)cols));
        G4S_HIP_TRY(hipMemset(deg.p, 0, deg.bytes));
        hipLaunchKernelGGL(tb_col_degree_kernel, dim3(tb_grid(nnz)), dim3(256), 0, nullptr, nnz, d_colids, deg.as<int>());
        hipLaunchKernelGGL(tb_iota_kernel, dim3((cols + 255) / 256), dim3(256), 0, nullptr, cols, order_in.as<int>());
        G4S_HIP_TRY(hipGetLastError());
        size_t tb = 0;
        G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, deg.as<int>(), deg_s.as<int>(), order_in.as<int>(), order.as<int>(), cols, 0, 32, nullptr));
        G4S_TRY(tmp0.alloc(tb));
        G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp0.p, tb, deg.as<int>(), deg_s.as<int>(), order_in.as<int>(), order.as<int>(), cols, 0, 32, nullptr));
        // drop trailing x tiles whose columns are (nearly) unused: a tile is a candidate while it holds at least min_cell entries per y tile on average
        std::vector<int> top((size_t)XT * kXTile);
        G4S_HIP_TRY(hipMemcpy(top.data(), deg_s.p, sizeof(int) * top.size(), hipMemcpyDeviceToHost));
        const long long min_cell0 = std::max<long long>(1, env_int("G4S_TB_MIN_CELL", 256));
        int keep = 0;
        for (int j = 0; j < XT; ++j) {
            long long in_tile = 0;
            for (int i = 0; i < kXTile; ++i) in_tile += top[(size_t)j * kXTile + i];
            if (in_tile < min_cell0 * NT / 2) break;
            keep = j + 1;
        }
        if (env_int("G4S_TB_XTILES", -1) < 0) XT = keep;
    }
    P->XT = XT;
    const int nhot = XT * kXTile;
    if (XT > 0) {
        hipLaunchKernelGGL(tb_colmap_hot_kernel, dim3((nhot + 255) / 256), dim3(256), 0, nullptr, nhot, order.as<int>(), colmap.as<unsigned>());
        G4S_TRY(P->hot_cols.alloc(sizeof(int) * (size_t)nhot));
        G4S_TRY(P->hot_x.alloc(sizeof(double) * (size_t)nhot));
        G4S_HIP_TRY(hipMemcpy(P->hot_cols.p, order.p, sizeof(int) * (size_t)nhot, hipMemcpyDeviceToDevice));
    }
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipDeviceSynchronize());
    deg.release(); deg_s.release(); order_in.release(); order.release(); tmp0.release();

    // ---- 3. cells: count the entries of every candidate hot cell, then sort the entries by cell (hot cells first, then (column band, tile))
    const int CB = P->CB = (cols + kCBand - 1) >> kCBandBits;
    const long long n_hot_cells = (long long)NT * XT, n_cold_cells = (long long)CB * NT, ncells = n_hot_cells + n_cold_cells;
    if (ncells + 1 > (1ll << 31)) return set_error(G4S_ERR_UNSUPPORTED, "tb_build: too many cells (%lld)", ncells);
    int key_bits = 1;
    while ((1ll << key_bits) < ncells + 1) ++key_bits;
    const int min_cell = (int)std::max<long long>(1, env_int("G4S_TB_MIN_CELL", 256));
    TbBuf rowid, cell_count, key, key_s, idx, perm, tmp, start;
    const size_t n4 = sizeof(unsigned) * (size_t)nnz;
    G4S_TRY(rowid.alloc(n4));
    hipLaunchKernelGGL(tb_rowid_kernel, dim3(tb_grid(nnz)), dim3(256), 0, nullptr, rows, nnz, d_rowptr, rowid.as<int>());
    G4S_TRY(cell_count.alloc(sizeof(int) * (size_t)std::max<long long>(1, n_hot_cells)));
    G4S_HIP_TRY(hipMemset(cell_count.p, 0, cell_count.bytes));
    if (XT > 0)
        hipLaunchKernelGGL(tb_cell_count_kernel, dim3(tb_grid(nnz)), dim3(256), 0, nullptr, nnz, rowid.as<int>(), d_colids, colmap.as<unsigned>(), d_tile_of_chunk.as<int>(), XT,
                           cell_count.as<int>());
    G4S_TRY(key.alloc(n4)); G4S_TRY(key_s.alloc(n4)); G4S_TRY(idx.alloc(n4)); G4S_TRY(perm.alloc(n4));
    hipLaunchKernelGGL(tb_keys_kernel, dim3(tb_grid(nnz)), dim3(256), 0, nullptr, nnz, rowid.as<int>(), d_colids, colmap.as<unsigned>(), d_tile_of_chunk.as<int>(), XT, NT,
                       cell_count.as<int>(), min_cell, key.as<unsigned>(), idx.as<unsigned>());
    G4S_HIP_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key.as<unsigned>(), key_s.as<unsigned>(), idx.as<unsigned>(), perm.as<unsigned>(), (int)nnz, 0, key_bits, nullptr));
    G4S_TRY(tmp.alloc(tmp_bytes));
    G4S_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key.as<unsigned>(), key_s.as<unsigned>(), idx.as<unsigned>(), perm.as<unsigned>(), (int)nnz, 0, key_bits, nullptr));
    G4S_HIP_TRY(hipDeviceSynchronize());
    key.release(); idx.release(); tmp.release(); cell_count.release();
    G4S_TRY(start.alloc(sizeof(int) * (size_t)(ncells + 1)));
    hipLaunchKernelGGL(tb_cell_starts_kernel, dim3(tb_grid(ncells + 1)), dim3(256), 0, nullptr, nnz, key_s.as<unsigned>(), ncells, start.as<int>());
    G4S_HIP_TRY(hipGetLastError());
    std::vector<int> h_start((size_t)ncells + 1);
    G4S_HIP_TRY(hipMemcpy(h_start.data(), start.p, sizeof(int) * h_start.size(), hipMemcpyDeviceToHost));
    start.release();

    // ---- 4. padded layouts and descriptors (host)
    auto pad_up = [](long long v) { return (v + kPad - 1) & ~(long long)(kPad - 1); };
    std::vector<int> h_shift((size_t)ncells, 0);
    std::vector<TileDesc> tiles((size_t)NT);
    std::vector<BatchDesc> batches;
    std::vector<int> batch_xt, batch_xt1;                           // x tile of each batch's cell and of the following cell
    std::vector<long long> tile_work((size_t)NT, 0);
    long long hp = 0;
    int n_cells = 0;
    for (int t = 0; t < NT; ++t) {
        tiles[t] = TileDesc{};
        tiles[t].b0 = (int)batches.size();
        std::vector<int> xts;                                       // x tiles of this tile's hot cells, in order
        std::vector<int> first_batch;                               // index (into batches) of each cell's first batch
        for (int j = 0; j < XT; ++j) {
            const long long q = (long long)t * XT + j;
            const int n = h_start[q + 1] - h_start[q];
            h_shift[q] = (int)(hp - h_start[q]);
            if (n > 0) {
                const long long np = pad_up(n);
                first_batch.push_back((int)batches.size());
                xts.push_back(j);
                for (long long off = 0; off < np; off += kBatch)
                    batches.push_back(BatchDesc{(int)(hp + off), (int)std::min<long long>(kBatch, np - off), off == 0 ? 1 : 0, -1});
                hp += np;
                tile_work[t] += n;
                ++n_cells;
            }
        }
        first_batch.push_back((int)batches.size());
        for (size_t i = 0; i < xts.size(); ++i)
            for (int bb = first_batch[i]; bb < first_batch[i + 1]; ++bb) {
                batches[bb].xt2 = i + 2 < xts.size() ? xts[i + 2] : -1;          // every batch of a cell carries it: a split item may start mid-cell
                batch_xt.push_back(xts[i]);
                batch_xt1.push_back(i + 1 < xts.size() ? xts[i + 1] : -1);
            }
        tiles[t].b1 = (int)batches.size();
        tiles[t].xt0 = xts.size() > 0 ? xts[0] : -1;
        tiles[t].xt1 = xts.size() > 1 ? xts[1] : -1;
        tiles[t].r0 = tile_chunk0[t] * kRowChunk;
        tiles[t].r1 = std::min(rows, tile_chunk0[t + 1] * kRowChunk);
        tiles[t].c0 = h_tile_c0[t];
    }
    P->hot_entries = h_start[n_hot_cells];
    P->cold_entries = nnz - P->hot_entries;
    P->hot_padded = hp;
    if (hp + 64 > INT32_MAX) return set_error(G4S_ERR_UNSUPPORTED, "tb_build: padded hot length exceeds int32");
    // cold: (column band, tile) order; positions are relative to the cold arrays
    std::vector<std::vector<ColdChunk>> per_tile((size_t)NT);
    std::vector<ColdItem> items;
    long long cp = 0;
    for (int c = 0; c < CB; ++c) {
        const long long band0 = cp;
        for (int t = 0; t < NT; ++t) {
            const long long q = n_hot_cells + (long long)c * NT + t;
            const int n = h_start[q + 1] - h_start[q];
            h_shift[q] = (int)(cp - h_start[q]);
            if (n > 0) {
                per_tile[t].push_back(ColdChunk{(int)cp, (int)pad_up(n)});
                cp += pad_up(n);
                tile_work[t] += 2ll * n;
            }
        }
        for (long long e = band0; e < cp; e += kColdItem) items.push_back(ColdItem{c, (int)e, (int)std::min<long long>(cp, e + kColdItem), 0});
    }
    P->cold_padded = cp;
    if (cp + 64 > INT32_MAX) return set_error(G4S_ERR_UNSUPPORTED, "tb_build: padded cold length exceeds int32");
    std::vector<ColdChunk> chunks;
    for (int t = 0; t < NT; ++t) {
        tiles[t].q0 = (int)chunks.size();
        chunks.insert(chunks.end(), per_tile[t].begin(), per_tile[t].end());
        tiles[t].q1 = (int)chunks.size();
    }
    // A y tile whose work exceeds the cap (hub rows: the 64 rows that hold R-MAT's largest hubs carry 7 times the average tile) is shared by
    // several work items, each with its own LDS copy of the y tile; their sums meet in y through global atomics (tb_prepare_kernel
    // pre-scales those rows). Work = hot entries + 2 · cold entries; heaviest items first: the tail of the launch is made of light ones.
    long long total_work = 0;
    for (int t = 0; t < NT; ++t) total_work += tile_work[t];
    const long long cap = std::max<long long>(16384, env_int("G4S_TB_ITEM_CAP", total_work * 3 / (4 * 256)));
    std::vector<TileDesc> work_items;
    std::vector<long long> item_work;
    std::vector<int2> split_blocks;                                 // 256-row pieces of the split tiles' row ranges
    for (int t = 0; t < NT; ++t) {
        const TileDesc &T = tiles[t];
        const int k = (int)((tile_work[t] + cap - 1) / cap);
        if (k <= 1) { work_items.push_back(T); item_work.push_back(tile_work[t]); continue; }
        for (int r = T.r0; r < T.r1; r += 256) split_blocks.push_back(make_int2(r, std::min(T.r1, r + 256)));
        const long long share = (tile_work[t] + k - 1) / k;
        int b = T.b0, q = T.q0;
        while (b < T.b1 || q < T.q1) {
            TileDesc it = T;
            it.split = 1;
            it.b0 = b; it.q0 = q;
            long long w = 0;
            while (b < T.b1 && w < share) w += batches[b++].n;
            while (b == T.b1 && q < T.q1 && w < share) w += 2ll * chunks[q++].n;
            it.b1 = b; it.q1 = q;
            it.xt0 = it.b0 < it.b1 ? batch_xt[it.b0] : -1;
            it.xt1 = it.b0 < it.b1 ? batch_xt1[it.b0] : -1;
            work_items.push_back(it);
            item_work.push_back(w);
        }
    }
    const int NI = (int)work_items.size();
    std::vector<int> tord((size_t)NI);
    for (int t = 0; t < NI; ++t) tord[t] = t;
    std::stable_sort(tord.begin(), tord.end(), [&](int a, int b) { return item_work[a] > item_work[b]; });
    std::vector<TileDesc> tiles_sorted((size_t)NI);
    for (int t = 0; t < NI; ++t) tiles_sorted[t] = work_items[tord[t]];
    P->n_work_items = NI; P->n_split_blocks = (int)split_blocks.size();
    G4S_TRY(P->split_blocks.upload(split_blocks));
    std::stable_sort(items.begin(), items.end(), [](const ColdItem &a, const ColdItem &b) { return (a.e1 - a.e0) > (b.e1 - b.e0); });
    P->n_cells = n_cells; P->n_batches = (int)batches.size(); P->n_chunks = (int)chunks.size(); P->n_items = (int)items.size();
    if (batches.empty()) batches.push_back(BatchDesc{0, 0, 0, -1});
    if (chunks.empty()) chunks.push_back(ColdChunk{0, 0});
    G4S_TRY(P->tiles.upload(tiles_sorted)); G4S_TRY(P->batches.upload(batches)); G4S_TRY(P->chunks.upload(chunks)); G4S_TRY(P->items.upload(items));

    // ---- 5. fill
    TbBuf d_shift;
    G4S_TRY(d_shift.upload(h_shift));
    G4S_TRY(P->h_meta.alloc(sizeof(unsigned short) * 2 * (size_t)(hp + 64)));
    G4S_TRY(P->h_val.alloc(sizeof(double) * (size_t)(hp + 64)));
    G4S_TRY(P->c_lcol.alloc(sizeof(unsigned short) * (size_t)(cp + 64))); G4S_TRY(P->c_lrow.alloc(sizeof(unsigned short) * (size_t)(cp + 64)));
    G4S_TRY(P->c_val.alloc(sizeof(double) * (size_t)(cp + 64))); G4S_TRY(P->prod.alloc(sizeof(double) * (size_t)(cp + 64)));
    G4S_HIP_TRY(hipMemset(P->h_meta.p, 0xFF, P->h_meta.bytes));    // pad flag (and local row 0xFFFF) everywhere; real entries overwrite it. A pad's
    G4S_HIP_TRY(hipMemset(P->c_lcol.p, 0xFF, P->c_lcol.bytes));    // product is forced to 0 and zero sums are never added, so its row is never used
    G4S_HIP_TRY(hipMemset(P->c_lrow.p, 0, P->c_lrow.bytes));
    G4S_HIP_TRY(hipMemset(P->h_val.p, 0, P->h_val.bytes)); G4S_HIP_TRY(hipMemset(P->c_val.p, 0, P->c_val.bytes));
    G4S_HIP_TRY(hipMemset(P->prod.p, 0, P->prod.bytes));
    // cold positions: the sorted index i of a cold entry counts the hot entries in front of it; shift already folds that in (cp starts at 0)
    hipLaunchKernelGGL(tb_fill_kernel, dim3(tb_grid(nnz)), dim3(256), 0, nullptr, nnz, key_s.as<unsigned>(), perm.as<unsigned>(), rowid.as<int>(), d_colids, d_values,
                       colmap.as<unsigned>(), P->rowbits.as<unsigned long long>(), P->rowpre.as<int>(), d_tile_c0.as<int>(), XT, NT, d_shift.as<int>(),
                       P->h_meta.as<unsigned short>(), P->h_val.as<double>(),
                       P->c_lcol.as<unsigned short>(), P->c_lrow.as<unsigned short>(), P->c_val.as<double>());
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipDeviceSynchronize());

    P->dbg = (int)env_int("G4S_TB_DBG", 0);
    P->lds_tile = sizeof(double) * (kYTile + 3 * kXTile);
    P->lds_cold = sizeof(double) * kCBand;
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(tb_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_tile));
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(tb_cold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_cold));
    P->bytes = (long long)(P->hot_cols.bytes + P->hot_x.bytes + P->h_meta.bytes + P->h_val.bytes + P->c_lcol.bytes + P->c_lrow.bytes + P->c_val.bytes +
                           P->prod.bytes + P->tiles.bytes + P->batches.bytes + P->chunks.bytes + P->items.bytes + P->rowbits.bytes + P->rowpre.bytes);
    if (getenv("G4S_DEBUG"))
        fprintf(stderr, "g4s tile-blocked SpMV plan: %d y tiles (<= %d compact rows), %d x tiles, %d column bands; nnz %lld: hot %lld in %d cells / %d batches (padded %lld), cold %lld in %d chunks "
                        "(padded %lld), %d cold items; %d tile work items (cap %lld), %.2f GB\n",
                NT, ytile, XT, CB, nnz, P->hot_entries, P->n_cells, P->n_batches, hp, P->cold_entries, P->n_chunks, cp, P->n_items, P->n_work_items, cap, P->bytes / 1e9);
    *out = guard.release();
    return G4S_OK;
}

void tb_destroy(TbPlan *P) { delete P; }

long long tb_bytes(const TbPlan *P) { return P ? P->bytes : 0; }

int tb_spmv(TbPlan *P, const double *x, double *y, double alpha, double beta, hipStream_t s)
{
    const int nhot = P->XT * kXTile;
    if (nhot || P->n_split_blocks)
        hipLaunchKernelGGL(tb_prepare_kernel, dim3(P->n_split_blocks + (nhot + 255) / 256), dim3(256), 0, s, P->n_split_blocks, P->split_blocks.as<int2>(), y, beta, nhot,
                           P->hot_cols.as<int>(), x, P->hot_x.as<double>());
    if (P->n_items)
        hipLaunchKernelGGL(tb_cold_kernel, dim3(P->n_items), dim3(kTbThreads), P->lds_cold, s, P->items.as<ColdItem>(), P->cols, P->c_lcol.as<unsigned short>(), P->c_val.as<double>(),
                           x, P->prod.as<double>());
    hipLaunchKernelGGL(tb_tile_kernel, dim3(P->n_work_items), dim3(kTbThreads), P->lds_tile, s, P->tiles.as<TileDesc>(), P->batches.as<BatchDesc>(), P->chunks.as<ColdChunk>(),
                       P->h_meta.as<uint2_t>(), P->h_val.as<double>(), P->hot_x.as<double>(), P->c_lrow.as<unsigned short>(),
                       P->prod.as<double>(), P->rowbits.as<unsigned long long>(), P->rowpre.as<int>(), y, alpha, beta, P->dbg);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

} // namespace g4s
