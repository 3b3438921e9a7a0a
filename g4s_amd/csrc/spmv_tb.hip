// spmv_tb.hip — tile-blocked fp64 SpMV for matrices whose x gathers have no locality (power-law graphs). Round-2 EXPERIMENT next to the
// propagation-blocked path (spmv_pb.hip, the default): opt in with G4S_SPMV_IMPL=tb. Parity-green on every SpMV test; on configs[1] it reaches
// 0.43–0.45 ms against 0.40 ms for the default (first version 0.61–0.66 ms) — DESIGN.md §4.1 has the byte model and the measurements.
//
// Idea: a product needs x[col] and y[row] in the SAME LDS, or the product has to travel through HBM (18 B per partial sum in spmv_pb).
//   * rows: the non-empty rows (41 % of configs[1]) are numbered consecutively and cut into y tiles of at most 8 192 rows with about equal
//     work; a tile's slice of y lives in LDS (64 KiB) and is written to HBM once, together with the zeros of the empty rows of its range.
//   * columns: ranked by degree; the leading ranks form x tiles of kXTile columns. hot_x = x in rank order is gathered at the start of
//     every product (a few MB, L2-resident afterwards).
//   * a cell (y tile, x tile) with at least G4S_TB_MIN_CELL entries is HOT: its entries are stored tile-major / x-tile-major as (local column
//     u16, local row u16, value f64) = 12 B; tb_tile_kernel multiplies them against the x tile in LDS and adds into the y tile with LDS atomics
//     after a 32-entry DPP segmented scan — no partial sum leaves the CU. x tiles arrive by LDS-DMA (global_load_lds_dwordx4, issued from an asm
//     statement so that hipcc does not fence every LDS atomic behind it) two cells ahead into a ring of three buffers; the entry stream runs kDepth
//     batches ahead in registers through a branch-free, fully padded batch list (every stream load unconditional and clamped, so hipcc's vmcnt
//     counting is exact); cell transitions wait only for the x tile they need (counted vmcnt) and use a bare s_barrier.
//   * all other entries are COLD and take the propagation route without merging: tb_cold_kernel walks them column band by column band (16 K
//     natural columns of x in LDS) and scatters the products, 64 bytes per 8-entry span, into tile-major order; the tile kernel reads its tile's
//     products back as one contiguous stream. 28 B per cold entry.
//   * work items (a tile, or a run of cells / cold chunks of a heavy tile: the 64 rows holding R-MAT's largest hubs carry 7× the average) are
//     pulled heaviest-first from a counter by one persistent 1024-thread workgroup per CU.
// What bounds it (stamps, G4S_TB_DBG=64): every hot batch costs one dependent chain of ≈ 1 µs (descriptor → stream data → LDS gather → DPP scan
// → LDS atomic) with all sixteen waves of the workgroup in the same phase, and configs[1] has 48 700 hot cells of 1 600 entries on average —
// ≈ 70 000 batch steps, 57 % full — so the hot loop runs at ≈ 14 GB/s per CU whatever the prefetch depth. Fewer, larger cells (a higher
// threshold) move entries to the 28-byte route faster than they remove steps.
// fp64 sums are accumulated by LDS atomics: within the 1e-10 tolerance of the oracle, not bit for bit, last bits may differ from run to run.
#include "common.hpp"
#include "spmv_pb.hpp"
#include <hipcub/hipcub.hpp>
#include <functional>
#include <algorithm>
#include <memory>
#include <vector>

namespace g4s {

namespace {

#ifndef G4S_TB_XTILE
#define G4S_TB_XTILE 3072
#endif
constexpr int kXTile = G4S_TB_XTILE;          // columns per hot x tile: 24 KiB of fp64, three of them in LDS (filled by LDS-DMA, two cells ahead)
constexpr int kXBufs = 3;
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
constexpr int kMaxXTiles = 256;              // at most 786 K ranked columns
constexpr int kColdDepth = 4;                // cold pieces in flight per wave
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

// One work item of the tile kernel: a y tile (natural rows [r0,r1), first compact row c0) with a run of its hot cells — batches [b0,b1),
// a multiple of kDepth long, padded with empty batches — and its cold products [k0,k1) (entries of the tile-major product array). split: the tile's work
// is shared by several items (their sums meet in y through atomics). xt0, xt1: x tiles of the item's first two cells (−1: none).
struct TileDesc { int b0, b1, k0, k1, r0, r1, c0, xt0, split, xt1, pad1, pad2; };
// Hot entries [e0, e0+n) of ONE cell (n <= kBatch, a multiple of kPad; 0 = padding). flags: bit 0 first batch of its cell (a transition: the x
// tile of this cell must have landed, all waves leave the previous cell, the x tile of the cell after next is requested), bits 1-2 the LDS x
// buffer of this cell (cell index mod 3), bits 4-7 min(batches of the previous two cells, kDepth + 1) = how many younger stream batches may stay in
// flight at the transition. xt_next: x tile of the item's cell after next (−1: none), valid on a first batch.
struct BatchDesc { int e0, n, flags, xt_next; };
constexpr int kMaxItemBatches = 1024;                                // batch descriptors of one work item staged in LDS (16 KiB): see tb_tile_kernel
struct ColdPiece { int e0, n; };                                    // cold entries [e0, e0+n), n <= kPiece (0 = padding), of one (column band, tile) chunk
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
                               int XT, int NT, const int *__restrict__ shift, const int *__restrict__ shift_cons,
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
            c_val[pos] = values[k];
            c_lrow[i + shift_cons[q]] = (unsigned short)lrow;     // the consumer reads its tile's cold products as one stream: tile-major positions
        }
    }
}

// ================================================================================================ SpMV kernels
// One launch ahead of the products: blocks [0, n_split) pre-scale y for the rows of split y tiles (their work items add into y with
// atomics), the remaining blocks gather x in rank order into hot_x.
__global__ void tb_prepare_kernel(int *__restrict__ item_counter, int n_split, const int2 *__restrict__ split_blocks, double *__restrict__ y, double beta,
                                  int nhot, const int *__restrict__ hot_cols, const double *__restrict__ x, double *__restrict__ hot_x)
{
    const int b = blockIdx.x;
    if (b == 0 && threadIdx.x == 0) *item_counter = 0;             // the tile kernel's work queue
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
                                                              const double *__restrict__ c_val, const int *__restrict__ span_dst, const double *__restrict__ x,
                                                              double *__restrict__ prod)
{
    extern __shared__ double tb_lds[];
    double *xs = tb_lds;                                           // kCBand doubles
    const ColdItem it = items[blockIdx.x];
    const int c0 = it.cband << kCBandBits;
    constexpr int U = 4;
    const long long p_end = it.e1 / 2, p_last = p_end - 1;         // pair indices
    long long base = it.e0 / 2 + (int)threadIdx.x;
    unsigned lc[U], lc_n[U];
    int ds[U], ds_n[U];
    double2_t v[U], v_n[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long p = min(base + (long long)u * kTbThreads, p_last);
        lc[u] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lcol) + p);
        v[u] = tb_stream_load(reinterpret_cast<const double2_t *>(c_val) + p);
        ds[u] = span_dst[p >> 2];
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
                ds_n[u] = span_dst[p >> 2];
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
                *reinterpret_cast<double2_t *>(prod + ds[u] + 2 * (int)(p & 3)) = o;     // the span's 64 bytes in the consumer's (tile-major) order
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) { lc[u] = lc_n[u]; v[u] = v_n[u]; ds[u] = ds_n[u]; }
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
    const double p0 = (!live || l0 == 0xFFFFu) ? 0.0 : v[0] * xb[l0 == 0xFFFFu ? 0u : l0];     // 0xFFFF = pad slot
    const double p1 = (!live || l1 == 0xFFFFu) ? 0.0 : v[1] * xb[l1 == 0xFFFFu ? 0u : l1];
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

// LDS-DMA: a wave instruction moves 64 × 16 bytes from per-lane global addresses to 1 KiB of LDS at M0 — no VGPR, nothing to wait for until
// the data is needed. The x tile of the NEXT cell is requested one cell ahead into the buffer the previous cell has just left. Written as an
// asm statement on purpose (cdna_hip_programming.md §5.7): issued through the builtin, hipcc has to assume the LDS write may alias the y tile
// and puts an `s_waitcnt vmcnt(0)` in front of every LDS atomic that follows — draining the entry stream each time. M0 is saved and restored
// inside the statement; the landing is waited for explicitly at the next cell transition.
__device__ __forceinline__ void tb_dma_x_tile(const double *__restrict__ src, double *lds_dst, int wave, int lane)
{
    constexpr int kChunks = kXTile * 8 / 1024;                      // 1 KiB pieces of a tile
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_dst;
    for (int k = wave; k < kChunks; k += kTbThreads / 64) {
        const double *g = src + k * 128 + lane * 2;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)k * 1024u));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
}

// Persistent: one 1024-thread workgroup per CU pulls work items (heaviest first) from a counter — a fresh workgroup of this size with
// 144 KiB of LDS costs ≈ 10 µs to start, three rounds of them were a quarter of the kernel.
__global__ __launch_bounds__(kTbThreads) void tb_tile_kernel(const TileDesc *__restrict__ tiles, int n_items, int *__restrict__ item_counter,
                                                              const BatchDesc *__restrict__ batches,
                                                              const uint2_t *__restrict__ h_meta, const double *__restrict__ h_val, const double *__restrict__ hot_x,
                                                              const unsigned short *__restrict__ c_lrow, const double *__restrict__ prod,
                                                              const unsigned long long *__restrict__ rowbits, const int *__restrict__ rowpre,
                                                              double *__restrict__ y, double alpha, double beta, int dbg, unsigned long long *__restrict__ stamps)
{
    extern __shared__ double tb_lds[];
    double *ys = tb_lds;                                           // kYTile doubles
    double *xbuf = tb_lds + kYTile;                                // three x tiles: cell i of the item reads buffer i % 3
    int *slot = reinterpret_cast<int *>(tb_lds + kYTile + kXBufs * kXTile);
    // The item's batch descriptors, copied to LDS once per item. Read from global memory they were scalar loads (the index is uniform), and
    // hipcc waits for a scalar load where it is first used: two exposed L2 round trips per batch step (lgkmcnt is shared with LDS, so a
    // scalar load issued ahead would stall the next LDS wait instead). Measured: no change in the step time — see DESIGN.md §7.
    int4 *bd = reinterpret_cast<int4 *>(tb_lds + kYTile + kXBufs * kXTile + 2);
    auto desc = [&](int i) { const int4 v = bd[i]; return BatchDesc{v.x, v.y, v.z, v.w}; };
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // diagnostic build only (G4S_TB_DBG & 64): shader-clock stamps at the section boundaries, written to a buffer nothing else reads
#define G4S_TB_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(size_t)item * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    for (;;) {
        if (tid == 0) *slot = atomicAdd(item_counter, 1);
        __syncthreads();
        const int item = *slot;
        if (item >= n_items) break;
        const TileDesc T = tiles[item];
        G4S_TB_STAMP(0);
        for (int i = tid; i < kYTile; i += kTbThreads) ys[i] = 0.0;
        const int nbd = T.b1 - T.b0 + 2 * kDepth;                  // <= kMaxItemBatches (tb_build); the list is padded past its end
        for (int i = tid; i < nbd; i += kTbThreads) bd[i] = reinterpret_cast<const int4 *>(batches)[T.b0 + i];
        __syncthreads();

        // ---- prologue: the first kDepth batches of the entry stream (the lists are padded, so there is always a descriptor) and the x tile of
        // the first cell. Every stream load is unconditional and clamped: the compiler counts them exactly and leaves the younger ones in flight.
        constexpr int D = kDepth;
        uint2_t meta[D];
        double2_t val[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const BatchDesc d = desc(u);
            const long long p = ((long long)d.e0 + min(2 * tid, max(d.n - 2, 0))) >> 1;
            meta[u] = tb_stream_load(h_meta + p);
            val[u] = tb_stream_load(reinterpret_cast<const double2_t *>(h_val) + p);
        }
        if (T.xt0 >= 0) tb_dma_x_tile(hot_x + (size_t)T.xt0 * kXTile, xbuf, wave, lane);
        if (T.xt1 >= 0) tb_dma_x_tile(hot_x + (size_t)T.xt1 * kXTile, xbuf + kXTile, wave, lane);
        __syncthreads();                                           // ys is zero
        G4S_TB_STAMP(1);

        // ---- cold products of this tile: one contiguous stream in the consumer's order (tb_cold_kernel scattered them here), a pair per lane,
        // kColdDepth batches in flight, no descriptors
        if (!(dbg & 1)) {
            constexpr int DC = kColdDepth;
            unsigned rr[DC];
            double2_t pp[DC];
            const int k_last = max(T.k1 - 2, 0);
#pragma unroll
            for (int u = 0; u < DC; ++u) {
                const long long p = (long long)min(T.k0 + u * kBatch + 2 * tid, k_last) >> 1;
                rr[u] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lrow) + p);
                pp[u] = tb_stream_load(reinterpret_cast<const double2_t *>(prod) + p);
            }
            for (int base = T.k0; base < T.k1; base += kBatch * DC) {
#pragma unroll
                for (int u = 0; u < DC; ++u) {
                    const int e = base + u * kBatch + 2 * tid;
                    if (e < T.k1) {
                        const unsigned r0 = rr[u] & 0xFFFFu, r1 = rr[u] >> 16;
                        double a = pp[u][0], b = pp[u][1];
                        if (r0 == r1) { b += a; a = 0.0; }
                        if (a != 0.0) atomicAdd(&ys[r0], a);
                        if (b != 0.0) atomicAdd(&ys[r1], b);
                    }
                    const long long p = (long long)min(e + kBatch * DC, k_last) >> 1;
                    rr[u] = tb_stream_load(reinterpret_cast<const unsigned *>(c_lrow) + p);
                    pp[u] = tb_stream_load(reinterpret_cast<const double2_t *>(prod) + p);
                }
            }
        }
        G4S_TB_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the first x tile has landed (and the prologue batches with it)
        __syncthreads();
        G4S_TB_STAMP(3);

        // ---- hot cells. At the first batch of a cell: wait until the x tile requested one cell ago has landed — only the stream batches issued
        // after that request may stay in flight (in-order vmcnt) — then one barrier (every wave has left the previous cell, whose buffer is the
        // next DMA's target), then the request for the next cell's x tile.
        for (int b = T.b0; b < ((dbg & 32) ? T.b0 : T.b1); b += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const BatchDesc d = desc(b - T.b0 + u);
                const int buf = (d.flags >> 1) & 3;
                if (d.flags & 1) {
                    // the tile of THIS cell was requested two cells ago; the request for the next cell's tile (one cell ago, >= 1 hidden DMA
                    // instruction per wave) and `keep` stream batches are younger and may stay in flight
                    const int keep = (d.flags >> 4) & 15;
                    switch (keep) {                                  // keep == kDepth + 1: no wait (see above)
                    case 0: case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                    case 5: if (kDepth >= 5) asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                    case 6: if (kDepth >= 6) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
                    case 7: if (kDepth >= 7) asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
                    case 8: if (kDepth >= 8) asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
                    default: break;
                    }
                    // keep == kDepth + 1: the request is older than the batch the previous iteration already waited for
                    if (!(dbg & 8)) __builtin_amdgcn_s_barrier();  // no fence: the x buffers are only read, the y tile only takes atomics
                    if (d.xt_next >= 0 && !(dbg & 4)) tb_dma_x_tile(hot_x + (size_t)d.xt_next * kXTile, xbuf + ((buf + 2) % kXBufs) * kXTile, wave, lane);
                }
                if (!(dbg & 2)) tb_accumulate_pair(2 * tid < d.n, meta[u][0], meta[u][1], val[u], xbuf + buf * kXTile, ys);
                const BatchDesc dn = desc(b - T.b0 + u + D);
                const long long p = ((long long)dn.e0 + min(2 * tid, max(dn.n - 2, 0))) >> 1;
                meta[u] = tb_stream_load(h_meta + p);
                val[u] = tb_stream_load(reinterpret_cast<const double2_t *>(h_val) + p);
            }
        }
        __syncthreads();
        G4S_TB_STAMP(4);

        // ---- the tile's range of natural rows: non-empty rows take their sum from LDS, empty rows get beta·y; four rows per lane in flight
        for (int r = T.r0 + tid; r < T.r1; r += 4 * kTbThreads) {
            unsigned long long bits[4];
            int pre[4];
            double yo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rj = min(r + j * kTbThreads, T.r1 - 1);
                bits[j] = rowbits[rj / kRowChunk];
                pre[j] = rowpre[rj / kRowChunk];
                yo[j] = (beta != 0.0 && !T.split) ? y[rj] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rj = r + j * kTbThreads;
                if (rj < T.r1) {
                    const unsigned long long bit = 1ull << (rj % kRowChunk);
                    double s = 0.0;
                    if (bits[j] & bit) s = alpha * ys[pre[j] + __popcll(bits[j] & (bit - 1ull)) - T.c0];
                    if (T.split) { if (s != 0.0) atomicAdd(&y[rj], s); }   // y was pre-scaled by beta (tb_prepare_kernel)
                    else y[rj] = beta == 0.0 ? s : s + beta * yo[j];
                }
            }
        }
        G4S_TB_STAMP(5);
        __syncthreads();                                           // every lane has read its sums before the next item zeroes the tile
    }
#undef G4S_TB_STAMP
}

} // namespace

struct TbPlan {
    int rows = 0, cols = 0, NT = 0, XT = 0, CB = 0;
    long long nnz = 0, hot_entries = 0, cold_entries = 0, hot_padded = 0, cold_padded = 0;
    int n_cells = 0, n_batches = 0, n_chunks = 0, n_items = 0, n_work_items = 0, n_split_blocks = 0;
    TbBuf stamps, span_dst, counter, split_blocks, hot_cols, hot_x, h_meta, h_val, c_lcol, c_lrow, c_val, prod, tiles, batches, items, rowbits, rowpre;
    size_t lds_tile = 0, lds_cold = 0;
    int n_cus = 256;
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
    struct HotCell { int xt, e0, n; };
    std::vector<std::vector<HotCell>> tcells((size_t)NT);
    std::vector<long long> tile_work((size_t)NT, 0);
    long long hp = 0;
    int n_cells = 0;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < XT; ++j) {
            const long long q = (long long)t * XT + j;
            const int n = h_start[q + 1] - h_start[q];
            h_shift[q] = (int)(hp - h_start[q]);
            if (n > 0) {
                tcells[t].push_back(HotCell{j, (int)hp, (int)pad_up(n)});
                hp += pad_up(n);
                tile_work[t] += n;
                ++n_cells;
            }
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
    // consumer order of the cold products: tile-major, column band inside the tile (the same padded chunks, renumbered). span_dst[s] = consumer
    // position of producer span s (8 entries = 64 bytes of products); shift_cons[q] maps a sorted cold entry to its consumer position.
    std::vector<int> h_shift_cons((size_t)ncells, 0), h_span_dst((size_t)(cp / kPad) + 1, 0);
    std::vector<std::vector<int>> cons_off((size_t)NT);             // consumer offset of every chunk of a tile, in per_tile order (+ the tile's end)
    {
        long long kc = 0;
        std::vector<size_t> next_chunk((size_t)NT, 0);
        for (int t = 0; t < NT; ++t) {
            cons_off[t].reserve(per_tile[t].size() + 1);
            for (const ColdChunk &ch : per_tile[t]) { cons_off[t].push_back((int)kc); kc += ch.n; }
            cons_off[t].push_back((int)kc);
        }
        for (int c = 0; c < CB; ++c)
            for (int t = 0; t < NT; ++t) {
                const long long q = n_hot_cells + (long long)c * NT + t;
                const int n = h_start[q + 1] - h_start[q];
                if (n <= 0) continue;
                const size_t k = next_chunk[t]++;                  // chunks were appended to per_tile[t] in ascending column band
                const int dst = cons_off[t][k];
                h_shift_cons[q] = dst - h_start[q];
                for (int e = 0; e < per_tile[t][k].n; e += kPad) h_span_dst[(size_t)(per_tile[t][k].e0 + e) / kPad] = dst + e;
            }
    }
    // Work items. A y tile whose work exceeds the cap (hub rows: the 64 rows that hold R-MAT's largest hubs carry 7 times the average tile) is
    // shared by several items — whole cells, then whole cold chunks — each with its own LDS copy of the y tile; their sums meet in y through
    // global atomics (tb_prepare_kernel pre-scales those rows). Work = hot entries + 2 · cold entries; heaviest items first.
    long long total_work = 0;
    for (int t = 0; t < NT; ++t) total_work += tile_work[t];
    const long long cap = std::max<long long>(16384, env_int("G4S_TB_ITEM_CAP", total_work * 3 / (4 * 256)));
    std::vector<TileDesc> work_items;
    std::vector<long long> item_work;
    std::vector<BatchDesc> batches;
    std::vector<int2> split_blocks;                                 // 256-row pieces of the split tiles' row ranges
    auto emit_item = [&](int t, size_t c0, size_t c1, size_t k0, size_t k1, int split, long long w) {
        TileDesc it{};
        it.r0 = tile_chunk0[t] * kRowChunk;
        it.r1 = std::min(rows, tile_chunk0[t + 1] * kRowChunk);
        it.c0 = h_tile_c0[t];
        it.split = split;
        it.xt0 = c1 > c0 ? tcells[t][c0].xt : -1;
        it.xt1 = c1 > c0 + 1 ? tcells[t][c0 + 1].xt : -1;
        it.b0 = (int)batches.size();
        int prev_nb = kDepth + 1, prev2_nb = kDepth + 1;                    // batches of the previous two cells (the item's first cells wait in the prologue)
        for (size_t c = c0; c < c1; ++c) {
            const HotCell &hc = tcells[t][c];
            const int nb = (hc.n + kBatch - 1) / kBatch, buf = (int)((c - c0) % kXBufs);
            for (int b = 0; b < nb; ++b) {
                int flags = buf << 1;
                if (b == 0) flags |= 1 | (std::min(prev_nb + prev2_nb, kDepth + 1) << 4);
                batches.push_back(BatchDesc{hc.e0 + b * kBatch, std::min(kBatch, hc.n - b * kBatch), flags, b == 0 && c + 2 < c1 ? tcells[t][c + 2].xt : -1});
            }
            prev2_nb = prev_nb;
            prev_nb = nb;
        }
        while ((batches.size() - (size_t)it.b0) % kDepth) batches.push_back(BatchDesc{0, 0, 0, -1});
        it.b1 = (int)batches.size();
        it.k0 = cons_off[t][k0];
        it.k1 = cons_off[t][k1];
        work_items.push_back(it);
        item_work.push_back(w);
    };
    int n_chunks_total = 0;
    for (int t = 0; t < NT; ++t) {
        n_chunks_total += (int)per_tile[t].size();
        // an item's batch descriptors live in LDS (kMaxItemBatches, less the two groups the stream runs ahead): heavy tiles are cut by that too
        constexpr int kBatchLimit = kMaxItemBatches - 2 * kDepth - kDepth;
        long long tile_batches = 0;
        for (const HotCell &hc : tcells[t]) {
            const long long nb = (hc.n + kBatch - 1) / kBatch;
            if (nb > kBatchLimit) return set_error(G4S_ERR_UNSUPPORTED, "tb_build: a cell of %d entries exceeds the per-item batch list", hc.n);
            tile_batches += nb;
        }
        const int k = (int)std::max<long long>((tile_work[t] + cap - 1) / cap, (tile_batches + kBatchLimit - 1) / kBatchLimit);
        if (k <= 1) { emit_item(t, 0, tcells[t].size(), 0, per_tile[t].size(), 0, tile_work[t]); continue; }
        const int r0 = tile_chunk0[t] * kRowChunk, r1 = std::min(rows, tile_chunk0[t + 1] * kRowChunk);
        for (int r = r0; r < r1; r += 256) split_blocks.push_back(make_int2(r, std::min(r1, r + 256)));
        const long long share = (tile_work[t] + k - 1) / k;
        size_t c = 0, q = 0;
        while (c < tcells[t].size() || q < per_tile[t].size()) {
            const size_t c_begin = c, q_begin = q;
            long long w = 0, nbat = 0;
            while (c < tcells[t].size() && w < share && nbat + (tcells[t][c].n + kBatch - 1) / kBatch <= kBatchLimit) {
                nbat += (tcells[t][c].n + kBatch - 1) / kBatch;
                w += tcells[t][c++].n;
            }
            while (c == tcells[t].size() && q < per_tile[t].size() && w < share) w += 2ll * per_tile[t][q++].n;
            emit_item(t, c_begin, c, q_begin, q, 1, w);
        }
    }
    for (int i = 0; i < 2 * kDepth; ++i) batches.push_back(BatchDesc{0, 0, 0, -1});             // the stream runs kDepth batches past an item's end; its LDS copy 2·kDepth
    const int NI = (int)work_items.size();
    std::vector<int> tord((size_t)NI);
    for (int t = 0; t < NI; ++t) tord[t] = t;
    std::stable_sort(tord.begin(), tord.end(), [&](int a, int b) { return item_work[a] > item_work[b]; });
    std::vector<TileDesc> tiles_sorted((size_t)NI);
    for (int t = 0; t < NI; ++t) tiles_sorted[t] = work_items[tord[t]];
    P->n_work_items = NI; P->n_split_blocks = (int)split_blocks.size();
    G4S_TRY(P->split_blocks.upload(split_blocks));
    std::stable_sort(items.begin(), items.end(), [](const ColdItem &a, const ColdItem &b) { return (a.e1 - a.e0) > (b.e1 - b.e0); });
    P->n_cells = n_cells; P->n_batches = (int)batches.size(); P->n_chunks = n_chunks_total; P->n_items = (int)items.size();
    G4S_TRY(P->tiles.upload(tiles_sorted)); G4S_TRY(P->batches.upload(batches)); G4S_TRY(P->span_dst.upload(h_span_dst)); G4S_TRY(P->items.upload(items));

    // ---- 5. fill
    TbBuf d_shift, d_shift_cons;
    G4S_TRY(d_shift.upload(h_shift));
    G4S_TRY(d_shift_cons.upload(h_shift_cons));
    G4S_TRY(P->counter.alloc(sizeof(int)));
    G4S_HIP_TRY(hipMemset(P->counter.p, 0, sizeof(int)));
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
                       colmap.as<unsigned>(), P->rowbits.as<unsigned long long>(), P->rowpre.as<int>(), d_tile_c0.as<int>(), XT, NT, d_shift.as<int>(), d_shift_cons.as<int>(),
                       P->h_meta.as<unsigned short>(), P->h_val.as<double>(),
                       P->c_lcol.as<unsigned short>(), P->c_lrow.as<unsigned short>(), P->c_val.as<double>());
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(hipDeviceSynchronize());

    P->dbg = (int)env_int("G4S_TB_DBG", 0);
    { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) P->n_cus = prop.multiProcessorCount; }
    if (P->dbg & 64) G4S_TRY(P->stamps.alloc(sizeof(unsigned long long) * 8 * (size_t)P->n_work_items));
    P->lds_tile = sizeof(double) * (kYTile + kXBufs * kXTile) + 16 + sizeof(int4) * kMaxItemBatches;   // + the work-queue slot + the item's batch descriptors
    P->lds_cold = sizeof(double) * kCBand;
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(tb_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_tile));
    G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(tb_cold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)P->lds_cold));
    P->bytes = (long long)(P->hot_cols.bytes + P->hot_x.bytes + P->h_meta.bytes + P->h_val.bytes + P->c_lcol.bytes + P->c_lrow.bytes + P->c_val.bytes +
                           P->prod.bytes + P->tiles.bytes + P->batches.bytes + P->span_dst.bytes + P->items.bytes + P->rowbits.bytes + P->rowpre.bytes);
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
    hipLaunchKernelGGL(tb_prepare_kernel, dim3(std::max(1, P->n_split_blocks + (nhot + 255) / 256)), dim3(256), 0, s, P->counter.as<int>(), P->n_split_blocks,
                       P->split_blocks.as<int2>(), y, beta, nhot, P->hot_cols.as<int>(), x, P->hot_x.as<double>());
    if (P->n_items)
        hipLaunchKernelGGL(tb_cold_kernel, dim3(P->n_items), dim3(kTbThreads), P->lds_cold, s, P->items.as<ColdItem>(), P->cols, P->c_lcol.as<unsigned short>(), P->c_val.as<double>(),
                           P->span_dst.as<int>(), x, P->prod.as<double>());
    hipLaunchKernelGGL(tb_tile_kernel, dim3(std::min(P->n_work_items, P->n_cus)), dim3(kTbThreads), P->lds_tile, s, P->tiles.as<TileDesc>(), P->n_work_items, P->counter.as<int>(),
                       P->batches.as<BatchDesc>(), P->h_meta.as<uint2_t>(), P->h_val.as<double>(), P->hot_x.as<double>(), P->c_lrow.as<unsigned short>(),
                       P->prod.as<double>(), P->rowbits.as<unsigned long long>(), P->rowpre.as<int>(), y, alpha, beta, P->dbg, (P->dbg & 64) ? P->stamps.as<unsigned long long>() : nullptr);
    G4S_HIP_TRY(hipGetLastError());
    if (P->dbg & 64) {                                             // diagnostic: mean shader-clock ticks per section over the work items of this launch
        G4S_HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)P->n_work_items * 8);
        G4S_HIP_TRY(hipMemcpy(h.data(), P->stamps.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        double sec[5] = {0, 0, 0, 0, 0}, tot = 0;
        for (int i = 0; i < P->n_work_items; ++i) { for (int k = 0; k < 5; ++k) sec[k] += (double)(h[(size_t)i * 8 + k + 1] - h[(size_t)i * 8 + k]); tot += (double)(h[(size_t)i * 8 + 5] - h[(size_t)i * 8]); }
        static int printed = 0;
        if (printed++ < 3)
            fprintf(stderr, "g4s tb sections (mean ticks per item over %d items): zero+prologue %.0f, cold %.0f, first-tile wait %.0f, hot %.0f, flush %.0f, total %.0f\n", P->n_work_items,
                    sec[0] / P->n_work_items, sec[1] / P->n_work_items, sec[2] / P->n_work_items, sec[3] / P->n_work_items, sec[4] / P->n_work_items, tot / P->n_work_items);
    }
    return G4S_OK;
}

} // namespace g4s
