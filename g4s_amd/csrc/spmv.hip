// spmv.hip — fp64 CSR SpMV for gfx950 (B2 of include/g4s.h):  y = alpha·A·x + beta·y.
//
// The reference has no CSR mat-vec (mv/mv.c:6-27 times dense BLAS-2 on a dense copy of the graph matrix); the
// operation is defined in DESIGN.md §SpMV and restated on the CPU in oracle/g4s_oracle.c:oracle_spmv_csr.
//
// Execution plan (built once per matrix in g4s_csr_create, the role BIN plays for SpGEMM in mm/inc/BIN.h:
// classify rows by work, then give each worker an equal share):
//   * stream blocks — consecutive rows whose nonzeros fit one LDS tile (<= TILE_NNZ products, <= TILE_ROWS rows).
//     A 256-thread workgroup streams colids/values with fully coalesced loads, gathers x, writes the products to
//     LDS, then reduces each row out of LDS. With >128 rows in the block one lane sums one row left to right —
//     the oracle's order, so those rows are bit-identical to it; with fewer rows 2..64 lanes share a row and
//     finish with a wavefront shuffle reduction (segmented by row).
//   * long rows (> TILE_NNZ nonzeros; the hubs of a power-law graph) — split into LONG_CHUNK pieces, one workgroup
//     each, partial sums to a workspace; a second tiny kernel adds a row's partials in chunk order. No atomics, so
//     results are reproducible run to run.
// Stream blocks are dealt to the 8 XCDs in contiguous runs (blockIdx%8 selects the run) so that neighbouring row
// blocks, which touch neighbouring parts of x on banded/stencil matrices, share one L2.
#include "common.hpp"
#include "prims.hpp"
#include "spmv_pb.hpp"
#include "spmv_bcsr.hpp"
#include <algorithm>
#include <vector>

namespace {

constexpr int WG = 256;
constexpr int TILE_NNZ = 2048;            // fp64 products staged in LDS per workgroup (16 KiB)
#ifndef G4S_TILE_ROWS
#define G4S_TILE_ROWS 1024
#endif
constexpr int TILE_ROWS = G4S_TILE_ROWS;  // row cap per stream block (4 KiB of staged row pointers)
constexpr int LONG_CHUNK = 2048;          // nonzeros per long-row chunk (8192 until round 5: a chunk is one workgroup's serial loop of eight rounds)
constexpr int UNROLL = TILE_NNZ / WG;     // independent loads in flight per lane
constexpr int kLaneRowMax = 64;           // a longer row inside a many-row block is summed by a wavefront, not by one lane

struct LongChunk { int32_t row, k0, k1, slot; };
struct LongRow { int32_t row, slot0, nslots, pad; };

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

#ifndef G4S_STREAM_NT_Y
#define G4S_STREAM_NT_Y 0
#endif
// y stores: plain by default (a nontemporal variant measured no gain on the stencil matrices and −4 % on the cache-resident one)
__device__ __forceinline__ void store_y(double *y, int r, double s, double alpha, double beta)
{
    const double v = beta == 0.0 ? alpha * s : alpha * s + beta * y[r];
#if G4S_STREAM_NT_Y
    __builtin_nontemporal_store(v, y + r);
#else
    y[r] = v;
#endif
}

template <bool NT>
__global__ __launch_bounds__(WG) void spmv_csr_adaptive_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colids, const double *__restrict__ values,
    const double *__restrict__ x, double *__restrict__ y,
    const int4 *__restrict__ blocks, int n_stream, int stream_per_xcd,
    const LongChunk *__restrict__ chunks, int n_chunks, int chunks_pad, double *__restrict__ partials,
    double alpha, double beta)
{
    __shared__ double prod[TILE_NNZ];
    __shared__ int32_t rp[TILE_ROWS + 1];
    const int tid = threadIdx.x;

    if ((int)blockIdx.x < chunks_pad) {
        // ---- long-row chunk: strided private sums, then workgroup reduction
        if ((int)blockIdx.x >= n_chunks) return;
        const LongChunk c = chunks[blockIdx.x];
        double acc = 0.0;
        int k = c.k0 + tid;
        for (; k + 3 * WG < c.k1; k += 4 * WG) {
            const int c0 = stream_load<NT>(colids + k), c1 = stream_load<NT>(colids + k + WG);
            const int c2 = stream_load<NT>(colids + k + 2 * WG), c3 = stream_load<NT>(colids + k + 3 * WG);
            const double v0 = stream_load<NT>(values + k), v1 = stream_load<NT>(values + k + WG);
            const double v2 = stream_load<NT>(values + k + 2 * WG), v3 = stream_load<NT>(values + k + 3 * WG);
            const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
            acc += v0 * x0; acc += v1 * x1; acc += v2 * x2; acc += v3 * x3;
        }
        for (; k < c.k1; k += WG) acc += stream_load<NT>(values + k) * x[stream_load<NT>(colids + k)];
        acc = wave_sum(acc);
        if ((tid & 63) == 0) prod[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) partials[c.slot] = (prod[0] + prod[1]) + (prod[2] + prod[3]);
        return;
    }

    // ---- stream block (XCD-contiguous remap of the block index)
    const int bid = (int)blockIdx.x - chunks_pad;
    // stream_per_xcd > 0: XCD-contiguous runs (block b and b+8 share an XCD); 0: blocks in launch order
    const int lb = stream_per_xcd > 0 ? (bid % g4s::kXcds) * stream_per_xcd + bid / g4s::kXcds : bid;
    if (lb >= n_stream) return;
    const int4 blk = blocks[lb];
    const int r0 = blk.x, nrows = blk.y, k0 = blk.z, nnzb = blk.w;

    for (int r = tid; r <= nrows; r += WG) rp[r] = stream_load<NT>(rowptr + r0 + r);

    if (nnzb > 0) {
        // Branch-free: lanes past the block's last nonzero re-read it (a broadcast, never used) so that all UNROLL
        // index/value loads, then all UNROLL gathers, are in flight together instead of one wait per predicated load.
        const int last = nnzb - 1;
        int cidx[UNROLL];
        double val[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int j = min(i * WG + tid, last);
            cidx[i] = stream_load<NT>(colids + k0 + j);
            val[i] = stream_load<NT>(values + k0 + j);
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) val[i] = val[i] * x[cidx[i]];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) prod[i * WG + tid] = val[i];   // slots >= nnzb are written but never read
    }
    __syncthreads();

    if (nrows * 2 > WG || nnzb <= 16 * nrows) {
        // one lane per row, left-to-right (the oracle's summation order): blocks of many rows, and blocks of short rows however few
        // (a stencil's last block, the forced cuts of the device-side plan builder) — rows of a stencil or band are exact everywhere
        // (Round 5: a row of more than kLaneRowMax entries in such a block — a block of a power-law matrix: hundreds of short rows and one of a thousand
        // entries — is left to a whole wavefront below. As one lane's loop it was a thousand dependent LDS reads: 20 of the 24 µs of a product on a 5e5-entry
        // R-MAT, profiles/r05_small_sizes.txt. Stencil and band rows are far shorter, so their exact order stays.)
        __shared__ int heavy[TILE_NNZ / kLaneRowMax], n_heavy;
        if (tid == 0) n_heavy = 0;
        __syncthreads();
        for (int r = tid; r < nrows; r += WG) {
            const int a = rp[r] - k0, b = rp[r + 1] - k0;
            if (b - a > kLaneRowMax) { const int h = atomicAdd(&n_heavy, 1); if (h < TILE_NNZ / kLaneRowMax) heavy[h] = r; else { double s = 0.0; for (int j = a; j < b; ++j) s += prod[j]; store_y(y, r0 + r, s, alpha, beta); } continue; }
            double s = 0.0;
            for (int j = a; j < b; ++j) s += prod[j];
            store_y(y, r0 + r, s, alpha, beta);
        }
        __syncthreads();
        const int nh = min(n_heavy, TILE_NNZ / kLaneRowMax);        // (no more such rows fit a block; the guard above is for safety)
        for (int h = tid >> 6; h < nh; h += WG / 64) {              // one wavefront per heavy row: strided partial sums, then a shuffle reduction
            const int r = heavy[h], a = rp[r] - k0, b = rp[r + 1] - k0;
            double s = 0.0;
            for (int j = a + (tid & 63); j < b; j += 64) s += prod[j];
            s = wave_sum(s);
            if ((tid & 63) == 0) store_y(y, r0 + r, s, alpha, beta);
        }
    } else {
        // tpr lanes per row (power of two, <= 64, tpr·nrows <= WG), strided partials + shuffle reduction
        int tpr = 64;
        while (tpr * nrows > WG) tpr >>= 1;
        const int g = tid / tpr, sub = tid & (tpr - 1);
        double s = 0.0;
        if (g < nrows) {
            const int a = rp[g] - k0, b = rp[g + 1] - k0;
            for (int j = a + sub; j < b; j += tpr) s += prod[j];
        }
        for (int off = tpr >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (g < nrows && sub == 0) store_y(y, r0 + g, s, alpha, beta);
    }
}

__global__ void spmv_long_fixup_kernel(const LongRow *__restrict__ lrows, int n_long,
                                       const double *__restrict__ partials, double *__restrict__ y,
                                       double alpha, double beta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_long) return;
    const LongRow lr = lrows[i];
    double s = 0.0;
    for (int j = 0; j < lr.nslots; ++j) s += partials[lr.slot0 + j];
    store_y(y, lr.row, s, alpha, beta);
}

// ---- diagonal-structured matrices (stencils, banded): the index-free path.
// When every entry's offset col − row comes from a small set (≤ 32 distinct values: 5 for the 5-point, 7 for the 7-point Laplacian, 2·hb+1 for
// a band) the column indices carry no information: the values are stored by diagonal, dia[d·ld + row] = A(row, row + off[d]) (0 where the
// diagonal has no entry in that row — a boundary row), plus one 32-bit presence mask per row. One lane per row: all nd value loads and all nd
// (unit-stride, shifted) loads of x go out together, then the present products are added in ascending offset = ascending column order —
// the order of the CSR row, so the result is bit-identical to the oracle. 8·nd + 4 + 8 bytes per row instead of 12·nd + 4 + 8, and the
// x loads are coalesced instead of gathered (what north_star calls the dense-tile fallback, in the form that pays for a mat-vec: drop the
// indices — a 2-flop-per-entry product gains nothing from the matrix cores).
constexpr int kMaxDiags = 32;
struct DiaOffsets { int off[kMaxDiags]; };
// Which rows an XCD walks. Blocks of one XCD walk a contiguous eighth of the rows, so that the shifted reads of x of neighbouring row blocks go through one L2.
// A 3-D stencil has two diagonals one PLANE away (±431² for the 431³ Laplacian) and x[row + plane] is needed again one and two planes later; with contiguous
// eighths that reuse distance is three planes of x plus the streams in between, more than a 4 MiB L2, so x crosses the fabric three times (PMC: 6.72 GB fetched
// for 5.44 GB of own reads). Round 3 built a plane-sliced walk (XCD j takes the j-th eighth of EVERY plane, plane after plane: 5.65 GB fetched, L2 hit 0.36
// instead of 0.20) and a variant with staggered starting planes; on one handle in one process they ran 1 % and 5–8 % SLOWER on four boxes (DESIGN §4.1:
// profiles/r03_ab_lap7_walks.txt) — the kernel is not bound by the bytes that cross the fabric — and were removed in round 4. Likewise the switch back to one
// row per lane (two rows per lane: −4 % / −10 %, profiles/r03, tools history).

template <int ND, bool NT>
__global__ __launch_bounds__(WG) void spmv_dia_kernel(int rows, int cols, int nd, DiaOffsets offs, long long ld, const double *__restrict__ dia,
                                                       const unsigned *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y,
                                                       double alpha, double beta, int blocks_per_xcd, int row0 /* first row of this launch */)
{
    // blocks of one XCD walk a contiguous range of rows: neighbouring row blocks read neighbouring parts of x through the same L2
    const int b = (int)blockIdx.x;
    const int lb = (b % g4s::kXcds) * blocks_per_xcd + b / g4s::kXcds;
    const int row = row0 + lb * WG + (int)threadIdx.x;
    if (row >= rows) return;
    const unsigned m = mask[row];
    double v[ND], xv[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd) {
            v[d] = NT ? __builtin_nontemporal_load(dia + (long long)d * ld + row) : dia[(long long)d * ld + row];   // G4S_SPMV_NO_NT is honoured here too
            const int c = min(max(row + offs.off[d], 0), cols - 1);       // absent entries read a clamped (unused) position
            xv[d] = x[c];
        }
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd && ((m >> d) & 1u)) s += v[d] * xv[d];
    store_y(y, row, s, alpha, beta);
}

// Two consecutive rows per lane: the diagonal values, the masks and y move as 16-byte / 8-byte accesses per lane instead of 8 / 4 (the memory pipeline's
// preferred width); the arithmetic of a row is unchanged (same products, same order: bit-identical). Launched when y is 16-byte aligned and the matrix
// has at most 16 diagonals (registers); the last lane of an odd row count takes the one-row path above through `rows2`.
typedef double dia_double2 __attribute__((ext_vector_type(2)));
typedef unsigned dia_uint2 __attribute__((ext_vector_type(2)));
template <int ND, bool NT>
__global__ __launch_bounds__(WG) void spmv_dia2_kernel(int rows2 /* even part of the row count */, int cols, int nd, DiaOffsets offs, long long ld, const double *__restrict__ dia,
                                                        const unsigned *__restrict__ mask, const double *__restrict__ x, double *__restrict__ y,
                                                        double alpha, double beta, int blocks_per_xcd)
{
    const int b = (int)blockIdx.x;
    const int lb = (b % g4s::kXcds) * blocks_per_xcd + b / g4s::kXcds;
    const int row = 2 * (lb * WG + (int)threadIdx.x);
    if (row >= rows2) return;
    const dia_uint2 m = *reinterpret_cast<const dia_uint2 *>(mask + row);
    dia_double2 v[ND];
    double x0[ND], x1[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd) {
            const dia_double2 *p = reinterpret_cast<const dia_double2 *>(dia + (long long)d * ld + row);
            v[d] = NT ? __builtin_nontemporal_load(p) : *p;
            const int c = row + offs.off[d];
            x0[d] = x[min(max(c, 0), cols - 1)];
            x1[d] = x[min(max(c + 1, 0), cols - 1)];
        }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd) {
            if ((m[0] >> d) & 1u) s0 += v[d][0] * x0[d];
            if ((m[1] >> d) & 1u) s1 += v[d][1] * x1[d];
        }
    dia_double2 out;
    if (beta == 0.0) { out[0] = alpha * s0; out[1] = alpha * s1; }
    else {
        const dia_double2 old = *reinterpret_cast<const dia_double2 *>(y + row);
        out[0] = alpha * s0 + beta * old[0]; out[1] = alpha * s1 + beta * old[1];
    }
    *reinterpret_cast<dia_double2 *>(y + row) = out;
}

// One thread per row scatters the row's entries into their diagonals; fail |= 1 when an offset is not in the candidate set, a row holds
// the same column twice (a diagonal has one slot per row), or a row's columns are not ascending — the kernel adds the products in offset
// order, which is the oracle's (stored) order only for sorted rows, and "bit-identical" is what this path promises.
__global__ void dia_fill_kernel(int rows, int nd, DiaOffsets offs, long long ld, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colids,
                                const double *__restrict__ values, double *__restrict__ dia, unsigned *__restrict__ mask, int *__restrict__ fail)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    unsigned m = 0;
    int prev = -1;
    for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
        const int o = colids[k] - row;
        int lo = 0, hi = nd;                                       // offs.off is ascending
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (offs.off[mid] < o) lo = mid + 1; else hi = mid; }
        if (lo >= nd || offs.off[lo] != o || lo <= prev) { atomicOr(fail, 1); return; }   // lo <= prev: a repeated or a descending column
        prev = lo;
        m |= 1u << lo;
        dia[(long long)lo * ld + row] = values[k];
    }
    mask[row] = m;
}

// flag |= 1 if any column index is outside [0, cols): an out-of-range gather would fault the GPU.
__global__ void check_colids_kernel(const int32_t *__restrict__ colids, int64_t nnz, int32_t cols, int *flag)
{
    int bad = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = colids[k];
        bad |= (c < 0) | (c >= cols);
    }
    if (bad) atomicOr(flag, 1);
}

} // namespace

struct g4s_csr_s {
    int32_t rows = 0, cols = 0;
    int64_t nnz = 0;
    const int32_t *d_rowptr = nullptr;
    const int32_t *d_colids = nullptr;
    const double *d_values = nullptr;
    bool owns = false;
    bool use_nt = true;
    int4 *d_blocks = nullptr;
    int n_stream = 0, stream_per_xcd = 0;
    bool xcd_runs = true;
    LongChunk *d_chunks = nullptr;
    int n_chunks = 0, chunks_pad = 0;
    LongRow *d_long_rows = nullptr;
    int n_long = 0;
    double *d_partials = nullptr;
    int64_t plan_bytes = 0;
    double *d_dia = nullptr;        // diagonal-structured path: nd·ld values by diagonal
    unsigned *d_dia_mask = nullptr; // presence bits per row
    int dia_nd = 0;
    long long dia_ld = 0;
    DiaOffsets dia_offs{};
    unsigned flags = 0;             // of g4s_csr_create
    bool rowptr_checked = false;    // check_rowptr_device has passed
    bool stream_plan = false;       // the row-streaming plan exists (a large matrix that took the blocked path builds it only if it ever needs it)
    g4s::PbPlan *pb = nullptr;      // propagation-blocked path (spmv_pb.hip) for matrices without gather locality
    g4s::BcsrPlan *bcsr = nullptr;  // block-row form of an assembled FE matrix (spmv_bcsr.hip)
};

namespace {

int finish_plan(g4s_csr_s *A, size_t n_blocks, std::vector<LongChunk> &chunks, std::vector<LongRow> &lrows);

// Row classification + equal-share blocking on the host (one pass over rowptr).
int build_plan(g4s_csr_s *A, const int32_t *rowptr)
{
    const int32_t rows = A->rows;
    if (rows > 0) {
        if (rowptr[0] != 0) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr[0] != 0 (zero-based CSR expected)");
        if ((int64_t)rowptr[rows] != A->nnz) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr[rows] != nnz");
    }
    std::vector<int4> blocks;
    std::vector<LongChunk> chunks;
    std::vector<LongRow> lrows;
    blocks.reserve((size_t)(A->nnz / TILE_NNZ + rows / TILE_ROWS + 16));
    int32_t r = 0;
    while (r < rows) {
        int64_t len = (int64_t)rowptr[r + 1] - rowptr[r];
        if (len < 0) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr decreases at row %d", r);
        if (len > TILE_NNZ) {
            LongRow lr{r, (int32_t)chunks.size(), 0, 0};
            for (int64_t k = rowptr[r]; k < rowptr[r + 1]; k += LONG_CHUNK) {
                int64_t ke = k + LONG_CHUNK < rowptr[r + 1] ? k + LONG_CHUNK : rowptr[r + 1];
                chunks.push_back(LongChunk{r, (int32_t)k, (int32_t)ke, (int32_t)chunks.size()});
                lr.nslots++;
            }
            lrows.push_back(lr);
            ++r;
            continue;
        }
        const int32_t rb = r;
        int64_t nz = 0;
        while (r < rows && r - rb < TILE_ROWS) {
            len = (int64_t)rowptr[r + 1] - rowptr[r];
            if (len < 0) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr decreases at row %d", r);
            if (len > TILE_NNZ || nz + len > TILE_NNZ) break;
            nz += len;
            ++r;
        }
        blocks.push_back(make_int4(rb, r - rb, rowptr[rb], (int)nz));
    }
    if (!blocks.empty()) {
        G4S_HIP_TRY(g4s::device_malloc((void **)&A->d_blocks, sizeof(int4) * blocks.size()));
        G4S_HIP_TRY(hipMemcpy(A->d_blocks, blocks.data(), sizeof(int4) * blocks.size(), hipMemcpyHostToDevice));
    }
    return finish_plan(A, blocks.size(), chunks, lrows);
}

// The same plan built on the device, for matrices whose row pointer is already there: reading 80 M row pointers back and walking them
// on the host cost 100 ms of a 140 ms create (431³ stencil). Rows are cut into runs of kPlanRun; one thread walks each run with the
// greedy rule of build_plan (a binary search per block instead of a row-by-row loop), first counting, then — after a scan — writing. The
// forced cut at the end of a run leaves one short block per run (the kernel reduces short-row blocks with one lane per row, see above).
// Long rows are collected through a counter (few: the hubs) and laid out on the host as before.
constexpr int kPlanRun = 4096;
struct PlanLong { int32_t row, k0, k1; };
template <bool WRITE>
__global__ void plan_walk_kernel(int rows, int nruns, const int32_t *__restrict__ rowptr, const int *__restrict__ blk_off, int *__restrict__ blk_count,
                                 int4 *__restrict__ blocks, PlanLong *__restrict__ longs, int long_cap, int *__restrict__ n_long, int *__restrict__ fail)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nruns) return;
    int r = c * kPlanRun, cnt = 0;
    const int end = min(rows, r + kPlanRun);
    int4 *out = WRITE ? blocks + blk_off[c] : nullptr;
    while (r < end) {
        const int k0 = rowptr[r], len = rowptr[r + 1] - k0;
        if (len < 0) { if (!WRITE) atomicExch(fail, r + 1); return; }
        if (len > TILE_NNZ) {
            if (!WRITE) { const int slot = atomicAdd(n_long, 1); if (slot < long_cap) longs[slot] = PlanLong{r, k0, k0 + len}; }
            ++r;
            continue;
        }
        int lo = r + 1, hi = min(end, r + TILE_ROWS);               // the last row boundary e in [lo, hi] with rowptr[e] − k0 <= TILE_NNZ
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rowptr[mid] - k0 <= TILE_NNZ && rowptr[mid] >= k0) lo = mid; else hi = mid - 1;
        }
        if (WRITE) out[cnt] = make_int4(r, lo - r, k0, rowptr[lo] - k0);
        ++cnt;
        r = lo;
    }
    if (!WRITE) blk_count[c] = cnt;
}
__global__ void plan_monotone_kernel(int rows, const int32_t *__restrict__ rowptr, int *__restrict__ fail)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows && rowptr[r + 1] < rowptr[r]) atomicExch(fail, r + 1);
}

int finish_plan(g4s_csr_s *A, size_t n_blocks, std::vector<LongChunk> &chunks, std::vector<LongRow> &lrows)
{
    A->n_stream = (int)n_blocks;
    // Contiguous runs of blocks per XCD pay off while the matrix stays cache-resident across launches (+12 % on the 80 MB 5-point
    // Laplacian); on matrices far beyond the 256 MiB Infinity Cache plain launch order measured 2 % faster (banded 10M, 7-point 431³).
    A->xcd_runs = 12 * A->nnz <= (256ll << 20);
    A->stream_per_xcd = (A->n_stream + g4s::kXcds - 1) / g4s::kXcds;
    A->n_chunks = (int)chunks.size();
    A->chunks_pad = (A->n_chunks + g4s::kXcds - 1) / g4s::kXcds * g4s::kXcds;
    A->n_long = (int)lrows.size();
    if (A->n_chunks) {
        G4S_HIP_TRY(g4s::device_malloc((void **)&A->d_chunks, sizeof(LongChunk) * chunks.size()));
        G4S_HIP_TRY(hipMemcpy(A->d_chunks, chunks.data(), sizeof(LongChunk) * chunks.size(), hipMemcpyHostToDevice));
        G4S_HIP_TRY(g4s::device_malloc((void **)&A->d_long_rows, sizeof(LongRow) * lrows.size()));
        G4S_HIP_TRY(hipMemcpy(A->d_long_rows, lrows.data(), sizeof(LongRow) * lrows.size(), hipMemcpyHostToDevice));
        G4S_HIP_TRY(g4s::device_malloc((void **)&A->d_partials, sizeof(double) * chunks.size()));
    }
    A->plan_bytes = (int64_t)(sizeof(int4) * n_blocks + sizeof(LongChunk) * chunks.size() + sizeof(LongRow) * lrows.size() + sizeof(double) * chunks.size());
    return G4S_OK;
}

// The row pointers of a device-resident matrix: zero-based, ending at nnz, never decreasing. Every plan builder relies on it (binary searches over rowptr).
int check_rowptr_device(g4s_csr_s *A)
{
    if (A->rowptr_checked) return G4S_OK;
    const int32_t rows = A->rows;
    int32_t ends[2] = {0, 0};
    G4S_HIP_TRY(hipMemcpy(&ends[0], A->d_rowptr, sizeof(int32_t), hipMemcpyDeviceToHost));
    G4S_HIP_TRY(hipMemcpy(&ends[1], A->d_rowptr + rows, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (ends[0] != 0) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr[0] != 0 (zero-based CSR expected)");
    if ((int64_t)ends[1] != A->nnz) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr[rows] != nnz");
    int *d_fail = nullptr, h_fail = 0;
    if (g4s::scratch_alloc((void **)&d_fail, sizeof(int), nullptr) != G4S_OK) return G4S_ERR_NOMEM;
    hipError_t e = hipMemsetAsync(d_fail, 0, sizeof(int), nullptr);
    hipLaunchKernelGGL(plan_monotone_kernel, dim3((rows + 255) / 256), dim3(256), 0, nullptr, rows, A->d_rowptr, d_fail);
    if (e == hipSuccess) e = hipMemcpy(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost);
    g4s::scratch_free(d_fail, nullptr);
    if (e != hipSuccess) return g4s::set_error(G4S_ERR_HIP, "g4s_csr_create: rowptr check failed: %s", hipGetErrorString(e));
    if (h_fail) return g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: rowptr decreases at row %d", h_fail - 1);
    A->rowptr_checked = true;
    return G4S_OK;
}

int build_plan_device(g4s_csr_s *A)
{
    const int32_t rows = A->rows;
    G4S_TRY(check_rowptr_device(A));
    const int nruns = (rows + kPlanRun - 1) / kPlanRun;
    const int long_cap = (int)std::min<int64_t>(A->nnz / TILE_NNZ + 1, rows);
    int *d_cnt = nullptr, *d_off = nullptr, *d_scalars = nullptr;   // scalars: [0] n_long, [1] fail (row + 1)
    PlanLong *d_longs = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_cnt); (void)hipFree(d_off); (void)hipFree(d_scalars); (void)hipFree(d_longs); };
#define PLAN_TRY(expr) do { if ((expr) != hipSuccess) { const hipError_t e_ = hipGetLastError(); cleanup(); return g4s::set_error(G4S_ERR_HIP, "g4s_csr_create (device plan): %s", hipGetErrorString(e_)); } } while (0)
    PLAN_TRY(g4s::device_malloc((void **)&d_cnt, sizeof(int) * ((size_t)nruns + 1)));
    PLAN_TRY(g4s::device_malloc((void **)&d_off, sizeof(int) * ((size_t)nruns + 1)));
    PLAN_TRY(g4s::device_malloc((void **)&d_scalars, sizeof(int) * 2));
    PLAN_TRY(g4s::device_malloc((void **)&d_longs, sizeof(PlanLong) * (size_t)long_cap));
    PLAN_TRY(hipMemset(d_scalars, 0, sizeof(int) * 2));
    PLAN_TRY(hipMemset(d_cnt, 0, sizeof(int) * ((size_t)nruns + 1)));
    int h_scalars[2] = {0, 0};
    hipLaunchKernelGGL(plan_walk_kernel<false>, dim3((nruns + 63) / 64), dim3(64), 0, nullptr, rows, nruns, A->d_rowptr, (const int *)nullptr, d_cnt, (int4 *)nullptr, d_longs,
                       long_cap, d_scalars, d_scalars + 1);
    PLAN_TRY(hipGetLastError());                                   // the two plan kernels' launches were never checked
    if (g4s::prims::exclusive_scan(d_cnt, d_off, (long long)nruns + 1, nullptr) != G4S_OK) { cleanup(); return G4S_ERR_HIP; }
    int n_blocks = 0;
    PLAN_TRY(hipMemcpy(&n_blocks, d_off + nruns, sizeof(int), hipMemcpyDeviceToHost));
    PLAN_TRY(hipMemcpy(h_scalars, d_scalars, sizeof(int) * 2, hipMemcpyDeviceToHost));
    if (n_blocks) {
        PLAN_TRY(g4s::device_malloc((void **)&A->d_blocks, sizeof(int4) * (size_t)n_blocks));
        hipLaunchKernelGGL(plan_walk_kernel<true>, dim3((nruns + 63) / 64), dim3(64), 0, nullptr, rows, nruns, A->d_rowptr, d_off, d_cnt, A->d_blocks, d_longs, long_cap,
                           d_scalars, d_scalars + 1);
        PLAN_TRY(hipGetLastError());
    }
    std::vector<PlanLong> longs((size_t)std::min(h_scalars[0], long_cap));
    if (!longs.empty()) PLAN_TRY(hipMemcpy(longs.data(), d_longs, sizeof(PlanLong) * longs.size(), hipMemcpyDeviceToHost));
    PLAN_TRY(hipDeviceSynchronize());
    cleanup();
#undef PLAN_TRY
    std::sort(longs.begin(), longs.end(), [](const PlanLong &a, const PlanLong &b) { return a.row < b.row; });
    std::vector<LongChunk> chunks;
    std::vector<LongRow> lrows;
    for (const PlanLong &l : longs) {
        LongRow lr{l.row, (int32_t)chunks.size(), 0, 0};
        for (int64_t k = l.k0; k < l.k1; k += LONG_CHUNK) {
            chunks.push_back(LongChunk{l.row, (int32_t)k, (int32_t)std::min<int64_t>(k + LONG_CHUNK, l.k1), (int32_t)chunks.size()});
            lr.nslots++;
        }
        lrows.push_back(lr);
    }
    G4S_TRY(finish_plan(A, (size_t)n_blocks, chunks, lrows));
    A->stream_plan = true;
    return G4S_OK;
}

void release(g4s_csr_s *A)
{
    if (!A) return;
    if (A->owns) {
        (void)hipFree((void *)A->d_rowptr);
        (void)hipFree((void *)A->d_colids);
        (void)hipFree((void *)A->d_values);
    }
    (void)hipFree(A->d_blocks);
    (void)hipFree(A->d_chunks);
    (void)hipFree(A->d_long_rows);
    (void)hipFree(A->d_partials);
    (void)hipFree(A->d_dia);
    (void)hipFree(A->d_dia_mask);
    g4s::pb_destroy(A->pb);
    g4s::bcsr_destroy(A->bcsr);
    delete A;
}

// Try the diagonal-structured form: candidate offsets from a sample of rows (first, middle, last 2048), then one pass over the matrix that
// either fills the diagonals or reports an entry outside the candidate set. Kept when the diagonals are at least 60 % full.
int try_build_dia(g4s_csr_s *A)
{
    const int32_t rows = A->rows;
    if (rows < 1024 || A->nnz < 4096) return G4S_OK;
    std::vector<int> offs;
    const int S = 2048;
    std::vector<int32_t> cbuf;
    for (int part = 0; part < 3; ++part) {
        const int32_t ra = part == 0 ? 0 : (part == 1 ? std::max(0, rows / 2 - S / 2) : std::max(0, rows - S)), rb = std::min(rows, ra + S);
        std::vector<int32_t> rp_slice((size_t)(rb - ra) + 1);     // the sampled rows' pointers, from the device copy
        G4S_HIP_TRY(hipMemcpy(rp_slice.data(), A->d_rowptr + ra, sizeof(int32_t) * rp_slice.size(), hipMemcpyDeviceToHost));
        const int32_t *h_rowptr = rp_slice.data() - ra;
        const int64_t k0 = h_rowptr[ra], k1 = h_rowptr[rb];
        if (k1 - k0 > 64ll * S) return G4S_OK;                     // rows this long are not a stencil
        cbuf.resize((size_t)(k1 - k0));
        if (k1 > k0) G4S_HIP_TRY(hipMemcpy(cbuf.data(), A->d_colids + k0, sizeof(int32_t) * (size_t)(k1 - k0), hipMemcpyDeviceToHost));
        for (int32_t r = ra; r < rb; ++r)
            for (int64_t k = h_rowptr[r]; k < h_rowptr[r + 1]; ++k) offs.push_back(cbuf[(size_t)(k - k0)] - r);
        std::sort(offs.begin(), offs.end());
        offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
        if ((int)offs.size() > kMaxDiags) return G4S_OK;
    }
    const int nd = (int)offs.size();
    if (nd == 0 || (double)A->nnz < 0.6 * (double)nd * rows) return G4S_OK;
    DiaOffsets D{};
    for (int d = 0; d < nd; ++d) D.off[d] = offs[d];
    const long long ld = ((long long)rows + 63) / 64 * 64;
    double *dia = nullptr; unsigned *mask = nullptr; int *d_fail = nullptr, h_fail = 0;
    if (g4s::device_malloc((void **)&dia, sizeof(double) * (size_t)(ld * nd)) != hipSuccess) { (void)hipGetLastError(); return G4S_OK; }   // no room: stay on CSR
    if (g4s::device_malloc((void **)&mask, sizeof(unsigned) * (size_t)rows) != hipSuccess || g4s::device_malloc((void **)&d_fail, sizeof(int)) != hipSuccess) {
        (void)hipGetLastError(); (void)hipFree(dia); (void)hipFree(mask); return G4S_OK;
    }
    hipError_t e = hipMemset(dia, 0, sizeof(double) * (size_t)(ld * nd));
    if (e == hipSuccess) e = hipMemset(d_fail, 0, sizeof(int));
    if (e != hipSuccess) {                                          // (the three buffers used to leak on this path)
        (void)hipFree(dia); (void)hipFree(mask); (void)hipFree(d_fail);
        return g4s::set_error(G4S_ERR_HIP, "diagonal form: hipMemset failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dia_fill_kernel, dim3((rows + 255) / 256), dim3(256), 0, nullptr, rows, nd, D, ld, A->d_rowptr, A->d_colids, A->d_values, dia, mask, d_fail);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&h_fail, d_fail, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_fail);
    if (e != hipSuccess || h_fail) { (void)hipFree(dia); (void)hipFree(mask); return e == hipSuccess ? G4S_OK : g4s::set_error(G4S_ERR_HIP, "diagonal fill failed: %s", hipGetErrorString(e)); }
    A->d_dia = dia; A->d_dia_mask = mask; A->dia_nd = nd; A->dia_ld = ld; A->dia_offs = D;
    A->plan_bytes += (int64_t)(sizeof(double) * (size_t)(ld * nd) + sizeof(unsigned) * (size_t)rows);
    return G4S_OK;
}

} // namespace

G4S_API g4s_status g4s_csr_create(g4s_csr_t *out, int32_t rows, int32_t cols, int64_t nnz,
                                  const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_REQUIRE(rows >= 0 && cols >= 0 && nnz >= 0, "negative dimension");
    G4S_REQUIRE(nnz <= INT32_MAX, "nnz exceeds the int32 index type of the reference (mm/inc/define.h:14)");
    G4S_REQUIRE(rowptr, "rowptr is NULL");
    G4S_REQUIRE(nnz == 0 || (colids && values), "colids/values NULL with nnz > 0");
    int ndev = 0;
    G4S_HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return g4s::set_error(G4S_ERR_HIP, "no HIP device");

    // the create's scratch requests (scans, flags) come from a per-call arena, not from the stream-ordered pool: the pool's first use in a process creates it — 3 ms
    // in the middle of the first create (round 5); everything the plan KEEPS is allocated on its own
    struct CreateArena { CreateArena() { g4s::arena_enter(); } ~CreateArena() { g4s::arena_leave(nullptr, false); } } create_arena;
    g4s_csr_s *A = new (std::nothrow) g4s_csr_s();
    if (!A) return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    A->rows = rows; A->cols = cols; A->nnz = nnz;
    A->flags = flags;
    A->use_nt = !(flags & G4S_SPMV_NO_NT);

    std::vector<int32_t> h_rowptr_copy;
    const int32_t *h_rowptr = nullptr;
    int st = G4S_OK;
    auto fail = [&](int code) { release(A); return code; };

    // Row pointers that live on the device are planned there once the matrix is large (G4S_PLAN_HOST forces the host builder, G4S_PLAN_DEVICE
    // the device one at any size: the tests compare the two).
    const bool plan_on_device = (flags & G4S_DEVICE_POINTERS) && !getenv("G4S_PLAN_HOST") && (rows >= (1 << 18) || getenv("G4S_PLAN_DEVICE")) && rows > 0;
    if (flags & G4S_DEVICE_POINTERS) {
        A->d_rowptr = rowptr; A->d_colids = colids; A->d_values = values; A->owns = false;
        if (!plan_on_device) {
            h_rowptr_copy.resize((size_t)rows + 1);
            if (hipMemcpy(h_rowptr_copy.data(), rowptr, sizeof(int32_t) * ((size_t)rows + 1), hipMemcpyDeviceToHost) != hipSuccess)
                return fail(g4s::set_error(G4S_ERR_HIP, "g4s_csr_create: D2H copy of rowptr failed"));
            h_rowptr = h_rowptr_copy.data();
        }
    } else {
        A->owns = true;
        void *p = nullptr;
        if (g4s::device_malloc(&p, sizeof(int32_t) * ((size_t)rows + 1)) != hipSuccess) return fail(g4s::set_error(G4S_ERR_NOMEM, "hipMalloc rowptr"));
        A->d_rowptr = (const int32_t *)p;
        if (g4s::device_malloc(&p, sizeof(int32_t) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(g4s::set_error(G4S_ERR_NOMEM, "hipMalloc colids"));
        A->d_colids = (const int32_t *)p;
        if (g4s::device_malloc(&p, sizeof(double) * (size_t)(nnz ? nnz : 1)) != hipSuccess) return fail(g4s::set_error(G4S_ERR_NOMEM, "hipMalloc values"));
        A->d_values = (const double *)p;
        if (hipMemcpy((void *)A->d_rowptr, rowptr, sizeof(int32_t) * ((size_t)rows + 1), hipMemcpyHostToDevice) != hipSuccess ||
            (nnz && hipMemcpy((void *)A->d_colids, colids, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess) ||
            (nnz && hipMemcpy((void *)A->d_values, values, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess))
            return fail(g4s::set_error(G4S_ERR_HIP, "g4s_csr_create: H2D upload failed"));
        h_rowptr = rowptr;
    }

    // The row-streaming plan of a LARGE device-resident matrix waits for the path choice below (round 5): a matrix that takes the blocked path never runs it,
    // and its two plan_walk launches were 3 ms of every create on configs[1]. Only the row-pointer checks every builder relies on run here.
    st = plan_on_device ? check_rowptr_device(A) : build_plan(A, h_rowptr);
    if (st != G4S_OK) return fail(st);
    if (!plan_on_device) A->stream_plan = true;

    // Column range check on the device copy (an out-of-range gather is a GPU fault, not an error code).
    if (nnz > 0) {
        int *d_flag = nullptr, h_flag = 0;
        if (hipMalloc((void **)&d_flag, sizeof(int)) != hipSuccess) return fail(g4s::set_error(G4S_ERR_NOMEM, "hipMalloc flag"));
        (void)hipMemset(d_flag, 0, sizeof(int));
        int grid = (int)((nnz + 255) / 256 < 4096 ? (nnz + 255) / 256 : 4096);
        hipLaunchKernelGGL(check_colids_kernel, dim3(grid), dim3(256), 0, 0, A->d_colids, nnz, cols, d_flag);
        hipError_t e = hipMemcpy(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipFree(d_flag);
        if (e != hipSuccess) return fail(g4s::set_error(G4S_ERR_HIP, "g4s_csr_create: column check failed: %s", hipGetErrorString(e)));
        if (h_flag) return fail(g4s::set_error(G4S_ERR_INVALID, "g4s_csr_create: a column index is outside [0, cols)"));
    }
    // Path choice: the row-streaming kernel unless the x gathers have no locality (or the caller forces one).
    const bool want_pb = (flags & G4S_SPMV_BLOCKED) || (!(flags & G4S_SPMV_STREAM) && g4s::pb_should_use(rows, cols, nnz, A->d_colids));
    if (want_pb && nnz > 0) {
        st = g4s::pb_build(&A->pb, rows, cols, nnz, A->d_rowptr, A->d_colids, A->d_values, (flags & G4S_SPMV_UPDATABLE) != 0);
        A->plan_bytes += g4s::pb_bytes(A->pb);
        if (st != G4S_OK && (flags & G4S_SPMV_BLOCKED)) return fail(st);   // auto mode falls back to the streaming path
    }
    if (!A->pb && !A->stream_plan) {
        st = build_plan_device(A);
        if (st != G4S_OK) return fail(st);
    }
    // stencil / banded matrices: the index-free diagonal form (not when the caller forces the CSR kernels)
    if (!A->pb && !(flags & G4S_SPMV_STREAM) && nnz > 0) {
        st = try_build_dia(A);
        if (st != G4S_OK) return fail(st);
    }
    // assembled FE matrices (aligned b×b blocks, b rows with one column list): one block-column id per block instead of b² column ids
    if (!A->pb && !A->d_dia && !(flags & G4S_SPMV_STREAM) && nnz > 0) {
        st = g4s::bcsr_try_build(&A->bcsr, rows, cols, nnz, A->d_rowptr, A->d_colids, A->d_values, A->use_nt);
        if (st != G4S_OK) return fail(st);
        A->plan_bytes += g4s::bcsr_bytes(A->bcsr);
    }
    *out = A;
    return G4S_OK;
}

namespace {
// new values into the diagonals. The pattern is the plan's: row r's entries are its present diagonals in ascending offset order (dia_fill_kernel checked
// that at create), so entry j of the row belongs to the j-th set bit of its mask — no column ids, no search; one lane per row, the loads of a row's values in
// flight together, the stores unit-stride per diagonal.
template <int ND>
__global__ __launch_bounds__(256) void dia_refill_kernel(int rows, int nd, long long ld, const int32_t *__restrict__ rowptr, const unsigned *__restrict__ mask,
                                                         const double *__restrict__ values, double *__restrict__ dia)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const unsigned m = mask[row];
    const int k0 = rowptr[row], last = max(rowptr[row + 1] - 1, k0);
    double v[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) v[d] = values[min(k0 + (int)__popc(m & ((1u << d) - 1u)), last)];   // (absent diagonals read a neighbour: not stored)
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd && ((m >> d) & 1u)) dia[(long long)d * ld + row] = v[d];
}
} // namespace

// New values, same pattern (citcoms/lib/Drive_solvers.c:88,134 → construct_stiffness_B_matrix, Construct_arrays.c:740: the stiffness matrix is rebuilt
// before every Stokes solve and inside the viscosity iteration). The CSR array is replaced (owned copy: copied into; borrowed: the handle borrows the new
// array), then whatever the plan keeps of the values in another order is refreshed on `stream`: the regrouped producer stream of the blocked path (one
// gather pass through its value map), the diagonals, the block-major copy; the row-streaming kernel reads the CSR array itself.
G4S_API g4s_status g4s_csr_update_values(g4s_csr_t A, const double *values, unsigned flags, void *stream)
{
    G4S_REQUIRE(A, "NULL handle");
    if (A->nnz == 0) return G4S_OK;
    hipStream_t s = g4s::as_stream(stream);
    const bool dev = (flags & G4S_DEVICE_POINTERS) != 0;
    if (values && values != A->d_values) {
        if (A->owns) {
            G4S_HIP_TRY(hipMemcpyAsync(const_cast<double *>(A->d_values), values, sizeof(double) * (size_t)A->nnz, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        } else {
            G4S_REQUIRE(dev, "the handle borrows device arrays: new values must be a device array too (it is borrowed from here on)");
            A->d_values = values;
        }
    }
    if (A->pb) {
        if (g4s::pb_has_value_map(A->pb)) return g4s::pb_update_values(A->pb, A->d_values, s);
        // created without G4S_SPMV_UPDATABLE: the regrouping is done again (the cost of a create, on the NULL stream like a create)
        G4S_HIP_TRY(hipStreamSynchronize(s));
        A->plan_bytes -= g4s::pb_bytes(A->pb);
        g4s::pb_destroy(A->pb);
        A->pb = nullptr;
        const int st = g4s::pb_build(&A->pb, A->rows, A->cols, A->nnz, A->d_rowptr, A->d_colids, A->d_values, false);
        if (st != G4S_OK) {                                         // no blocked plan any more: the handle must still multiply — the row-streaming plan, built now if it never was
            if (!A->stream_plan) G4S_TRY(build_plan_device(A));
            return st;
        }
        A->plan_bytes += g4s::pb_bytes(A->pb);
        return G4S_OK;
    }
    if (A->d_dia) {
        const dim3 grid((A->rows + 255) / 256), block(256);
        if (A->dia_nd <= 8) hipLaunchKernelGGL(dia_refill_kernel<8>, grid, block, 0, s, A->rows, A->dia_nd, A->dia_ld, A->d_rowptr, A->d_dia_mask, A->d_values, A->d_dia);
        else if (A->dia_nd <= 16) hipLaunchKernelGGL(dia_refill_kernel<16>, grid, block, 0, s, A->rows, A->dia_nd, A->dia_ld, A->d_rowptr, A->d_dia_mask, A->d_values, A->d_dia);
        else hipLaunchKernelGGL(dia_refill_kernel<32>, grid, block, 0, s, A->rows, A->dia_nd, A->dia_ld, A->d_rowptr, A->d_dia_mask, A->d_values, A->d_dia);
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }
    if (A->bcsr) return g4s::bcsr_update_values(A->bcsr, A->d_colids, A->d_values, s);
    return G4S_OK;
}

G4S_API g4s_status g4s_csr_destroy(g4s_csr_t A)
{
    release(A);
    return G4S_OK;
}

G4S_API g4s_status g4s_csr_get_info(g4s_csr_t A, g4s_csr_info *info)
{
    G4S_REQUIRE(A && info, "NULL argument");
    info->rows = A->rows; info->cols = A->cols; info->nnz = A->nnz;
    info->stream_blocks = A->n_stream; info->long_rows = A->n_long; info->long_chunks = A->n_chunks;
    info->tile_nnz = TILE_NNZ; info->tile_rows = TILE_ROWS; info->long_chunk_nnz = LONG_CHUNK;
    info->algorithmic_bytes = 12 * A->nnz + 4 * ((int64_t)A->rows + 1) + 8 * (int64_t)A->rows + 8 * (int64_t)A->cols;
    info->plan_bytes = A->plan_bytes;
    info->spmv_path = A->pb ? 1 : (A->d_dia ? 3 : (A->bcsr ? 4 : 0));
    return G4S_OK;
}

G4S_API g4s_status g4s_csr_device_arrays(g4s_csr_t A, const int32_t **rowptr, const int32_t **colids, const double **values)
{
    G4S_REQUIRE(A, "NULL handle");
    if (rowptr) *rowptr = A->d_rowptr;
    if (colids) *colids = A->d_colids;
    if (values) *values = A->d_values;
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv(g4s_csr_t A, const double *x_dev, double *y_dev, double alpha, double beta, void *stream)
{
    G4S_REQUIRE(A, "NULL handle");
    if (A->rows == 0) return G4S_OK;
    G4S_REQUIRE(y_dev, "y is NULL");
    G4S_REQUIRE(x_dev || A->nnz == 0, "x is NULL");
    G4S_REQUIRE((const void *)x_dev != (const void *)y_dev, "x and y must not alias");
    hipStream_t s = g4s::as_stream(stream);
    if (A->pb) return g4s::pb_spmv(A->pb, x_dev, y_dev, alpha, beta, s);
    if (A->bcsr) return g4s::bcsr_spmv(A->bcsr, x_dev, y_dev, alpha, beta, s);
    if (A->d_dia) {
        // two rows per lane where it applies (≤ 16 diagonals, y 16-byte aligned): the even part of the rows; an odd last row by the one-row kernel
        const bool two = A->dia_nd <= 16 && (reinterpret_cast<uintptr_t>(y_dev) & 15u) == 0 && A->rows >= 2;
        const int rows2 = two ? (A->rows & ~1) : 0;
        if (rows2) {
            const int nblocks = (rows2 / 2 + WG - 1) / WG, per_xcd = (nblocks + g4s::kXcds - 1) / g4s::kXcds;
            const dim3 grid((unsigned)(per_xcd * g4s::kXcds)), block(WG);
#define G4S_DIA2_LAUNCH(ND)                                                                                                                                           \
    do {                                                                                                                                                          \
        if (A->use_nt) hipLaunchKernelGGL((spmv_dia2_kernel<ND, true>), grid, block, 0, s, rows2, A->cols, A->dia_nd, A->dia_offs, A->dia_ld, A->d_dia, A->d_dia_mask, x_dev, y_dev, alpha, beta, per_xcd);  \
        else hipLaunchKernelGGL((spmv_dia2_kernel<ND, false>), grid, block, 0, s, rows2, A->cols, A->dia_nd, A->dia_offs, A->dia_ld, A->d_dia, A->d_dia_mask, x_dev, y_dev, alpha, beta, per_xcd);           \
    } while (0)
            if (A->dia_nd <= 8) G4S_DIA2_LAUNCH(8);
            else G4S_DIA2_LAUNCH(16);
#undef G4S_DIA2_LAUNCH
        }
        const int tail0 = rows2;                                   // rows [tail0, rows) by the one-row kernel: all of them, or the odd last one
        if (tail0 < A->rows) {
            const int n_tail = A->rows - tail0;
            const int nblocks = (n_tail + WG - 1) / WG, per_xcd = (nblocks + g4s::kXcds - 1) / g4s::kXcds;
            const dim3 grid((unsigned)(per_xcd * g4s::kXcds)), block(WG);
#define G4S_DIA_LAUNCH(ND)                                                                                                                                            \
    do {                                                                                                                                                          \
        if (A->use_nt) hipLaunchKernelGGL((spmv_dia_kernel<ND, true>), grid, block, 0, s, A->rows, A->cols, A->dia_nd, A->dia_offs, A->dia_ld, A->d_dia, A->d_dia_mask, x_dev, y_dev, alpha, beta, per_xcd, tail0);  \
        else hipLaunchKernelGGL((spmv_dia_kernel<ND, false>), grid, block, 0, s, A->rows, A->cols, A->dia_nd, A->dia_offs, A->dia_ld, A->d_dia, A->d_dia_mask, x_dev, y_dev, alpha, beta, per_xcd, tail0);           \
    } while (0)
            if (A->dia_nd <= 8) G4S_DIA_LAUNCH(8);
            else if (A->dia_nd <= 16) G4S_DIA_LAUNCH(16);
            else G4S_DIA_LAUNCH(32);
#undef G4S_DIA_LAUNCH
        }
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }
    const int grid = A->chunks_pad + A->stream_per_xcd * g4s::kXcds;
    if (grid > 0) {
        if (A->use_nt)
            hipLaunchKernelGGL(spmv_csr_adaptive_kernel<true>, dim3(grid), dim3(WG), 0, s, A->d_rowptr, A->d_colids, A->d_values,
                               x_dev, y_dev, A->d_blocks, A->n_stream, A->xcd_runs ? A->stream_per_xcd : 0, A->d_chunks, A->n_chunks,
                               A->chunks_pad, A->d_partials, alpha, beta);
        else
            hipLaunchKernelGGL(spmv_csr_adaptive_kernel<false>, dim3(grid), dim3(WG), 0, s, A->d_rowptr, A->d_colids, A->d_values,
                               x_dev, y_dev, A->d_blocks, A->n_stream, A->xcd_runs ? A->stream_per_xcd : 0, A->d_chunks, A->n_chunks,
                               A->chunks_pad, A->d_partials, alpha, beta);
        G4S_HIP_TRY(hipGetLastError());
    }
    if (A->n_long > 0) {
        hipLaunchKernelGGL(spmv_long_fixup_kernel, dim3((A->n_long + 255) / 256), dim3(256), 0, s, A->d_long_rows, A->n_long,
                           A->d_partials, y_dev, alpha, beta);
        G4S_HIP_TRY(hipGetLastError());
    }
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_csr_i32_f64(int32_t rows, int32_t cols, const int32_t *rowptr, const int32_t *colids,
                                        const double *values, const double *x, double *y,
                                        double alpha, double beta, unsigned flags)
{
    G4S_REQUIRE(rows >= 0 && cols >= 0, "negative dimension");
    G4S_REQUIRE(rowptr, "rowptr is NULL");
    if (rows == 0) return G4S_OK;
    G4S_REQUIRE(y, "y is NULL");
    const bool dev = (flags & G4S_DEVICE_POINTERS) != 0;
    int32_t nnz32 = 0;
    if (dev) G4S_HIP_TRY(hipMemcpy(&nnz32, rowptr + rows, sizeof(int32_t), hipMemcpyDeviceToHost));
    else nnz32 = rowptr[rows];
    G4S_REQUIRE(nnz32 >= 0, "rowptr[rows] is negative");
    g4s_csr_t A = nullptr;
    // one call, one product: the blocked path's regrouping (tens of ms) cannot pay off — stay on the streaming path unless asked
    if (!(flags & G4S_SPMV_BLOCKED)) flags |= G4S_SPMV_STREAM;
    G4S_TRY(g4s_csr_create(&A, rows, cols, nnz32, rowptr, colids, values, flags));
    int st = G4S_OK;
    if (dev) {
        st = g4s_spmv(A, x, y, alpha, beta, nullptr);
        if (st == G4S_OK && hipStreamSynchronize(nullptr) != hipSuccess) st = g4s::set_error(G4S_ERR_HIP, "synchronize failed");
    } else {
        double *dx = nullptr, *dy = nullptr;
        if (g4s::device_malloc((void **)&dx, sizeof(double) * (size_t)(cols ? cols : 1)) != hipSuccess ||
            g4s::device_malloc((void **)&dy, sizeof(double) * (size_t)rows) != hipSuccess) {
            st = g4s::set_error(G4S_ERR_NOMEM, "hipMalloc of x/y failed");
        } else if ((cols && hipMemcpy(dx, x, sizeof(double) * (size_t)cols, hipMemcpyHostToDevice) != hipSuccess) ||
                   (beta != 0.0 && hipMemcpy(dy, y, sizeof(double) * (size_t)rows, hipMemcpyHostToDevice) != hipSuccess)) {
            st = g4s::set_error(G4S_ERR_HIP, "H2D copy of x/y failed");
        } else {
            st = g4s_spmv(A, dx, dy, alpha, beta, nullptr);
            if (st == G4S_OK && hipMemcpy(y, dy, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost) != hipSuccess)
                st = g4s::set_error(G4S_ERR_HIP, "D2H copy of y failed");
        }
        (void)hipFree(dx);
        (void)hipFree(dy);
    }
    g4s_csr_destroy(A);
    return st;
}
