// spgemm.hip — fp64 CSR SpGEMM C = A·B for gfx950 (B1 of include/g4s.h).
//
// Follows the reference's two-phase row-wise Gustavson algorithm with a per-row open-addressing hash accumulator
// (HashSpGEMM, mm/inc/hash_mult.h:1028-1057): symbolic counts the distinct columns of every output row and an exclusive
// scan gives crpt (hash_symbolic, :496-508, kernel :65-109); numeric re-inserts with multiply/add, then compacts and sorts
// each row by column (hash_numeric, :559-608; sort_and_store_table2mat, :526-553). Hash = (key·107) & (size−1), linear
// probing, empty = −1 (hash_mult.h:89,100; define.h:12). Rows are classified by the same flop upper bound the reference
// bins by (BIN::set_intprod_num / set_bin_id, BIN.h:78-95,158-177: min(Σ_j nnz(B(acol_j,:)), cols)).
//
// MI355X mapping (one accumulator per ROW in LDS instead of one table per THREAD in cache); DESIGN.md §4.2 has the class table:
//   * tiny rows  (bound ≤ 32)     one wavefront per row, 64-slot table, 8 lanes per A-entry
//   * small rows (bound ≤ 512)    one 256-thread workgroup per row, 1024-slot table (numeric: + in-LDS bitonic sort)
//   * larger rows, B ≤ 4 M       LDS bitmap windows instead of a table, one row per workgroup of a persistent grid, three shapes
//     non-empty columns           (1024 / 512 / 256 threads: windows of 2^20 / 2^19 / 2^18 columns, 1 / 2 / 4 workgroups per CU) on B's
//                                 non-empty columns renumbered (build_column_map): symbolic marks, popcounts and emits the sorted
//                                 columns (spgemm_symbolic_window_kernel); numeric adds the values per chunk of sorted columns through
//                                 a bucketed slot index (spgemm_numeric_big_kernel, rows up to 1 M outputs). A row's products are
//                                 dealt to the waves flat, in 64-entry units of the B rows (flat_products); the long classes take
//                                 their rows longest first through a counter.
//   * mid-size rows, wider B      key tables up to 32 K slots (the largest one optimistic: it hands the row to the window kernel
//                                 when a probe sequence gets long)
//   * hub rows (flop > 2 M in     bitmap-rank path in HBM: mark columns in a per-row bitmap, popcount-prefix it; a column's
//     symbolic, > 1 M outputs     rank is its position in the sorted output row, so products are atomically added straight
//     in numeric)                 into C (no table, no sort).
// Numeric classes use the EXACT row sizes known from symbolic, so the tables are at most half full.
// Integer results (crpt, ccol) are exact; fp64 sums are accumulated with LDS/HBM atomics, i.e. in a different order than
// the reference's (j outer, k inner): equal within the 1e-10 relative tolerance of the north star, not bit for bit.
#include "common.hpp"
#ifndef G4S_SPGEMM_UPR
#define G4S_SPGEMM_UPR 6   /* 4 until the end of round 4; with a round's descriptors requested together 5 / 6 / 7 / 8 units ran 31.2 / 31.0 / 31.25 / 31.75 ms against 31.6 (2: 34.1) */
#endif
#ifndef G4S_KO
#define G4S_KO 0   // timing-only knock-outs of the big-row numeric kernel (wrong results; tools/ab_variants.sh): 1 no halvings, 2 no LDS atomics, 4 no bucket index, 8 no stores, 16 no accumulate step
#endif
#include "prims.hpp"
#include "readback.hpp"
#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr int kHashScal = 107; // HASH_SCAL, mm/inc/define.h:12
constexpr int kEmpty = -1;

// ------------------------------------------------------------------------------------------------ small utilities
// Device memory of a product: small transients from the library's never-trimming pool, large blocks (>= 64 MiB: outputs, column
// scratch, hub bitmaps) from its caching allocator (runtime.cpp) — plain hipMalloc of gigabytes took anything between 1 ms and 4 s
// from one call to the next (measured on the one-call form: 95 ms to 2300 ms for the same product).
thread_local hipStream_t t_stream = nullptr;   // the stream of the API call in progress on this thread: pool allocations and frees are ordered on it
thread_local bool t_idle = false;              // the call's stream has been synchronised and nothing was launched since: its big blocks go back to the cache without a device-wide wait
struct DevBuf {
    void *p = nullptr;
    hipStream_t s = nullptr;
    bool big = false;
    ~DevBuf() { release(); }
    int alloc(size_t bytes, bool for_caller = false)              // for_caller: the pointer is handed out and comes back through g4s_dev_free
    {
        release();
        big = for_caller || bytes >= ((size_t)64 << 20);
        if (big) return g4s::big_alloc(&p, bytes);
        s = t_stream;
        return g4s::scratch_alloc(&p, bytes, s);
    }
    void release()
    {
        if (!p) return;
        if (big) (void)g4s::big_free(p, t_idle); else g4s::scratch_free(p, s);
        p = nullptr;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Transients of one API call come from the per-call arena (runtime.cpp). Declared before every DevBuf of the call, so it ends after them.
struct ArenaScope {
    ArenaScope() { g4s::arena_enter(); }
    ~ArenaScope() { g4s::arena_leave(t_stream, t_idle); }
};

// (Round 4 tried a second stream for the short-row kernels — one wavefront per row, bound by their chain of dependent loads — beside the pre-passes of the
// window classes, which are latency-bound too: 33.5 ms against 32.6 for the call, three alternating runs each. The persistent window kernels behind the pre-passes
// start on CUs the short rows still hold, and their longest-first tickets then run late on those. One stream.)

// Host-side phase times of one call under G4S_DEBUG (declared first in a function: its destructor runs after every buffer of the call has been released).
struct DbgPhases {
    using clk = std::chrono::steady_clock;
    const char *what;
    bool on;
    clk::time_point t0, last;
    std::string line;
    explicit DbgPhases(const char *w) : what(w), on(getenv("G4S_DEBUG") != nullptr), t0(clk::now()), last(t0) {}
    void mark(const char *name)
    {
        if (!on) return;
        const auto now = clk::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.2f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~DbgPhases()
    {
        if (!on) return;
        mark("release");
        fprintf(stderr, "g4s %s host phases (ms):%s | total %.2f\n", what, line.c_str(), std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    }
};

__device__ __forceinline__ int lds_peek(const int *p) { return *reinterpret_cast<const volatile int *>(p); }

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------ flop per row (a4)
// Eight lanes per row (most rows of a power-law graph hold a handful of entries); a row past 64 entries is left to the whole
// workgroup afterwards — eight lanes looping over a hub's 16 K entries were a 2 ms tail on their own — and the grand total takes
// one atomic per workgroup, not one per row. BIN::set_intprod_num (BIN.h:78-95) / compute_flop (mkl_mult.h:8-38).
__global__ __launch_bounds__(256) void row_flop_kernel(int M, const int *__restrict__ arpt, const int *__restrict__ acol,
                                                        const int *__restrict__ brpt, long long *__restrict__ row_flop,
                                                        unsigned long long *__restrict__ total)
{
    __shared__ long long s_f[32];
    __shared__ long long s_part[4];
    const int t = threadIdx.x, sub = t & 7, r = t >> 3;
    const int row = blockIdx.x * 32 + r;
    const int a0 = row < M ? arpt[row] : 0, a1 = row < M ? arpt[row + 1] : 0;
    const bool is_long = a1 - a0 > 64;
    long long f = 0;
    if (!is_long)
        for (int j = a0 + sub; j < a1; j += 8) {
            const int c = acol[j];
            f += brpt[c + 1] - brpt[c];
        }
    f += __shfl_down(f, 4, 8);
    f += __shfl_down(f, 2, 8);
    f += __shfl_down(f, 1, 8);
    if (sub == 0) s_f[r] = is_long ? -1 : f;
    __syncthreads();
    for (int q = 0; q < 32; ++q) {
        if (s_f[q] >= 0) continue;                                 // uniform
        const int qrow = blockIdx.x * 32 + q;
        long long g = 0;
        for (int j = arpt[qrow] + t; j < arpt[qrow + 1]; j += 256) {
            const int c = acol[j];
            g += brpt[c + 1] - brpt[c];
        }
        g = wave_sum_ll(g);
        if ((t & 63) == 0) s_part[t >> 6] = g;
        __syncthreads();
        if (t == 0) s_f[q] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
    if (t < 32) {
        const long long v = s_f[t];
        if (blockIdx.x * 32 + t < M) row_flop[blockIdx.x * 32 + t] = v;
        long long sum = blockIdx.x * 32 + t < M ? v : 0;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_down(sum, off, 32);
        if (t == 0 && sum) atomicAdd(total, (unsigned long long)sum);
    }
}

// Round 4: the same numbers, entry-parallel — the kernel above spends 0.7 ms of a 34 ms product waiting (8 lanes per row, most rows hold three entries, 59 % none:
// PMC waiting 0.90, issuing 0.07): every A-entry writes the length of its B row, an exclusive scan runs over the entries, and a row's flop is the difference of
// the scan at its two ends. Balanced whatever the row lengths are.
__global__ void entry_flop_kernel(long long annz, const int *__restrict__ acol, const int *__restrict__ brpt, long long *__restrict__ f)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > annz) return;
    if (k == annz) { f[k] = 0; return; }                           // (the scan runs over annz + 1 entries: its last output is the total)
    const int c = acol[k];
    f[k] = brpt[c + 1] - brpt[c];
}
// the same with the range check of A's column ids fused in (an id outside [0, K) counts no products and raises the flag: the caller reads flag and total together)
__global__ void entry_flop_checked_kernel(long long annz, const int *__restrict__ acol, int K, const int *__restrict__ brpt, long long *__restrict__ f, unsigned long long *flag)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > annz) return;
    if (k == annz) { f[k] = 0; return; }
    const int c = acol[k];
    const bool ok = c >= 0 && c < K;
    f[k] = ok ? brpt[c + 1] - brpt[c] : 0;
    if (!ok) atomicOr(flag, 1ull);
}
__global__ void row_flop_from_scan_kernel(int M, const int *__restrict__ arpt, const long long *__restrict__ P, long long *__restrict__ row_flop)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) row_flop[i] = P[arpt[i + 1]] - P[arpt[i]];
}

// Range check of A's column ids against B's row count (an out-of-range id would fault in brpt[c]).
__global__ void check_range_kernel(const int *__restrict__ ids, long long n, int bound, int *flag)
{
    int bad = 0;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const int c = ids[k];
        bad |= (c < 0) | (c >= bound);
    }
    if (bad) atomicOr(flag, 1);
}

// B's rows must be sorted by column (the merge and window kernels cut them at column boundaries; repeated columns are fine). Counted in two passes that
// cost nothing next to the product: every position whose column is smaller than its predecessor's (check_descents_kernel, fused with the range check), and
// how many of those are the first entry of a row that follows a non-empty row (row_start_descents_kernel) — the only place a descent may be. Sorted iff equal.
__global__ void check_descents_kernel(const int *__restrict__ ids, long long n, int bound, int *__restrict__ flag, unsigned long long *__restrict__ descents)
{
    int bad = 0;
    unsigned long long d = 0;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const int c = ids[k];
        bad |= (c < 0) | (c >= bound);
        if (k > 0 && c < ids[k - 1]) ++d;
    }
    if (bad) atomicOr(flag, 1);
    __shared__ unsigned long long s_d[4];                          // one atomic per WORKGROUP: nearly every wave of a sparse matrix sees a row boundary, and
    d = (unsigned long long)wave_sum_ll((long long)d);             // 10^5 atomics on one counter cost more than the pass itself (0.2 ms)
    if ((threadIdx.x & 63) == 0) s_d[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0 && s_d[0] + s_d[1] + s_d[2] + s_d[3]) atomicAdd(descents, s_d[0] + s_d[1] + s_d[2] + s_d[3]);
}
__global__ __launch_bounds__(256) void row_start_descents_kernel(int K, const int *__restrict__ rpt, const int *__restrict__ ids, unsigned long long *__restrict__ legal)
{
    unsigned long long d = 0;
    const int nnz = rpt[K];
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < K; r += gridDim.x * blockDim.x) {
        if (r == 0) continue;
        const int k = rpt[r];
        if (k > rpt[r - 1] && k < nnz && ids[k] < ids[k - 1]) ++d;  // the first row that starts at k behind a non-empty row
    }
    __shared__ unsigned long long s_d[4];
    d = (unsigned long long)wave_sum_ll((long long)d);
    if ((threadIdx.x & 63) == 0) s_d[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0 && s_d[0] + s_d[1] + s_d[2] + s_d[3]) atomicAdd(legal, s_d[0] + s_d[1] + s_d[2] + s_d[3]);
}

// ------------------------------------------------------------------------------------------------ row classes
enum { CLS_EMPTY = 0, CLS_TINY, CLS_SMALL, CLS_MEDIUM, CLS_LARGE, CLS_M2, CLS_M3, CLS_HUB, CLS_RANK /* numeric only: the rows whose symbolic pass wrote cuts (spgemm_rank.hpp); size < 0 */, CLS_COUNT };

// Class i (1..6) takes a row whose size is <= lim[i-1]; sizes are the flop bound clipped at cols (BIN.h:164), except where
// raw[i-1] is set (the unclipped bound decides). Beyond the last limit: hub.
struct ClassLimits { long long lim[6]; int raw[6]; };
// symbolic: tables are keys only. TINY 64 slots, SMALL 1 K, MEDIUM 16 K, LARGE = optimistic 32 K-slot table — a failed attempt costs
// more than the bitmap path, so only rows whose raw bound is within 4/3 of the table's 24 K-entry limit try it. M2 (raw flop <= 2 M):
// LDS bitmap windows (spgemm_symbolic_window_kernel), which also takes the rows whose optimistic table filled up. Beyond: hub.
#ifndef G4S_SPGEMM_SYM_MEDIUM
#define G4S_SPGEMM_SYM_MEDIUM 4096                            /* rows of more products (clipped bound) write cuts and take the rank kernel: 8192 → 4096 −0.7 ms, 2048 +0.1, 1024 +1.8 on configs[2] (profiles/r05_spgemm_ab.txt) */
#endif
constexpr ClassLimits kSymLimits{{32, 512, G4S_SPGEMM_SYM_MEDIUM, 32768, 2097152, -1}, {0, 0, 0, 1, 1, 0}};
// numeric: by the exact nz of the output row; tables hold keys + fp64 at <= 50 % fill: TINY 64, SMALL 1 K, MEDIUM 2 K, LARGE 4 K, M2 8 K slots.
#ifndef G4S_SPGEMM_BIG_LIMIT
#define G4S_SPGEMM_BIG_LIMIT 1048576
#endif
constexpr ClassLimits kNumLimits{{32, 512, 1024, 2048, 4096, G4S_SPGEMM_BIG_LIMIT}, {0, 0, 0, 0, 0, 0}};   // M3: the all-LDS big-row kernel

// (Round 4: few workgroups, each on a contiguous range of rows, global atomics once per workgroup and class. The first form — one workgroup per 256 rows, its
// class counts added to eight global counters — put 65 K atomics on eight addresses of one L2 channel: 97 µs per launch for a 24 MB pass, four launches per product.)
constexpr int kClassBlocks = 512;
__device__ __forceinline__ int class_of(long long u, const ClassLimits &lim, int cols_clip)
{
    if (u < 0) return CLS_RANK;
    if (u == 0) return CLS_EMPTY;
    const long long uc = (cols_clip > 0 && u > cols_clip) ? cols_clip : u;   // BIN.h:164 clips the bound at cols
    int c = CLS_HUB;
#pragma unroll
    for (int i = 5; i >= 0; --i)
        if ((lim.raw[i] ? u : uc) <= lim.lim[i]) c = CLS_TINY + i;
    return c;
}
__global__ __launch_bounds__(256) void classify_kernel(int M, const long long *__restrict__ size, ClassLimits lim, int cols_clip,
                                                       int *__restrict__ cls, int *__restrict__ hist)
{
    __shared__ int s_hist[CLS_COUNT];
    if (threadIdx.x < CLS_COUNT) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int per = (M + (int)gridDim.x - 1) / (int)gridDim.x, r0 = blockIdx.x * per, r1 = min(M, r0 + per);
    int cnt[CLS_COUNT] = {0};
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const int c = class_of(size[i], lim, cols_clip);
        cls[i] = c;
#pragma unroll
        for (int k = 0; k < CLS_COUNT; ++k) cnt[k] += c == k;
    }
#pragma unroll
    for (int k = 0; k < CLS_COUNT; ++k) {
        int v = cnt[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_hist[k], v);
    }
    __syncthreads();
    if (threadIdx.x < CLS_COUNT && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

// the same ranges: count the range's rows per class, reserve the places with one atomic per class, then place the rows (wave ballots rank them)
__global__ __launch_bounds__(256) void scatter_rows_kernel(int M, const int *__restrict__ cls, const int *__restrict__ hist /* the class counts (classify_kernel) */,
                                                           int *__restrict__ cursor /* zeroed */, int *__restrict__ lists)
{
    __shared__ int s_cnt[CLS_COUNT], s_base[CLS_COUNT], s_off[CLS_COUNT];
    if (threadIdx.x < CLS_COUNT) {                                 // a class's list starts behind the classes in front of it: nine counts, summed here (they came from the
        s_cnt[threadIdx.x] = 0;                                    // host by a copy of their prefix sums until round 5: a host→device copy per classification)
        int o = 0;
        for (int k = 0; k < (int)threadIdx.x; ++k) o += hist[k];
        s_off[threadIdx.x] = o;
    }
    __syncthreads();
    const int per = (M + (int)gridDim.x - 1) / (int)gridDim.x, r0 = blockIdx.x * per, r1 = min(M, r0 + per), lane = threadIdx.x & 63;
    int cnt[CLS_COUNT] = {0};
    for (int i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
        const int c = cls[i];
#pragma unroll
        for (int k = 0; k < CLS_COUNT; ++k) cnt[k] += c == k;
    }
#pragma unroll
    for (int k = 0; k < CLS_COUNT; ++k) {
        int v = cnt[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0 && v) atomicAdd(&s_cnt[k], v);
    }
    __syncthreads();
    if (threadIdx.x < CLS_COUNT) { s_base[threadIdx.x] = s_off[threadIdx.x] + (s_cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]) : 0); s_cnt[threadIdx.x] = 0; }   // cursor[c] starts at the class offset
    __syncthreads();
    for (int i0 = r0; i0 < r1; i0 += blockDim.x) {                 // uniform trip count: the ballots below need every lane
        const int i = i0 + threadIdx.x;
        const int c = i < r1 ? cls[i] : -1;
        int place = 0;
#pragma unroll
        for (int k = 0; k < CLS_COUNT; ++k) {
            const unsigned long long m = __ballot(c == k);
            int wbase = 0;
            if (lane == 0 && m) wbase = atomicAdd(&s_cnt[k], __popcll(m));
            wbase = __shfl(wbase, 0, 64);
            if (c == k) place = s_base[k] + wbase + __popcll(m & ((1ull << lane) - 1ull));
        }
        if (i < r1) lists[place] = i;
    }
}

__global__ void gather_ranges_kernel(const int *__restrict__ rows, int n, const int *__restrict__ arpt, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = arpt[rows[i]]; out[2 * i + 1] = arpt[rows[i] + 1]; }
}

__global__ void nz_to_ll_kernel(int M, const int *__restrict__ crpt, const long long *__restrict__ cut_off /* not NULL: rows with cuts get a negative size (CLS_RANK) */, long long *__restrict__ nz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const long long v = (long long)crpt[i + 1] - crpt[i];
    nz[i] = (cut_off && cut_off[i] >= 0) ? -1 - v : v;
}

// ------------------------------------------------------------------------------------------------ LDS hash kernels
// THREADS threads cooperate on one row; WG = 256 (or 1024) threads host WG/THREADS rows. GROUP lanes share one A-entry and
// stride the matching B row (coalesced bcol/bval reads).
template <int TABLE>
__device__ __forceinline__ int hash_of(int key) { return (key * kHashScal) & (TABLE - 1); }

// Walk the A-entries first, first+stride, … of one row with a two-stage software pipeline: while the B row of entry j is
// processed, the row-pointer pair of entry j+stride and the column id of entry j+2·stride are already in flight. Without it
// every entry pays three dependent memory latencies (acol → brpt → bcol) back to back, and the one-workgroup-per-CU classes
// (128 KiB tables) are bound by exactly that chain. body(b0, b1, av) returns false to stop early.
template <bool WITH_VAL, typename Body>
__device__ __forceinline__ void walk_a_entries(int a0, int a1, int first, int stride, const int *__restrict__ acol,
                                               const double *__restrict__ aval, const int *__restrict__ blo, const int *__restrict__ bhi, Body body)
{
    int j = a0 + first;
    int c1 = 0, b0 = 0, b1 = 0;
    double av0 = 0.0, av1 = 0.0;
    if (j < a1) {
        const int c0 = acol[j];
        if (WITH_VAL) av0 = aval[j];
        b0 = blo[c0];
        b1 = bhi[c0];
    }
    if (j + stride < a1) {
        c1 = acol[j + stride];
        if (WITH_VAL) av1 = aval[j + stride];
    }
    while (j < a1) {
        const int j1 = j + stride, j2 = j1 + stride;
        int c2 = 0, nb0 = 0, nb1 = 0;
        double av2 = 0.0;
        if (j2 < a1) {
            c2 = acol[j2];
            if (WITH_VAL) av2 = aval[j2];
        }
        if (j1 < a1) {
            nb0 = blo[c1];
            nb1 = bhi[c1];
        }
        if (!body(b0, b1, av0)) return;
        b0 = nb0; b1 = nb1; av0 = av1; c1 = c2; av1 = av2; j = j1;
    }
}

// Long B rows. A lane group of 2^gs lanes walking a B row of thousands of entries serialises thousands of load → insert rounds
// while the rest of the workgroup idles (B-row lengths in a power-law graph are as skewed as A's). In the one-workgroup-per-row
// kernels a B row longer than kLongB is therefore only RECORDED (b0, b1, a's value) in a small LDS list during the walk and is
// processed afterwards by the whole workgroup, lanes striding it coalesced. A full list falls back to in-group processing.
constexpr int kLongCap = 255;                        // entries; slot 0 of the int4 array holds the counter
constexpr size_t kLongListBytes = sizeof(int4) * (kLongCap + 1);
// numeric tables of the one-row-per-workgroup kernels: keys + fp64 (12 B per slot), the long-B list, the dense copy for the sort
constexpr size_t num_lds_bytes(int table) { return (size_t)table * 12 + kLongListBytes + (size_t)table / 2 * 12 + 16; }
constexpr size_t sym_lds_bytes(int rpb, int table) { return sizeof(int) * ((size_t)rpb * table + 2 * rpb) + (rpb == 1 ? kLongListBytes : 0); }
__device__ __forceinline__ int long_b_threshold(int threads) { return threads >= 256 ? threads / 4 : 1 << 30; }

// Called by every lane of a group with the same (b0, b1). Returns true if the B row was deferred to the cooperative phase.
__device__ __forceinline__ bool defer_long_b(int4 *list, int b0, int b1, double av, int lane_in_group, int gmask, int threshold)
{
    if (b1 - b0 <= threshold) return false;
    int slot = kLongCap;
    if (lane_in_group == 0) slot = atomicAdd(&list[0].x, 1);
    slot = __shfl(slot, (int)(threadIdx.x & 63) & ~gmask, 64);      // the group leader's answer
    if (slot >= kLongCap) return false;
    if (lane_in_group == 0) {
        const long long bits = __double_as_longlong(av);
        list[1 + slot] = make_int4(b0, b1, (int)(bits & 0xFFFFFFFFll), (int)(bits >> 32));
    }
    return true;
}
__device__ __forceinline__ double long_b_value(const int4 &e) { return __longlong_as_double(((long long)e.w << 32) | (unsigned)e.z); }

// The cooperative phase. The deferred rows are cut into segments of 256 entries (64 lanes × 4 independent loads in flight per
// lane) and the segments are dealt round-robin to the waves of the workgroup, continuing across rows: a few rows of tens of
// thousands of entries (hub columns) and a few hundred rows of a few hundred entries both keep every wave busy, and no wave waits
// a memory round trip per row. body(col, b_value, a_value).
template <bool WITH_VAL, typename Body>
__device__ __forceinline__ void for_deferred_rows(const int4 *list, int t, int threads, const int *__restrict__ bcol,
                                                  const double *__restrict__ bval, Body body)
{
    const int nl = min(list[0].x, kLongCap);
    const int lane = t & 63, wave = t >> 6, waves = threads >> 6;   // waves is a power of two (256, 512 or 1024 threads)
    int dealt = 0;
    for (int i = 0; i < nl; ++i) {
        const int4 e = list[1 + i];
        const double av = WITH_VAL ? long_b_value(e) : 0.0;
        const int last = e.y - 1, nseg = (e.y - e.x + 255) >> 8;
        for (int sg = (wave - dealt) & (waves - 1); sg < nseg; sg += waves) {
            const int k = e.x + (sg << 8) + lane;
            if (k >= e.y) continue;
            const int k1 = min(k + 64, last), k2 = min(k + 128, last), k3 = min(k + 192, last);
            const int c0 = bcol[k], c1 = bcol[k1], c2 = bcol[k2], c3 = bcol[k3];
            double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
            if (WITH_VAL) { v0 = bval[k]; v1 = bval[k1]; v2 = bval[k2]; v3 = bval[k3]; }
            body(c0, v0, av);
            if (k + 64 < e.y) body(c1, v1, av);
            if (k + 128 < e.y) body(c2, v2, av);
            if (k + 192 < e.y) body(c3, v3, av);
        }
        dealt += nseg;
    }
}

// Lanes per A-entry: a power of two near a quarter of the row's average B-row length (flop_i / nnz(A_i)), at most 64 and at most
// the cooperating thread count. These kernels are bound by the number of sequential acol → brpt → bcol rounds, so the more
// A-entries in flight the better; long B rows still get wide coalesced reads.
__device__ __forceinline__ int group_shift(long long flop, int nnz_a, int threads)
{
    if (nnz_a <= 0) return 0;
    // a lane takes ~4 entries of a B row (independent loads, all in flight at once): a quarter of the average B-row length
    const long long avg = (flop + nnz_a - 1) / nnz_a, target = (avg + 3) / 4;
    int g = 0;
    while ((1 << g) < target && g < 6 && (2 << g) <= threads) ++g;
    return g;
}

template <int WGSIZE, int THREADS, int TABLE, bool OPTIMISTIC>
__global__ __launch_bounds__(WGSIZE) void spgemm_symbolic_lds_kernel(
    const int *__restrict__ rows, int nrows, const int *__restrict__ arpt, const int *__restrict__ acol,
    const int *__restrict__ brpt, const int *__restrict__ bcol, const long long *__restrict__ row_flop, int *__restrict__ row_nz,
    int *__restrict__ overflow_rows, int *__restrict__ overflow_count, const int *__restrict__ nrows_dev = nullptr /* not NULL: the row count lives on the device and the grid strides */)
{
    extern __shared__ int lds_i[];
    constexpr int RPB = WGSIZE / THREADS;           // rows per workgroup
    if (nrows_dev) nrows = *nrows_dev;
    for (int blk = blockIdx.x; blk * RPB < nrows; blk += gridDim.x) {
    constexpr int MAX_PROBES = OPTIMISTIC ? 512 : TABLE;   // optimistic tables give up when a probe sequence gets long, i.e. before they are full
    // Dynamic LDS only (a static array in front would break the 16-byte alignment of the int4 list, Guideline 17):
    // [tables: RPB·TABLE ints][long-B list: (kLongCap+1) int4, RPB == 1 only][s_cnt: RPB ints][s_ovf: RPB ints]
    int *s_cnt = lds_i + RPB * TABLE + (RPB == 1 ? 4 * (kLongCap + 1) : 0);
    int *s_ovf = s_cnt + RPB;
    const int sub = threadIdx.x / THREADS, t = threadIdx.x % THREADS;
    const int ridx = blk * RPB + sub;
    const int row = ridx < nrows ? rows[ridx] : -1;
    int *T = lds_i + sub * TABLE;
    int4 *longs = reinterpret_cast<int4 *>(lds_i + RPB * TABLE);   // RPB == 1 kernels only (TABLE·4 bytes is a multiple of 16)
    const int long_thr = RPB == 1 ? long_b_threshold(THREADS) : (1 << 30);
    for (int s = t; s < TABLE; s += THREADS) T[s] = kEmpty;
    if (t == 0) { s_cnt[sub] = 0; s_ovf[sub] = 0; if (RPB == 1) longs[0].x = 0; }
    __syncthreads();
    if (row >= 0) {
        const int a0 = arpt[row], a1 = arpt[row + 1];
        const int gs = group_shift(row_flop[row], a1 - a0, THREADS), gmask = (1 << gs) - 1;
        int cnt = 0;
        bool stop = false;
        auto insert = [&](int key) {
            int h = hash_of<TABLE>(key);
            int probes = 0;
            for (;; ++probes) {
                const int old = atomicCAS(&T[h], kEmpty, key);
                if (old == kEmpty) { cnt++; break; }
                if (old == key) break;
                h = (h + 1) & (TABLE - 1);
                if (probes >= MAX_PROBES) break;                   // never spin: sized tables cannot fill, optimistic ones give up
            }
            if (OPTIMISTIC && probes >= MAX_PROBES) { s_ovf[sub] = 1; stop = true; }   // the table is filling up: give the row to the bitmap path
        };
        walk_a_entries<false>(a0, a1, t >> gs, THREADS >> gs, acol, nullptr, brpt, brpt + 1, [&](int b0, int b1, double) {
            // the abort flag is read by all lanes of the wave in one instruction: the lanes of a group leave together, so the
            // shuffle inside defer_long_b never reads a lane that has already left
            if (OPTIMISTIC && lds_peek(&s_ovf[sub])) return false;
            if (RPB == 1 && defer_long_b(longs, b0, b1, 0.0, t & gmask, gmask, long_thr)) return true;
            for (int k = b0 + (t & gmask); k < b1 && !stop; k += gmask + 1) insert(bcol[k]);
            return true;
        });
        if (RPB == 1) {
            __syncthreads();                                        // RPB == 1: every thread of the workgroup has a row, so this is uniform
            const int nl = min(longs[0].x, kLongCap);
            for (int i = 0; i < nl && !stop; ++i) {
                const int4 e = longs[1 + i];
                for (int k = e.x + t; k < e.y && !stop; k += THREADS) insert(bcol[k]);
                if (OPTIMISTIC && lds_peek(&s_ovf[sub])) stop = true;
            }
        }
        atomicAdd(&s_cnt[sub], cnt);
    }
    __syncthreads();
    if (row >= 0 && t == 0) {
        const int total = s_cnt[sub];
        if (OPTIMISTIC && s_ovf[sub]) overflow_rows[atomicAdd(overflow_count, 1)] = row;   // no abort ⇒ every key was inserted ⇒ the count is exact
        else row_nz[row] = total;
    }
    __syncthreads();                                               // the tables are re-initialised by the next round
    }
}

template <int WGSIZE, int THREADS, int TABLE>
__global__ __launch_bounds__(WGSIZE) void spgemm_numeric_lds_kernel(
    const int *__restrict__ rows, int nrows, const int *__restrict__ arpt, const int *__restrict__ acol, const double *__restrict__ aval,
    const int *__restrict__ brpt, const int *__restrict__ bcol, const double *__restrict__ bval, const long long *__restrict__ row_flop,
    const int *__restrict__ crpt, int *__restrict__ ccol, double *__restrict__ cval, const int *__restrict__ nrows_dev = nullptr /* see the symbolic kernel */)
{
    extern __shared__ int lds_i[];
    constexpr int RPB = WGSIZE / THREADS;
    if (nrows_dev) nrows = *nrows_dev;
    for (int blk = blockIdx.x; blk * RPB < nrows; blk += gridDim.x) {
    const int sub = threadIdx.x / THREADS, t = threadIdx.x % THREADS;
    const int ridx = blk * RPB + sub;
    const int row = ridx < nrows ? rows[ridx] : -1;
    // layout: all fp64 value tables first (8-byte aligned), then the key tables
    double *V = reinterpret_cast<double *>(lds_i) + sub * TABLE;
    int *K = lds_i + RPB * TABLE * 2 + sub * TABLE;
    int4 *longs = reinterpret_cast<int4 *>(lds_i + RPB * TABLE * 3);   // RPB == 1 kernels only (TABLE·12 bytes is a multiple of 16)
    const int long_thr = RPB == 1 ? long_b_threshold(THREADS) : (1 << 30);
    for (int s = t; s < TABLE; s += THREADS) { K[s] = kEmpty; V[s] = 0.0; }
    if (RPB == 1 && t == 0) longs[0].x = 0;
    __syncthreads();
    if (row >= 0) {
        const int a0 = arpt[row], a1 = arpt[row + 1];
        const int gs = group_shift(row_flop[row], a1 - a0, THREADS), gmask = (1 << gs) - 1;
        auto insert = [&](int key, double tv) {
            int h = hash_of<TABLE>(key);
            for (int probes = 0; probes < TABLE; ++probes) {       // bounded: a wrong crpt from the caller must not hang the GPU
                const int old = atomicCAS(&K[h], kEmpty, key);
                if (old == kEmpty || old == key) { atomicAdd(&V[h], tv); break; }   // addop, hash_mult.h:588-593
                h = (h + 1) & (TABLE - 1);
            }
        };
        walk_a_entries<true>(a0, a1, t >> gs, THREADS >> gs, acol, aval, brpt, brpt + 1, [&](int b0, int b1, double av) {
            if (RPB == 1 && defer_long_b(longs, b0, b1, av, t & gmask, gmask, long_thr)) return true;
            for (int k = b0 + (t & gmask); k < b1; k += gmask + 1) insert(bcol[k], av * bval[k]);   // multop, hash_mult.h:583
            return true;
        });
        if (RPB == 1) {
            __syncthreads();                                        // uniform: with RPB == 1 every thread of the workgroup has this row
            for_deferred_rows<true>(longs, t, THREADS, bcol, bval, [&](int col, double bv, double av) { insert(col, av * bv); });
        }
    }
    __syncthreads();
    // sort_and_store_table2mat (hash_mult.h:526-553): compact the occupied slots, ascending keys, store.
    const int off = row >= 0 ? crpt[row] : 0, nz = row >= 0 ? crpt[row + 1] - off : 0;
    if constexpr (RPB == 1) {
        // The table is at most half full and often far emptier (the class spans a 16× range of row sizes): the occupied slots are
        // first gathered into a dense array (order irrelevant before the sort) and only next_pow2(nz) entries are sorted — a
        // 1024-slot table with 50 entries costs 21 compare-exchange rounds of 64 instead of 55 rounds of 1024.
        double *V2 = reinterpret_cast<double *>(lds_i + TABLE * 3 + 4 * (kLongCap + 1));
        int *K2 = reinterpret_cast<int *>(V2 + TABLE / 2);
        int *cnt = K2 + TABLE / 2;
        if (t == 0) *cnt = 0;
        __syncthreads();
        for (int s2 = t; s2 < TABLE; s2 += THREADS) {
            const int key = K[s2];
            if (key != kEmpty) {
                const int pos = atomicAdd(cnt, 1);
                if (pos < TABLE / 2) { K2[pos] = key; V2[pos] = V[s2]; }   // pos >= TABLE/2 only with a wrong crpt from the caller
            }
        }
        int n2 = 1;
        while (n2 < nz) n2 <<= 1;
        n2 = min(n2, TABLE / 2);
        __syncthreads();
        for (int s2 = min(*cnt, TABLE / 2) + t; s2 < n2; s2 += THREADS) K2[s2] = INT_MAX;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < n2; i += THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const int ki = K2[i], kj = K2[ixj];
                        const bool asc = (i & k) == 0;
                        if ((ki > kj) == asc) {
                            K2[i] = kj; K2[ixj] = ki;
                            const double vi = V2[i]; V2[i] = V2[ixj]; V2[ixj] = vi;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int s2 = t; s2 < min(nz, n2); s2 += THREADS) { ccol[off + s2] = K2[s2]; cval[off + s2] = V2[s2]; }
    } else {
        for (int s2 = t; s2 < TABLE; s2 += THREADS) if (K[s2] == kEmpty) K[s2] = INT_MAX;
        __syncthreads();
        for (int k = 2; k <= TABLE; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = t; i < TABLE; i += THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const int ki = K[i], kj = K[ixj];
                        const bool asc = (i & k) == 0;
                        if ((ki > kj) == asc) {
                            K[i] = kj; K[ixj] = ki;
                            const double vi = V[i]; V[i] = V[ixj]; V[ixj] = vi;
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int s2 = t; s2 < nz; s2 += THREADS) { ccol[off + s2] = K[s2]; cval[off + s2] = V[s2]; }
    }
    __syncthreads();                                               // the tables are re-initialised by the next round
    }
}

// ------------------------------------------------------------------------------------------------ big rows, all in LDS
// Rows with 4 K < nz <= 64 K do not fit a keys+fp64 table in LDS, and the HBM bitmap path pays a global atomic per product.
// Instead, one 1024-thread workgroup per row:
//   phase 1 — the row's distinct columns, SORTED, without a sort: per window of 2^20 columns, mark the window's columns in an LDS
//             bitmap (128 KiB), popcount-prefix it and emit the set bits in order into ccol;
//   phase 2 — per chunk of 8192 consecutive output entries: their (sorted) columns in LDS as keys, fp64 accumulators beside them,
//             every product whose column falls in the chunk finds its slot by binary search and is added with ds_add_f64.
// The products are traversed (#windows + #chunks) times; B rows come from L2. No global atomics, sorted output for free.
// Two shapes of the one-row-per-workgroup bitmap kernels, everything derived from the thread count T (a thread owns 32 consecutive bitmap words):
//   T = 1024: windows of 2^20 columns (128 KiB bitmap), value chunks of 8 192 outputs — 136 KiB of LDS, one workgroup per CU (the long rows);
//   T =  256: windows of 2^18 columns (32 KiB), chunks of 2 048 — 41 KiB, three workgroups per CU. The PMC passes of round 2 showed the big shape
//             waiting on dependent loads 60–75 % of its wave-cycles with the fabric at a quarter of its rate: the rows of a few thousand
//             products (most rows) gain nothing from 1 024 threads, and three of them in flight per CU hide each other's latency.
template <int T>
struct BigCfg {
    static_assert(T == 1024 || T == 512 || T == 256, "three shapes");
    static constexpr int kThreads = T;
    static constexpr int kWindowBits = T == 1024 ? 20 : T == 512 ? 19 : 18;   // columns per bitmap window = 1024·T
    static constexpr int kWindowWords = 1 << (kWindowBits - 5);      // = 32·T
    static constexpr int kChunk = 8 * T;                             // output entries per value pass
    static constexpr int kChunkBits = kWindowBits - 7;
    static constexpr int kStage = 5 * T;                             // list items (non-empty 4-word groups) staged in LDS per emit tile; with the T first positions behind it: the union with the flat lists (24·T bytes)
    static constexpr int kPerCu = 1024 / T;                          // workgroups per CU that fit in LDS (136 / 72 / 40 KiB each)
};
constexpr int kFlatUnitsPerRound = G4S_SPGEMM_UPR;   // 64-entry units of B rows a wave loads per round (independent loads in flight per lane)
// which shape takes which row class (G4S_SPGEMM_T_* override them for sweeps)
constexpr int kShapeNumMedium = 256, kShapeSymMedium = 256, kShapeNumLarge = 256, kShapeNumM2 = 256, kShapeNumM3 = 1024, kNumM3Cut = 8192;
inline int shape_of(const char *env, int dflt) { const char *e = getenv(env); const int v = e ? atoi(e) : dflt; return v == 256 || v == 512 ? v : 1024; }
// granularity of the window splits: the column range in 16 pieces (at most — the table holds a position per piece and B row), never finer
// than 2^16 columns; every window size is a multiple, and a value chunk's columns are bracketed by whole pieces
__host__ __device__ __forceinline__ int split_bits(int N)
{
    const int lg = N > 1 ? 32 - __builtin_clz((unsigned)(N - 1)) : 0;   // ceil(log2 N)
    return lg - 4 < 16 ? 16 : (lg - 4 > 20 ? 20 : lg - 4);
}
// Word w of a window lives at LDS slot w ^ (((w >> 6) & 7) << 2): the emit step gives each thread 32 consecutive words and reads them as eight
// 16-byte groups; unswizzled, the lanes of a ds_read_b128 group would all sit on two 16-byte slots of the 256-byte bank row. The XOR moves whole
// 4-word groups (bits 2–4 only), so a group stays one aligned 16-byte read, and within each of the instruction's four 16-lane groups
// (MI355X_MICROARCH.md §LDS) the sixteen lanes land on sixteen different slots.
__device__ __forceinline__ int bm_slot(int w) { return w ^ (((w >> 6) & 7) << 2); }
constexpr int kWindowMaxN = 4 << 20;                  // widest B for which the window kernels take the mid-size rows too
inline int mid_tables() { const char *e = getenv("G4S_SPGEMM_MID_TABLES"); return e ? atoi(e) : 0; }   // A/B: bit 0 the ≤ 1024 class (and the symbolic mid class), 1: ≤ 2048, 2: ≤ 4096 through key tables
inline int window_max_n() { const char *e = getenv("G4S_SPGEMM_WINDOW_MAX_N"); return e ? atoi(e) : kWindowMaxN; }   // tests force the table kernels with 0
// grid of the persistent big-row kernels: one workgroup per CU (their LDS allows no more), fewer when the class is small
inline int big_grid(int nrows, int wgs_per_cu = 1)
{
    // CU count of the CURRENT device (a process may drive several): cached per device id, not once per process
    static int cus_of[64] = {0};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
    if (cus_of[d] == 0) {
        hipDeviceProp_t p;
        cus_of[d] = (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    const int cus = cus_of[d];
    const int g = cus * wgs_per_cu;
    return nrows < g ? nrows : g;
}

// Inclusive prefix sum over the 64 lanes of a wave on the DPP datapath (row shifts inside each 16-lane row, then the row totals
// broadcast down): six full-rate VALU adds, no LDS crossbar traffic as with ds_bpermute-based shuffles.
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned x)
{
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);    // row_shr:8
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 → rows 1 and 3
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 → rows 2 and 3
    return x;
}

// ------------------------------------------------------------------------------------------------ short rows: one wavefront merges the row (round 4)
// Rows of at most kSmallCap products (47 % of the non-empty rows of R-MAT-21, 1.3 % of its products) used to take a 256-thread workgroup each: a 1 K-slot hash
// table, then — numeric — a compaction and a bitonic sort between 45 barriers (2.0 ms for 31 M products; symbolic 0.6 ms). Such a row is a handful of A-entries
// (2.2 on average) pointing at SORTED B rows, so one wavefront merges them instead, no block barrier anywhere:
//   1. the row's products go to LDS run by run (a run = one B row; coalesced loads), values multiplied;
//   2. every product finds its place in the merged order by rank: its index in its own run + the number of smaller columns in every other run (binary
//      searches in LDS; equal columns keep the order of their runs);
//   3. equal neighbours are one output column: the first of them adds the others up — in run order, i.e. in the reference's (j outer, k inner) order
//      (hash_mult.h:583-593), so these rows come out BIT-IDENTICAL to the oracle — and a ballot places the row's columns.
// Duplicate columns inside one B row are merged like any others. A row that does not fit (more than 64 A-entries, or more products than the cap — the numeric
// classes are cut by output length, not by products) is appended to a list that the table kernels take afterwards.
constexpr int kSmallCap = 512;
constexpr int kTinyCap = 64;                                     // the same kernel for the rows of at most 32 products (the tiny class of the symbolic phase): 1 KB of LDS per wave
                                                                  // instead of 7.5 — the CU holds 32 such waves instead of 20, and these rows are all latency (round 5)
template <bool NUMERIC, int CAP = kSmallCap>
constexpr int small_wave_ints() { return CAP + (NUMERIC ? 2 * CAP : 0) + CAP / 2 + CAP / 4; }   // columns | products (fp64) | source index per rank (u16) | run per product (u8)
template <bool NUMERIC, int CAP = kSmallCap>
__global__ __launch_bounds__(256) void spgemm_small_wave_kernel(
    const int *__restrict__ rows, int nrows, const int *__restrict__ arpt, const int *__restrict__ acol, const double *__restrict__ aval,
    const int *__restrict__ brpt, const int *__restrict__ bcol, const double *__restrict__ bval,
    int *__restrict__ row_nz /* symbolic: out */, const int *__restrict__ crpt, int *__restrict__ ccol, double *__restrict__ cval,
    int *__restrict__ ovf_rows, int *__restrict__ ovf_count)
{
    extern __shared__ int lds_i[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *Cin = lds_i + wave * small_wave_ints<NUMERIC, CAP>();
    double *Vin = reinterpret_cast<double *>(Cin + CAP);      // (byte offset a multiple of 8: every region is)
    unsigned short *Src = reinterpret_cast<unsigned short *>(Cin + CAP + (NUMERIC ? 2 * CAP : 0));
    const int ridx = blockIdx.x * 4 + wave;
    if (ridx >= nrows) return;                                     // (no block barrier in this kernel: waves come and go on their own)
    const int row = rows[ridx];
    const int a0 = arpt[row], na = arpt[row + 1] - a0;
    int b0 = 0, len = 0;
    double av = 0.0;
    if (lane < na) {
        const int c = acol[a0 + lane];
        b0 = brpt[c];
        len = brpt[c + 1] - b0;
        if (NUMERIC) av = aval[a0 + lane];
    }
    const int incl = (int)wave_inclusive_sum((unsigned)len), P = incl - len;
    const int flop = __builtin_amdgcn_readlane(incl, 63);
    if (na > 64 || flop > CAP) {                             // uniform
        if (lane == 0) ovf_rows[atomicAdd(ovf_count, 1)] = row;
        return;
    }
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // 1. + 2. flat over the row's PRODUCTS (round 5): lane l of a pass takes product i0 + l, whatever run it sits in — its run by a binary search over the lanes'
    // prefix sums (six shuffles), then one coalesced-as-can-be load per product. The first form walked the runs one after the other, a pass of 64 lanes per run:
    // a row of seven 7-entry runs (a stencil squared: 1.7 M such rows at 120³) spent seven mostly idle passes in step 1 and seven more, each with its own loop
    // over the other runs, in step 2 — 13.9 ms per product for 84 M flop (profiles/r05_small_sizes.txt).
    auto run_of = [&](int i) {                                      // the run e with P[e] <= i < P[e] + len[e] (runs without entries share their successor's P: skipped by the search)
        int lo = 0, hi = na - 1;
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi) >> 1;
            const bool above = __shfl(incl, mid, 64) > i;
            hi = above ? mid : hi;
            lo = above ? lo : min(mid + 1, na - 1);
        }
        return lo;
    };
    unsigned char *Run = reinterpret_cast<unsigned char *>(Src + CAP);   // the run of every product (na <= 64)
    for (int i0 = 0; i0 < flop; i0 += 64) {
        const int i = min(i0 + lane, flop - 1);
        const int e = run_of(i);
        const int rb0 = __shfl(b0, e, 64), rP = __shfl(P, e, 64);
        const int col = bcol[rb0 + (i - rP)];
        double prod = 0.0;
        if (NUMERIC) {                                             // (every lane takes part in the shuffles: a run's owner lane may hold no product of this pass)
            const long long bits = __double_as_longlong(av);
            const int lo32 = __shfl((int)(bits & 0xFFFFFFFFll), e, 64), hi32 = __shfl((int)(bits >> 32), e, 64);
            prod = __longlong_as_double(((long long)hi32 << 32) | (unsigned)lo32) * bval[rb0 + (i - rP)];   // multop, hash_mult.h:583
        }
        if (i0 + lane < flop) {
            Cin[i] = col;
            Run[i] = (unsigned char)e;
            if (NUMERIC) Vin[i] = prod;
        }
    }
    wave_sync();
    // rank of every product in the merged order: its index in its own run + the entries of every other run in front of it
    for (int i0 = 0; i0 < flop; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < flop;
        const int e = valid ? (int)Run[i] : 0, col = valid ? Cin[i] : 0;
        int rank = i - __shfl(P, e, 64);
        for (int e2 = 0; e2 < na; ++e2) {
            const int l2 = __builtin_amdgcn_readlane(len, e2), P2 = __builtin_amdgcn_readlane(P, e2);
            if (l2 == 0) continue;                                 // uniform
            int lo = 0, hi = e2 == e ? 0 : l2;                     // (its own run counts through its index)
            while (__any(lo < hi)) {
                const int mid = (lo + hi) >> 1, v = Cin[P2 + min(mid, l2 - 1)];
                const bool before = e2 < e ? v <= col : v < col;   // entries of run e2 in front of this one: smaller columns, and equal ones of an earlier run
                const bool go = lo < hi;
                lo = (go && before) ? mid + 1 : lo;
                hi = (go && !before) ? mid : hi;
            }
            rank += lo;
        }
        if (valid) Src[rank] = (unsigned short)i;
    }
    wave_sync();
    // 3. equal neighbours → one column; place the columns
    const int off = NUMERIC ? crpt[row] : 0;
    int base = 0;
    for (int i0 = 0; i0 < flop; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < flop;
        const int sidx = valid ? Src[i] : 0, col = Cin[sidx];
        const int prev = (valid && i > 0) ? Cin[Src[i - 1]] : -1;
        const bool head = valid && (i == 0 || col != prev);
        const unsigned long long mask = __ballot(head);
        if (NUMERIC && head) {
            double sum = Vin[sidx];
            for (int m = i + 1; m < flop; ++m) {                   // addop in run order, hash_mult.h:588-593
                const int s2 = Src[m];
                if (Cin[s2] != col) break;
                sum += Vin[s2];
            }
            const int o = off + base + __popcll(mask & ((1ull << lane) - 1ull));
            ccol[o] = col;
            cval[o] = sum;
        }
        base += __popcll(mask);
    }
    if (!NUMERIC && lane == 0) row_nz[row] = base;
}

// Window splits. B's rows are sorted by column, so the entries of row c that fall into bitmap window w are one contiguous piece:
// wsplit[(j − 1)·K + c] = first position of row c whose column is >= j·2^sb (sb = split_bits(N), j = 1 … W − 1; a coarser window is a union of pieces, a finer one a subset of one), computed once per product by
// window_splits_kernel. A window pass (and a value chunk, which spans one or a few windows) then walks only that piece of every B
// row instead of reading the whole row and discarding what is outside — with two windows that halves the products visited.
__global__ void window_splits_kernel(int K, int W, int sbits, const int *__restrict__ brpt, const int *__restrict__ bcol, int *__restrict__ wsplit)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)K * (W - 1)) return;
    const int j = (int)(idx / K) + 1, c = (int)(idx - (long long)(j - 1) * K);
    const int bound = j << sbits;
    int lo = brpt[c], hi = brpt[c + 1];
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (bcol[mid] < bound) lo = mid + 1; else hi = mid;
    }
    wsplit[idx] = lo;
}

// lo/hi arrays (indexed by B row) that bracket the columns [cfirst, clast]; without splits the whole row
__device__ __forceinline__ void window_bounds(const int *brpt, const int *wsplit, int K, int N, int cfirst, int clast, const int *&lo, const int *&hi)
{
    lo = brpt; hi = brpt + 1;
    if (!wsplit) return;
    const int sb = split_bits(N), wf = cfirst >> sb, wl = clast >> sb, W = (N + (1 << sb) - 1) >> sb;
    if (wf > 0) lo = wsplit + (size_t)(wf - 1) * K;
    if (wl < W - 1) hi = wsplit + (size_t)wl * K;
}

#ifdef G4S_PROFILE_BIG
// Section timers of the big-row kernels (tools/big_prof.py, tools/sym_prof.py): s_memtime deltas summed in registers, flushed with one atomic
// per slot by thread 0 of every 16th workgroup at BIG_PROF_FLUSH (a global atomic per stamp would itself be the longest thing in an inner loop).
__device__ unsigned long long g_big_prof[64];                     // [0, 16): numeric big-row kernel; [16, 32): symbolic window kernel and its emit step; [32, 48): rank kernel; [48, 56): its chunks by size
#define BIG_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[16] = {}; constexpr int prof_base = 0
#define BIG_PROF_DECL_SYM unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[16] = {}; constexpr int prof_base = 16
#define BIG_PROF_DECL_RANK unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[16] = {}; constexpr int prof_base = 32
#define BIG_PROF(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof_acc[slot] += n_ - prof_t; prof_t = n_; } while (0)
#ifdef G4S_PROF_LAST_WAVE   /* the report of the LAST wavefront instead of the first: a section that is long in one and a barrier wait in the other is imbalance */
#define BIG_PROF_TID (blockDim.x - 64)
#else
#define BIG_PROF_TID 0
#endif
#define BIG_PROF_FLUSH do { if (threadIdx.x == BIG_PROF_TID && (blockIdx.x & 15) == 0) {   /* one workgroup in 16 reports: the flush atomics of all of them would sit in front of the next loads */ _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) if (prof_acc[s_]) atomicAdd(&g_big_prof[prof_base + s_], prof_acc[s_]); } _Pragma("unroll") for (int s_ = 0; s_ < 16; ++s_) prof_acc[s_] = 0; } while (0)
#else
#define BIG_PROF_DECL
#define BIG_PROF_DECL_SYM
#define BIG_PROF_DECL_RANK
#define BIG_PROF(slot)
#define BIG_PROF_FLUSH
#endif

// LDS beside the bitmap in the one-row-per-workgroup kernels: [ctrl: 32 ints][scan: 64 ints][union of the emit stage and the flat lists]
// (T = 1024: 128 KiB + 24.4 KiB of the CU's 160).
template <int T>
struct BigSide {
    static constexpr int kCtrlInts = 32, kScanInts = 64, kTotalSlot = 24;   // ctrl[0 … 16): wave totals of the unit scan; ctrl[24]: the row's running count
    static constexpr int kUnitBatch = 2 * T;                                 // units mapped per batch (u16 entry index each)
    static constexpr size_t kFlatBytes = sizeof(int4) * T + sizeof(int) * (T + 4) + sizeof(unsigned short) * kUnitBatch;
    static constexpr size_t kStageBytes = sizeof(int) * (BigCfg<T>::kStage + T);   // + the emit step's per-thread first positions
    static constexpr size_t kBytes = sizeof(int) * (kCtrlInts + kScanInts) + (kFlatBytes > kStageBytes ? kFlatBytes : kStageBytes);
    int *ctrl, *scan, *stage, *P;
    int4 *E;            // per A-entry: its B row [x, y) and the bits of its value
    unsigned short *M;  // per unit of the batch: its A-entry
    __device__ __forceinline__ explicit BigSide(int *base_)
    {
        ctrl = base_; scan = base_ + kCtrlInts; stage = scan + kScanInts;
        E = reinterpret_cast<int4 *>(stage); P = reinterpret_cast<int *>(E + T); M = reinterpret_cast<unsigned short *>(P + T + 4);
    }
};
template <int T>
constexpr size_t big_lds_bytes() { return sizeof(unsigned) * BigCfg<T>::kWindowWords + BigSide<T>::kBytes; }

// The products of one row, flat. The rows of these classes are a few A-entries pointing at long B rows (R-MAT scale 21: a mean of 6 … 90
// entries per row, 90 % of the products in B rows of more than 256 entries), so lanes-per-A-entry groups leave most of the workgroup idle
// behind a handful of serial load → insert round trips. Instead:
//   1. entry pass — thread j takes A-entry j: (b0, b1) of its B row clipped to [lo, hi), its value, and its number of UNITS (64 consecutive
//      entries of the B row) go to LDS; a block scan of the unit counts gives every entry its first unit;
//   2. map — the units are numbered across the row; thread u finds unit u's entry by binary search over the scan (u16 per unit);
//   3. rounds — wave w takes UPR consecutive units at a time (UPR·64 products: UPR independent coalesced loads per lane in flight), the
//      waves striding the unit range, so every wave does the same work whatever the B-row lengths are.
// body(col[UPR], b_value[UPR], a_value[UPR], valid[UPR]) gets a lane's products of one round together, so that it can interleave their
// dependent LDS chains. Contains barriers: call from uniform control flow; on return every product has been handed to body.
struct FlatNoTail { __device__ __forceinline__ void operator()() const {} };
template <int T, bool WITH_VAL, int UPR, typename Body, typename Tail = FlatNoTail>
__device__ __forceinline__ void flat_products(int a0, int a1, const int *__restrict__ acol, const double *__restrict__ aval,
                                              const int *__restrict__ lo, const int *__restrict__ hi, const int *__restrict__ bcol,
                                              const double *__restrict__ bval, const BigSide<T> &sd, int t, Body body, Tail tail = Tail())
{
    // tail(): called once by every thread in front of the LAST barrier of the walk (or at the end when there is none): loads issued there have the
    // barrier's skew to arrive in (the numeric kernel starts the column-id gather of its store step there)
    bool tail_done = false;
    constexpr int kWaves = T / 64, kUB = BigSide<T>::kUnitBatch;
    const int lane = t & 63, wave = t >> 6;
    BIG_PROF_DECL;
    for (int e0 = a0; e0 < a1; e0 += T) {                           // tiles of T A-entries (one tile for all but hub-like rows of A)
        const int ne = min(T, a1 - e0);
        int units = 0;
        if (t < ne) {
            const int c = acol[e0 + t];
            const int b0 = lo[c], b1 = hi[c];
            const long long bits = WITH_VAL ? __double_as_longlong(aval[e0 + t]) : 0ll;
            sd.E[t] = make_int4(b0, b1, (int)(bits & 0xFFFFFFFFll), (int)(bits >> 32));
            units = b1 > b0 ? (b1 - b0 + 63) >> 6 : 0;
        }
        const int incl = (int)wave_inclusive_sum((unsigned)units);
        if (lane == 63) sd.ctrl[wave] = incl;
        __syncthreads();
        if (WITH_VAL) BIG_PROF(11);
        int first = incl - units, nu = 0;
#pragma unroll
        for (int u = 0; u < kWaves; ++u) {
            const int v = sd.ctrl[u];
            if (u < wave) first += v;
            nu += v;
        }
        if (t < ne) sd.P[t] = first;
        if (t == 0) sd.P[ne] = nu;
        __syncthreads();
        if (WITH_VAL) BIG_PROF(12);
        for (int ub0 = 0; ub0 < nu; ub0 += kUB) {
            const int nb = min(kUB, nu - ub0);
            for (int u = t; u < nb; u += T) {                       // the last entry j with P[j] <= unit (entries without units share their successor's P)
                const int unit = ub0 + u;
                int l = 0, h = ne;
                while (h - l > 1) {
                    const int mid = (l + h) >> 1;
                    if (sd.P[mid] <= unit) l = mid; else h = mid;
                }
                sd.M[u] = (unsigned short)l;
            }
            __syncthreads();
            if (WITH_VAL) BIG_PROF(13);
            for (int g = wave * UPR; g < nb; g += kWaves * UPR) {
                int kk[UPR], c[UPR];
                bool ok[UPR];
                double av[UPR], v[UPR];
                // (each stage for all UPR units before the next: written unit by unit, every LDS read was followed by its own wait)
                int uu[UPR], j[UPR], p0[UPR];
                int4 e[UPR];
#pragma unroll
                for (int q = 0; q < UPR; ++q) { uu[q] = min(g + q, nb - 1); j[q] = sd.M[uu[q]]; }
#pragma unroll
                for (int q = 0; q < UPR; ++q) { e[q] = sd.E[j[q]]; p0[q] = sd.P[j[q]]; }
#pragma unroll
                for (int q = 0; q < UPR; ++q) {
                    const int k = e[q].x + ((ub0 + uu[q] - p0[q]) << 6) + lane;
                    ok[q] = g + q < nb && k < e[q].y;
                    kk[q] = min(k, e[q].y - 1);                     // a unit exists only in a non-empty piece: e.y - 1 >= e.x
                    av[q] = WITH_VAL ? long_b_value(e[q]) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < UPR; ++q) {
                    c[q] = bcol[kk[q]];
                    v[q] = WITH_VAL ? bval[kk[q]] : 0.0;
                }
#ifdef G4S_PROFILE_BIG
                if (WITH_VAL) { BIG_PROF(2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BIG_PROF(4); }
#endif
                body(c, v, av, ok);
#ifdef G4S_PROFILE_BIG
                if (WITH_VAL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); BIG_PROF(5); }
#endif
            }
            if (e0 + T >= a1 && ub0 + kUB >= nu) { tail(); tail_done = true; }   // uniform: the last batch of the last tile
            __syncthreads();                                        // M — and after the last batch E and P — are rewritten next
            if (WITH_VAL) BIG_PROF(14);
        }
    }
    if (!tail_done) tail();
    BIG_PROF_FLUSH;
}

// Emits the set bits of one LDS bitmap window (words swizzled by bm_slot) in ascending column order to out[0 … total) and returns
// total (the same value in every thread). Thread t owns the 32 consecutive words [32t, 32t + 32), read as eight 16-byte groups (128 columns each); a
// block scan of the per-thread (set bits, non-empty groups) places its columns and its groups. The bits themselves are written group by group from a
// list of the non-empty GROUPS, one item per thread: in a power-law row the first few hundred columns are all present, and a thread emitting its own
// 1 024 columns would write a thousand ids while the rest write a handful (measured in round 1: 38 % of the numeric kernel; round 4 tried it again with
// an LDS stage and a wave-cooperative path for crowded threads: 2–3× slower than the list, the per-thread loops diverge on every word).
// Round 4: list items are 4-word groups instead of words — the list build is 8 steps per thread instead of 32 (it was 25 % of the symbolic window
// kernels), an item's columns are at most 128 consecutive ids, and the words come in as 16-byte LDS reads.
// Every group that holds a bit is written back as zero when its item is read: the bitmap is clean again when the call returns.
// s_scan: 32 ints of LDS scratch; stage: BigCfg<T>::kStage ints (list items), followed by T ints (per-thread first positions).
// Contains barriers: call from uniform control flow; the caller puts a barrier between this call and the next write to the bitmap.
template <int T>
__device__ __forceinline__ int emit_window_columns(unsigned *bm, int w0, int *__restrict__ out, int *s_scan, int *stage, int t)
{
    static_assert(BigCfg<T>::kWindowWords / T == 32, "emit layout");
    constexpr int S = BigCfg<T>::kStage;
    int *first_pos = stage + S;
    const int lane = t & 63, wave = t >> 6;
    const int kq = (t >> 1) & 7;                                    // bm_slot's XOR for this thread's block, in 4-word groups ((32t + i) >> 6 == t >> 1)
    BIG_PROF_DECL_SYM;
    unsigned gc_lo = 0, gc_hi = 0;                                  // set bits of groups 0–3 / 4–7, one byte each (a group holds at most 128)
    int cnt = 0, ng = 0;
    {
        const uint4 *blk = reinterpret_cast<const uint4 *>(bm + t * 32);
        uint4 g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = blk[q ^ kq];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = __popc(g[q].x) + __popc(g[q].y) + __popc(g[q].z) + __popc(g[q].w);
            cnt += c;
            ng += c != 0;
            if (q < 4) gc_lo |= (unsigned)c << (8 * q); else gc_hi |= (unsigned)c << (8 * (q - 4));
        }
    }
    BIG_PROF(8);
    const int incl_c = (int)wave_inclusive_sum((unsigned)cnt), incl_g = (int)wave_inclusive_sum((unsigned)ng);
    if (lane == 63) { s_scan[wave] = incl_c; s_scan[16 + wave] = incl_g; }
    __syncthreads();
    BIG_PROF(9);
    int p = incl_c - cnt, gq = incl_g - ng, total, total_g;          // first column / first list item of this thread within the window
    {
        // the wave totals: lanes 0 … T/64 − 1 read one each and a wave scan adds them (every thread reading all of them was 2·T/64 LDS reads per thread: the
        // CU issues one LDS instruction at a time, and 16 waves × 32 of them stood in front of the list build)
        constexpr int kW = T / 64;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const unsigned sc = (unsigned)wave_inclusive_sum(lane < kW ? (unsigned)s_scan[lane] : 0u);
        const unsigned sg = (unsigned)wave_inclusive_sum(lane < kW ? (unsigned)s_scan[16 + lane] : 0u);
        total = (int)__builtin_amdgcn_readlane(sc, kW - 1);
        total_g = (int)__builtin_amdgcn_readlane(sg, kW - 1);
        if (wv > 0) { p += (int)__builtin_amdgcn_readlane(sc, wv - 1); gq += (int)__builtin_amdgcn_readlane(sg, wv - 1); }
    }
    first_pos[t] = p;                                              // a list item is (group << 10 | offset from its owner's first column)
    for (int tile0 = 0; tile0 < total_g; tile0 += S) {              // uniform; one tile unless more than S groups hold columns
        const int tile1 = tile0 + S;
        if (gq < tile1 && gq + ng > tile0) {
            int q = 0, j = gq;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (int)(((i < 4 ? gc_lo : gc_hi) >> (8 * (i & 3))) & 0xffu);
                if (c) {
                    if (j >= tile0 && j < tile1) stage[j - tile0] = ((t * 8 + i) << 10) | q;
                    q += c;                                        // < 1024 before the thread's last group
                    ++j;
                }
            }
        }
        __syncthreads();
        BIG_PROF(10);
        const int n = min(S, total_g - tile0);
        // Two of a thread's items go through the three dependent LDS reads together — item, its group's words, its owner's first position — then through the bit loops
        // (1, 2 and all S / T at once measured within 2 % of each other). What the step costs is instruction issue: 150 M wave-iterations of the bit loops per product
        // for 1.9 G columns (13 active lanes each; a group holds 3 columns on average, 57 % of the groups one) at ≈ 12 VALU instructions per iteration. Round 4 also
        // tried a lane walking all its items as ONE queue of bits (fewer wave-iterations: max over lanes of a sum instead of a sum of maxima): each iteration then
        // carries the queue's selects, 25–30 instructions — 8.2 / 5.6 ms against 6.2 / 4.1 for the two window shapes.
        constexpr int kI = 2;
        for (int e0 = 0; e0 < n; e0 += kI * T) {
        unsigned item[kI];
        uint4 g[kI];
        int pos[kI];
#pragma unroll
        for (int k = 0; k < kI; ++k) item[k] = (unsigned)stage[min(e0 + t + k * T, max(n, 1) - 1)];
#pragma unroll
        for (int k = 0; k < kI; ++k) {
            const int gid = (int)(item[k] >> 10), tt = gid >> 3;    // owner thread and its group
            uint4 *gp = reinterpret_cast<uint4 *>(bm + tt * 32) + ((gid & 7) ^ ((tt >> 1) & 7));
            const bool mine = e0 + t + k * T < n;
            g[k] = mine ? *gp : make_uint4(0u, 0u, 0u, 0u);
            if (mine) *gp = make_uint4(0u, 0u, 0u, 0u);             // the window leaves the bitmap clean: the next one does not zero 32·T words first (12 % of the symbolic window kernels)
            pos[k] = first_pos[tt] + (int)(item[k] & 0x3ffu);
        }
#pragma unroll
        for (int k = 0; k < kI; ++k) {
            const int col0 = w0 + (int)((item[k] >> 10) << 7);
            int ps = pos[k];
            unsigned long long b = ((unsigned long long)g[k].y << 32) | g[k].x;
            while (b) {
                const int bit = __ffsll((long long)b) - 1;
                b &= b - 1;
                out[ps++] = col0 + bit;
            }
            b = ((unsigned long long)g[k].w << 32) | g[k].z;
            while (b) {
                const int bit = __ffsll((long long)b) - 1;
                b &= b - 1;
                out[ps++] = col0 + 64 + bit;
            }
        }
        }
        BIG_PROF(13);
        __syncthreads();
        BIG_PROF(11);
    }
    __syncthreads();                                               // (first_pos of a window without groups: nobody may still be reading it when the next call writes it)
    BIG_PROF(12);
    BIG_PROF_FLUSH;
    return total;
}

// Symbolic twin of phase 1 below: the number of distinct columns of a row whose key table would not fit LDS, counted with the
// same LDS bitmap windows (no hash table that can overflow, no HBM bitmap, no global atomics). One workgroup per row.
// pre_off / pre_cols (one-shot call only): rows with pre_off[row] >= 0 also write their sorted distinct columns to
// pre_cols[pre_off[row] …], so that the numeric phase does not have to mark and emit them a second time.
// out_rpt (the numeric phase's emit pass, round 4): the rows that carry NO columns (pre_off NULL or pre_off[row] < 0) write theirs to
// pre_cols[out_rpt[row] …] — i.e. into ccol at the row's own offset — and the rows that do carry them are skipped; row_nz may be NULL.
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void spgemm_symbolic_window_kernel(
    const int *__restrict__ rows, int nrows, int *__restrict__ next_row /* not NULL: rows handed out one at a time (list sorted longest first) */,
    int N, int K, const int *__restrict__ wsplit, const int *__restrict__ arpt, const int *__restrict__ acol,
    const int *__restrict__ brpt, const int *__restrict__ bcol, const long long *__restrict__ row_flop, int *__restrict__ row_nz,
    const long long *__restrict__ pre_off, int *__restrict__ pre_cols, const int *__restrict__ out_rpt, int nz_lo, int nz_hi)
{
    constexpr int kBigThreads = T, kBigWindowBits = BigCfg<T>::kWindowBits, kBigWindowWords = BigCfg<T>::kWindowWords;
    extern __shared__ int lds_i[];                                 // dynamic only (Guideline 17): the layout of the numeric big-row kernel
    unsigned *bm = reinterpret_cast<unsigned *>(lds_i);
    const BigSide<T> sd(lds_i + kBigWindowWords);
    int &s_total = sd.ctrl[BigSide<T>::kTotalSlot];
    const int t = threadIdx.x;
    // Persistent: the grid is a few workgroups per CU (as many as their LDS allows) and every workgroup walks its share of the class's rows —
    // starting a 1024-thread workgroup with this much LDS costs several µs, and a class holds 10^5 rows of a few thousand products each.
    for (int ridx = blockIdx.x;; ridx += gridDim.x) {
    if (next_row) {                                                // uniform
        if (t == 0) sd.ctrl[28] = atomicAdd(next_row, 1);
        __syncthreads();
        ridx = sd.ctrl[28];
        __syncthreads();
    }
    if (ridx >= nrows) break;
    const int row = rows[ridx];
    const int a0 = arpt[row], a1 = arpt[row + 1];
    long long po = pre_off ? pre_off[row] : -1;                    // uniform
    if (out_rpt) {                                                 // emit pass of the numeric phase: only the rows of this launch's size range without carried columns
        const int onz = out_rpt[row + 1] - out_rpt[row];
        if (po >= 0 || onz <= nz_lo || onz > nz_hi) continue;      // (nobody has touched LDS or a barrier for this row yet)
        po = out_rpt[row];
    }
    if (t == 0) s_total = 0;
    BIG_PROF_DECL_SYM;
    for (int w0 = 0; w0 < N; w0 += (1 << kBigWindowBits)) {
        for (int i = t; i < kBigWindowWords / 4; i += kBigThreads) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        BIG_PROF(0);
        const int w1 = min(N, w0 + (1 << kBigWindowBits));
        const int *wlo, *whi;
        window_bounds(brpt, wsplit, K, N, w0, w1 - 1, wlo, whi);
        flat_products<T, false, kFlatUnitsPerRound>(a0, a1, acol, nullptr, wlo, whi, bcol, nullptr, sd, t,
            [&](const int (&col)[kFlatUnitsPerRound], const double (&)[kFlatUnitsPerRound], const double (&)[kFlatUnitsPerRound], const bool (&ok)[kFlatUnitsPerRound]) {
#pragma unroll
                for (int q = 0; q < kFlatUnitsPerRound; ++q)
                    if (ok[q] && col[q] >= w0 && col[q] < w1) atomicOr(&bm[bm_slot((col[q] - w0) >> 5)], 1u << ((col[q] - w0) & 31));
            });
        BIG_PROF(1);
        if (po >= 0) {
            const int total = emit_window_columns<T>(bm, w0, pre_cols + po + s_total, sd.scan, sd.stage, t);
            if (t == 0) s_total += total;
            BIG_PROF(3);
        } else {
            int cnt = 0;
            for (int i = t; i < kBigWindowWords; i += kBigThreads) cnt += __popc(bm[i]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
            if ((t & 63) == 0 && cnt) atomicAdd(&s_total, cnt);
        }
        __syncthreads();
    }
    if (t == 0 && row_nz) row_nz[row] = s_total;
    __syncthreads();
    BIG_PROF(6);
    BIG_PROF_FLUSH;
    }
}


// ---- the symbolic window kernel on unit lists (round 4; the kernel above stays for the numeric phase's emit pass and as the fall-back).
// Same idea as the numeric kernel's unit lists (unit_kernel below): the (row, window) walk used to open with an entry pass, a block scan and a unit map —
// three barriers and a dependent load chain before the first column could be requested, once per window. A pre-pass (sym_unit_kernel) writes the units of every
// (list position i, window w, A-entry e), ordered (i, w, e): the piece of B row acol[e] inside window w, cut into runs of at most 64 entries. The kernel reads
// them with scalar loads, round 0 in front of the bitmap's zeroing, and takes its rows' metadata one row ahead.
// A row's metadata for the persistent window kernels, packed per LIST POSITION by a pre-pass (round 4): the kernels used to chase ticket → rows[] → arpt / crpt /
// pre_off → item_off → uoff with scalar loads, four dependent round trips that every wavefront of the workgroup sat through once per row (≈ 1.2 µs × 430 rows per
// workgroup in the long class, × 205 in the mid-size one); one 48-byte (numeric) / 32-byte (symbolic) scalar load now.
struct __attribute__((aligned(16))) NumRowMeta { int row, a0, a1, off, nz, u0, u1, pad; long long po, ioff; };
struct __attribute__((aligned(16))) SymRowMeta { int row, na, u0, u1; long long po, ioff; };
__global__ void num_row_meta_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const int *__restrict__ crpt, const long long *__restrict__ pre_off,
                                    const long long *__restrict__ item_off /* NULL: no unit lists */, const int *__restrict__ uoff, NumRowMeta *__restrict__ meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    NumRowMeta m;
    m.row = r; m.a0 = arpt[r]; m.a1 = arpt[r + 1]; m.off = crpt[r]; m.nz = crpt[r + 1] - m.off; m.pad = 0;
    m.po = pre_off ? pre_off[r] : -1;
    m.ioff = 0; m.u0 = 0; m.u1 = 0;
    if (item_off) {
        const long long total_items = item_off[n];
        m.ioff = item_off[i]; m.u0 = uoff[m.ioff]; m.u1 = uoff[min(m.ioff + (m.a1 - m.a0), total_items)];   // the units of the row's first chunk (a row outside the launch's size range has no items)
    }
    meta[i] = m;
}
__global__ void sym_row_meta_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const long long *__restrict__ pre_off,
                                    const long long *__restrict__ item_off, const int *__restrict__ uoff, SymRowMeta *__restrict__ meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    const long long total_items = item_off[n];
    SymRowMeta m;
    m.row = r; m.na = arpt[r + 1] - arpt[r]; m.po = pre_off ? pre_off[r] : -1; m.ioff = item_off[i];
    m.u0 = uoff[m.ioff]; m.u1 = uoff[min(m.ioff + m.na, total_items)];
    meta[i] = m;
}

struct __attribute__((aligned(8))) SymUnit { int bpos, len; };
struct __attribute__((aligned(16))) UnitDesc { int bpos, len, av_lo, av_hi; };   // a numeric unit (unit_kernel below), 16 bytes: one s_load_dwordx4
#include "spgemm_rank.hpp"
__global__ void sym_items_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, int nwin, long long *__restrict__ items)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) items[i] = i < n ? (long long)(arpt[rows[i] + 1] - arpt[rows[i]]) * nwin : 0;
}
template <bool EXPAND>
__global__ __launch_bounds__(256) void sym_unit_kernel(long long bound /* threads launched: an upper bound of the item count */, int n, const int *__restrict__ rows,
                                                       const long long *__restrict__ item_off /* n + 1 */, const int *__restrict__ arpt, const int *__restrict__ acol,
                                                       const int *__restrict__ brpt, int K, int N, const int *__restrict__ wsplit, int window_bits,
                                                       int *__restrict__ ucount /* !EXPAND: out, bound + 1 entries */, const int *__restrict__ uoff, SymUnit *__restrict__ U)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x, total = item_off[n];
    if (i > bound) return;
    if (i >= total) { if constexpr (!EXPAND) ucount[i] = 0; return; }   // (the scan runs over bound + 1 entries)
    int rl = 0, rh = n;                                             // the list position that holds the workgroup's first item (uniform), then a short walk forward
    while (rl < rh) {
        const int mid = rl + ((rh - rl) >> 1);
        if (item_off[mid + 1] > i0) rh = mid; else rl = mid + 1;
    }
    while (item_off[rl + 1] <= i) ++rl;
    const int row = rows[rl], a0 = arpt[row], na = arpt[row + 1] - a0;
    const long long idx = i - item_off[rl];
    const int w = (int)(idx / na), e = (int)(idx - (long long)w * na);
    const int c = acol[a0 + e];
    int lo = brpt[c], hi = brpt[c + 1];
    if (wsplit) {                                                  // the piece of the row inside window w (window_bounds)
        const int sb = split_bits(N), W = (N + (1 << sb) - 1) >> sb;
        const int cf = w << window_bits, cl = min(N, (w + 1) << window_bits) - 1, wf = cf >> sb, wl = cl >> sb;
        if (wf > 0) lo = wsplit[(size_t)(wf - 1) * K + c];
        if (wl < W - 1) hi = wsplit[(size_t)wl * K + c];
    }
    if constexpr (!EXPAND) ucount[i] = hi > lo ? (hi - lo + 63) >> 6 : 0;
    else {
        SymUnit *dst = U + uoff[i];
        for (int k = lo; k < hi; k += 64) *dst++ = SymUnit{k, min(64, hi - k)};
    }
}

// CUTS (round 5, spgemm_rank.hpp): pre_off / pre_cols are the rows' cut offsets and the cut array — the kernel counts and writes a row's cuts instead of its columns.
#ifndef G4S_SPGEMM_SYM_STREAM_GROUP
#define G4S_SPGEMM_SYM_STREAM_GROUP 8
#endif
constexpr int kSymStreamGroup = G4S_SPGEMM_SYM_STREAM_GROUP;
template <int T, bool CUTS>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void spgemm_symbolic_units_kernel(
    const int *__restrict__ rows, int nrows, int *__restrict__ next_row /* not NULL: rows handed out one at a time (list sorted longest first) */,
    int N, const int *__restrict__ arpt, const int *__restrict__ bcol, int *__restrict__ row_nz,
    const long long *__restrict__ pre_off, int *__restrict__ pre_cols,
    const long long *__restrict__ item_off, const int *__restrict__ uoff, const SymUnit *__restrict__ U, const SymRowMeta *__restrict__ meta /* per list position (sym_row_meta_kernel) */)
{
    constexpr int kBigWindowBits = BigCfg<T>::kWindowBits, kBigWindowWords = BigCfg<T>::kWindowWords, kU = kFlatUnitsPerRound, kWaves = T / 64;
    extern __shared__ int lds_i[];
    unsigned *bm = reinterpret_cast<unsigned *>(lds_i);
    const BigSide<T> sd(lds_i + kBigWindowWords);
    int &s_total = sd.ctrl[BigSide<T>::kTotalSlot];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    struct RowMeta { int row, na, u0, u1; long long po, ioff; };
    const long long total_items = item_off[nrows];
    auto load_meta = [&](int idx) {                                 // uniform index: one 32-byte scalar load
        const int ci = min(idx, nrows - 1);                        // (a ticket past the end reads the last row: never used)
        const SymRowMeta p = meta[ci];
        RowMeta m;
        m.row = p.row; m.na = p.na; m.po = p.po; m.ioff = p.ioff; m.u0 = p.u0; m.u1 = p.u1;
        return m;
    };
    int ridx = blockIdx.x, nridx = 0;
    if (next_row) {                                                // uniform
        if (t == 0) sd.ctrl[28] = atomicAdd(next_row, 1);
        __syncthreads();
        ridx = __builtin_amdgcn_readfirstlane(sd.ctrl[28]);
        __syncthreads();
    }
    RowMeta cur = load_meta(ridx), nxt = cur;
    const int nwin = (N + (1 << kBigWindowBits) - 1) >> kBigWindowBits;
    for (int i = t; i < kBigWindowWords / 4; i += T) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0u, 0u, 0u, 0u);   // once: every window leaves the bitmap clean
    __syncthreads();
    // Round 0 of the NEXT window — of this row, or the first of the workgroup's next row — is requested behind the current window's mark barrier and arrives under
    // its count / emit step (round 5): requested at the top of its own window, every window of every row began with an exposed ≈ 2 µs round trip (unit descriptors,
    // then the columns), four times per row in the mid-size class.
    int c[kU];
    bool ok[kU];
    bool have = false;                                             // c / ok hold round 0 of the window about to be marked
    // A window's units are DEALT to the waves (unit i → wave i % kWaves; the wave's k-th unit is k·kWaves + wave), as in the rank kernel: in blocks of kU the
    // waves of a window with a few units more than a multiple of kU·kWaves differed by kU units, and the window lasts as long as its busiest wave.
    auto request_round = [&](int u0, int nu, int (&cc)[kU], bool (&okk)[kU]) {   // the wave's first kU units
        SymUnit d[kU];                                              // all of the round's descriptors first (see the numeric kernel's request_round)
#pragma unroll
        for (int q = 0; q < kU; ++q) d[q] = U[u0 + min(q * kWaves + wave, max(nu, 1) - 1)];   // uniform: one s_load_dwordx2 each (a window without units reads a neighbour's or the pad entry: not used)
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const int len = nu > 0 ? d[q].len : 1, bpos = nu > 0 ? d[q].bpos : 0;
            okk[q] = q * kWaves + wave < nu && lane < len;
            cc[q] = bcol[bpos + min(lane, len - 1)];
        }
    };
    for (; ridx < nrows; ridx = nridx, cur = nxt) {
        if (next_row) { if (t == 0) sd.ctrl[29] = atomicAdd(next_row, 1); }   // read below, behind the first barrier of the row
        else { nridx = ridx + gridDim.x; nxt = load_meta(nridx); }
        const long long po = cur.po;                               // uniform
        const int na = cur.na;
        if (t == 0) s_total = 0;
        int cu0 = cur.u0, cu1 = cur.u1, cu2 = 0;
        BIG_PROF_DECL_SYM;
        for (int wi = 0; wi < nwin; ++wi) {
            const int w0 = wi << kBigWindowBits, w1 = min(N, w0 + (1 << kBigWindowBits));
            const int nu = cu1 - cu0;
            auto mark_round = [&](const int (&cc)[kU], const bool (&okk)[kU]) {
#pragma unroll
                for (int q = 0; q < kU; ++q)
                    if (okk[q] && cc[q] >= w0 && cc[q] < w1) atomicOr(&bm[bm_slot((cc[q] - w0) >> 5)], 1u << ((cc[q] - w0) & 31));
            };
            if (!have) request_round(cu0, nu, c, ok);              // (the workgroup's first window only)
            cu2 = uoff[min(cur.ioff + (long long)min(wi + 2, nwin) * na, total_items)];   // (consumed a window later)
            BIG_PROF(0);
            mark_round(c, ok);                                     // (the bitmap is clean: zeroed at the kernel's start, left clean by every window since)
            // the wave's units past the first kU, streamed (stream_unit_groups, spgemm_rank.hpp): 64 descriptors per vector load (lane e = the e-th of the batch),
            // the columns in two alternating groups. (Round by round — descriptors, wait, columns, wait, marks — a hub row's window of thousands of units paid two
            // memory round trips per kU units.)
            {
                const int mine = __builtin_amdgcn_readfirstlane((nu - wave + kWaves - 1) / kWaves);   // the wave's units in this window
                for (int eb = 0;; eb += 64) {
                    const int ne = max(0, min(64, mine - kU - eb));
                    if (ne <= 0) break;
                    const int2 dx = reinterpret_cast<const int2 *>(U)[cu0 + min((kU + eb + lane) * kWaves + wave, nu - 1)];
                    stream_unit_groups<kSymStreamGroup, int>(ne,
                        [&](int (&cc)[kSymStreamGroup], int e0) {
#pragma unroll
                            for (int q = 0; q < kSymStreamGroup; ++q) { const int e = min(e0 + q, ne - 1); cc[q] = bcol[__builtin_amdgcn_readlane(dx.x, e) + min(lane, __builtin_amdgcn_readlane(dx.y, e) - 1)]; }
                        },
                        [&](int (&cc)[kSymStreamGroup], int e0) {
#pragma unroll
                            for (int q = 0; q < kSymStreamGroup; ++q)
                                if (e0 + q < ne && lane < __builtin_amdgcn_readlane(dx.y, e0 + q) && cc[q] >= w0 && cc[q] < w1) atomicOr(&bm[bm_slot((cc[q] - w0) >> 5)], 1u << ((cc[q] - w0) & 31));
                        });
                    if (ne < 64) break;
                }
            }
            __syncthreads();
            BIG_PROF(1);
            if (wi == 0 && next_row) { nridx = __builtin_amdgcn_readfirstlane(sd.ctrl[29]); nxt = load_meta(nridx); }   // (nobody writes the slot again before this row's last barrier)
            {   // the next window's round 0 (this row's, or the next row's first): c / ok are free — every mark of this window has been issued
                const bool last_win = wi + 1 >= nwin;              // uniform
#ifdef G4S_SYM_NO_PREFETCH
                have = false;                                      // (A/B build: every window requests its own round 0 at its top, as until round 5)
#else
                have = !last_win || nridx < nrows;
#endif
                const int n0 = last_win ? nxt.u0 : cu1, n1 = last_win ? nxt.u1 : cu2;
                if (have) request_round(n0, n1 - n0, c, ok);
            }
            if (CUTS && po >= 0) {
                const int nseg = rank_segments(N), before = s_total;   // (s_total: written by thread 0 behind the barrier at the end of the previous window)
                const int total = count_and_cut_window<T>(bm, wi, before, nseg, pre_cols + po, pre_cols + po + nseg, sd.scan, t);
                if (t == 0) s_total = before + total;
                BIG_PROF(3);
            } else if (po >= 0) {
                const int total = emit_window_columns<T>(bm, w0, pre_cols + po + s_total, sd.scan, sd.stage, t);
                if (t == 0) s_total += total;
                BIG_PROF(3);
            } else {
                int cnt = 0;
                for (int i = t; i < kBigWindowWords; i += T) { cnt += __popc(bm[i]); bm[i] = 0u; }   // count and clean
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
                if (lane == 0 && cnt) atomicAdd(&s_total, cnt);
            }
            __syncthreads();
            cu0 = cu1; cu1 = cu2;
        }
        if (t == 0) row_nz[cur.row] = s_total;
        __syncthreads();
        BIG_PROF(6);
        BIG_PROF_FLUSH;
    }
}


// Exact splits for the value chunks of rows with more than one chunk (one-shot call: the sorted columns are known before the numeric phase).
// A chunk holds kChunk consecutive output columns; with the 16 window pieces as the only splits a chunk walked every B-row piece its column range
// touches — three units visited per unit used, flop-weighted, on R-MAT-21 (each visit: unit look-up, loads, range test). Here, once per product, every
// (A-entry e, interior chunk boundary b) of such a row gets the position in B row acol[e] of the first column >= C_b = sorted_cols[b · kChunk]: a binary
// search inside the window piece that holds C_b. Chunk q of entry e is then exactly [split(e, q), split(e, q + 1)) — no over-visit, no range test, and
// the entry pass reads its bounds by entry index (no acol → wsplit dependence). need[i] = entries · (chunks − 1) of the i-th row of the launch's list (0 for
// rows of one chunk); ct_off = its exclusive scan, indexed by list position like the kernel's own walk.
// One thread per (row, entry, boundary) item, the items numbered through ct_off (a hub row alone holds millions of them: one workgroup per row took 2.6 ms
// per launch for work of a few hundred µs).
__global__ __launch_bounds__(256) void chunk_splits_kernel(long long total, int n, const int *__restrict__ rows, int K, int N, const int *__restrict__ wsplit,
                                                           const int *__restrict__ arpt, const int *__restrict__ acol, const int *__restrict__ brpt,
                                                           const int *__restrict__ bcol /* window ids */, const int *__restrict__ crpt,
                                                           const long long *__restrict__ pre_off, const int *__restrict__ pre_cols, const int *__restrict__ ccol /* window ids of the rows without carried columns */, int chunk,
                                                           const long long *__restrict__ ct_off /* n + 1 */, int *__restrict__ ct,
                                                           const int *__restrict__ choff = nullptr /* rank launch: the rows' chunk lists (rank_chunks_kernel) give the boundaries */, const RankChunk *__restrict__ chunks = nullptr)
{
    const long long i0 = (long long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    // the list position whose items hold the workgroup's FIRST item — a uniform search (scalar loads, once per 256 items: a search per item was a chain of
    // 17 dependent loads in front of every 10-step split search) — then a short walk forward for the lanes whose item belongs to a later row
    int rl = 0, rh = n;
    while (rl < rh) {
        const int mid = rl + ((rh - rl) >> 1);
        if (ct_off[mid + 1] > i0) rh = mid; else rl = mid + 1;
    }
    if (i >= total) return;
    while (ct_off[rl + 1] <= i) ++rl;
    const int row = rows[rl], nz = crpt[row + 1] - crpt[row];
    const long long po = pre_off ? pre_off[row] : -1, idx = i - ct_off[rl];
    const int a0 = arpt[row], nb = choff ? choff[rl + 1] - choff[rl] - 1 : (nz + chunk - 1) / chunk - 1;
    const int e = (int)(idx / nb), b = (int)(idx - (long long)e * nb) + 1;
    const int c = acol[a0 + e], cb = choff ? chunks[choff[rl] + b].cstart : po >= 0 ? pre_cols[po + (long long)chunk * b] : ccol[crpt[row] + (long long)chunk * b];
    const int sb = split_bits(N), W = (N + (1 << sb) - 1) >> sb, wf = cb >> sb;
    int lo = (wsplit && wf > 0) ? wsplit[(size_t)(wf - 1) * K + c] : brpt[c];
    int hi = (wsplit && wf < W - 1) ? wsplit[(size_t)wf * K + c] : brpt[c + 1];
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (bcol[mid] < cb) lo = mid + 1; else hi = mid;
    }
    ct[i] = lo;
}

// Unit lists (round 4). The walk of a value chunk used to start with an entry pass (every A-entry's bounds from HBM into LDS), a block scan of the
// entries' unit counts and a binary search per unit, three barriers and a dependent load chain per chunk, before the first product could be requested —
// with nothing to overlap it: the kernel's LDS allows one workgroup per CU. All of that depends only on the row structure and the chunk splits, both known
// before the kernel starts. So a pre-pass writes, per (row, chunk), the list of its UNITS — up to 64 consecutive entries of one B row that fall into the
// chunk, with the A-value they are multiplied by — and the kernel reads its units with scalar loads, a round ahead of everything else:
//   item (list position i, chunk q, A-entry e), ordered by (i, q, e): piece [split(e, q), split(e, q + 1)) of B row acol[e]  (split(e, 0) = row start,
//   split(e, chunks) = row end: every entry of a B row used by row i lands in exactly one of the row's chunks);
//   unit_count_kernel: units of the item = ceil(piece / 64); an exclusive scan gives the item's first unit; unit_expand_kernel writes the descriptors.
// The lists are built by TASKS: one thread per (row, A-entry) writes the unit counts of the entry's chunks from the splits (chunk_splits_kernel: one thread per
// (entry, boundary) — 10^7 short independent searches beat 10^6 chains of them), a second pass over the same tasks writes the descriptors. (The first form, one
// thread per item re-deriving its row and re-loading its bounds, cost 0.75 + 0.36 + 0.69 ms per product; it left the tree in round 5.)
__global__ void unit_rows_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const int *__restrict__ crpt, int chunk, int nz_lo, int nz_hi,
                                 long long *__restrict__ tasks, long long *__restrict__ items)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    long long na = 0, nch = 0;
    if (i < n) {
        const int row = rows[i], nz = crpt[row + 1] - crpt[row];
        if (nz > nz_lo && nz <= nz_hi) { na = arpt[row + 1] - arpt[row]; nch = (nz + chunk - 1) / chunk; }
    }
    tasks[i] = na;
    items[i] = na * nch;
}
__global__ void diff_ll_kernel(int n, const long long *__restrict__ a, const long long *__restrict__ b, long long *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}
template <bool EXPAND>
__global__ __launch_bounds__(256) void unit_task_kernel(long long bound /* threads launched: an upper bound of the task count */, int n, const int *__restrict__ rows,
                                                        const long long *__restrict__ task_off, const long long *__restrict__ item_off,
                                                        const int *__restrict__ arpt, const int *__restrict__ acol, const double *__restrict__ aval,
                                                        const int *__restrict__ brpt, const int *__restrict__ bcol /* window ids */, const int *__restrict__ crpt,
                                                        const long long *__restrict__ pre_off, const int *__restrict__ pre_cols, const int *__restrict__ ccol, int chunk,
                                                        int *__restrict__ ct /* splits: item_off[i] − task_off[i] + e·(chunks − 1) */, int *__restrict__ ucount,
                                                        const int *__restrict__ uoff, UnitDesc *__restrict__ U, int have_ct /* !EXPAND: the splits are in ct already (chunk_splits_kernel) */,
                                                        const int *__restrict__ choff = nullptr /* rank launch: chunks per row from the chunk lists */)
{
    const long long t0 = (long long)blockIdx.x * blockDim.x, total = task_off[n];
    long long t = t0 + threadIdx.x;
    if constexpr (!EXPAND) { if (t > bound) return; }               // (EXPAND: every lane of a wave stays — the descriptors are written with wave shuffles)
    if constexpr (!EXPAND) { if (t == 0) ucount[item_off[n]] = 0; }       // (the scan of the counts runs over items + 1 entries)
    if (t0 >= total) return;                                         // (uniform)
    const bool valid = t < total;                                   // EXPAND: the lanes past the last task stay for the wave's shuffles, on the last task's data, writing nothing
    if (!valid) { if constexpr (!EXPAND) return; t = total - 1; }
    int rl = 0, rh = n;                                             // the list position that holds the workgroup's first task (uniform), then a short walk forward
    while (rl < rh) {
        const int mid = rl + ((rh - rl) >> 1);
        if (task_off[mid + 1] > t0) rh = mid; else rl = mid + 1;
    }
    while (task_off[rl + 1] <= t) ++rl;
    const int row = rows[rl], e = (int)(t - task_off[rl]);
    const int a0 = arpt[row], na = arpt[row + 1] - a0, off = crpt[row], nz = crpt[row + 1] - off, nb = choff ? choff[rl + 1] - choff[rl] - 1 : (nz + chunk - 1) / chunk - 1;
    const int c = acol[a0 + e];
    const int end = brpt[c + 1];
    int prev = brpt[c];
    const long long io = item_off[rl];
    int *ctr = ct + (io - task_off[rl]) + (long long)e * nb;
    if constexpr (!EXPAND) {
        const long long po = pre_off ? pre_off[row] : -1;
        const int *cols = po >= 0 ? pre_cols + po : ccol + off;    // the row's sorted distinct columns (window ids)
        for (int b = 1; b <= nb; ++b) {
            int lo = prev, hi = end;
            if (have_ct) lo = ctr[b - 1];                           // (uniform) one thread per (entry, boundary) has searched already: 10^7 short independent searches beat 10^6 chains of them
            else {
                const int cb = cols[(long long)chunk * b];
                while (lo < hi) {
                    const int mid = lo + ((hi - lo) >> 1);
                    if (bcol[mid] < cb) lo = mid + 1; else hi = mid;
                }
                ctr[b - 1] = lo;
            }
            ucount[io + (long long)(b - 1) * na + e] = (lo - prev + 63) >> 6;
            prev = lo;
        }
        ucount[io + (long long)nb * na + e] = (end - prev + 63) >> 6;
    } else {
        // The descriptors are written by the WAVE, not by the task's lane (a lane with a 16 K-entry B row wrote 256 descriptors one after the other while its
        // neighbours wrote one or two — 0.60 ms for 640 MB): per chunk q the lanes' unit counts are scanned, and every lane writes one descriptor per step — unit u
        // of the wave belongs to the first lane whose inclusive count exceeds u (six shuffles), whose piece, destination and A-value come over by shuffle. The
        // pieces of consecutive entries of a row and chunk are consecutive in U (layout (row, chunk, entry)), so a step is one coalesced 1 KB store.
        const long long bits = __double_as_longlong(aval[a0 + e]);
        const int lo32 = (int)(bits & 0xFFFFFFFFll), hi32 = (int)(bits >> 32);
        const int lane = threadIdx.x & 63;
        int maxnb = valid ? nb : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxnb = max(maxnb, __shfl_xor(maxnb, o, 64));
        for (int q = 0; q <= maxnb; ++q) {                          // (uniform per wave)
            const bool act = valid && q <= nb;
            const int s0 = act ? (q == 0 ? prev : ctr[q - 1]) : 0, s1 = act ? (q < nb ? ctr[q] : end) : 0;
            const int cnt = (s1 - s0 + 63) >> 6;
            const int dst = act ? uoff[io + (long long)q * na + e] : 0;
            const unsigned incl = wave_inclusive_sum((unsigned)cnt);
            const int P = (int)incl - cnt, wave_units = (int)__builtin_amdgcn_readlane(incl, 63);
            for (int r = 0; r < wave_units; r += 64) {               // (uniform per wave)
                const int u = r + lane;
                int jl = 0, jh = 63;                                 // the first lane whose inclusive count exceeds u
#pragma unroll
                for (int step = 0; step < 6; ++step) {
                    const int mid = (jl + jh) >> 1;
                    const bool above = (unsigned)__shfl((int)incl, mid, 64) > (unsigned)u;
                    jh = above ? mid : jh;
                    jl = above ? jl : mid + 1;
                }
                const int j = min(jl, 63);
                const int Pj = __shfl(P, j, 64), s0j = __shfl(s0, j, 64), s1j = __shfl(s1, j, 64), dj = __shfl(dst, j, 64), lj = __shfl(lo32, j, 64), hj = __shfl(hi32, j, 64);
                if (u < wave_units) {
                    const int k = u - Pj, b = s0j + 64 * k;
                    U[dj + k] = UnitDesc{b, min(64, s1j - b), lj, hj};
                }
            }
        }
    }
}

template <int T, bool UNITS>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void spgemm_numeric_big_kernel(
    const int *__restrict__ rows, int nrows, int nz_lo, int nz_hi /* rows with nz outside (nz_lo, nz_hi] are left to the other shape */,
    int *__restrict__ next_row /* not NULL: rows are handed out one at a time through this counter (a list sorted longest first) */, int N, int K, const int *__restrict__ wsplit, const int *__restrict__ arpt, const int *__restrict__ acol, const double *__restrict__ aval,
    const int *__restrict__ brpt, const int *__restrict__ bcol /* window ids: compact when col_of is given */, const int *__restrict__ col_of,
    const double *__restrict__ bval, const long long *__restrict__ row_flop,
    const int *__restrict__ crpt, int *__restrict__ ccol, double *__restrict__ cval,
    const long long *__restrict__ pre_off, const int *__restrict__ pre_cols,
    const long long *__restrict__ item_off /* UNITS: first item of the list's i-th row (unit_task_kernel) */, const int *__restrict__ uoff /* first unit of an item */,
    const UnitDesc *__restrict__ U, const NumRowMeta *__restrict__ meta /* per list position (num_row_meta_kernel) */)
{
    // No static __shared__ in this kernel: it would sit in front of the dynamic region and push the fp64 table of phase 2 off its
    // 8-byte alignment (cdna_hip_programming.md Guideline 17). Everything is carved from the dynamic region instead.
    constexpr int kBigThreads = T, kBigWindowWords = BigCfg<T>::kWindowWords, kBigChunk = BigCfg<T>::kChunk;
    extern __shared__ int lds_i[];
    const BigSide<T> sd(lds_i + kBigWindowWords);
    const int t = threadIdx.x;
    constexpr int kPerThread = kBigChunk / kBigThreads;             // 8 slots per thread
    int cc[kPerThread], nridx = 0;
    auto fetch_from = [&](const int *src, int qn) {                 // a chunk's sorted columns: all of a thread's loads in flight together
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) cc[u] = src[min(t + u * kBigThreads, qn - 1)];
    };                                                             // (no uniform loads here: hipcc reads those into SGPRs and waits on the spot)
    // A row's metadata (its id, its ranges in A and C, where its carried columns start) sit behind two dependent loads: ticket → rows[] → arpt / crpt / pre_off.
    // They are uniform, so they are read with scalar loads (readfirstlane makes the ticket an SGPR), and the NEXT row's are requested as soon as its ticket is
    // known — before the value chunks of the current row — instead of at the top of its turn (110 K rows × two round trips were ≈ 10 % of the kernel).
    struct RowMeta { int row, a0, a1, off, nz; long long po, ioff; int u0, u1; };
    long long total_items = 0;
    if constexpr (UNITS) total_items = item_off[nrows];
    auto load_meta = [&](int idx) {
        const int ci = min(idx, nrows - 1);                        // (a ticket past the end reads the last row: never used)
        const NumRowMeta p = meta[ci];                             // uniform index: scalar loads, ONE round trip
        RowMeta m;
        m.row = p.row; m.a0 = p.a0; m.a1 = p.a1; m.off = p.off; m.nz = p.nz; m.po = p.po; m.ioff = p.ioff; m.u0 = p.u0; m.u1 = p.u1;
        return m;
    };
    int ridx = blockIdx.x;
    if (next_row) {                                                // uniform
        if (t == 0) sd.ctrl[28] = atomicAdd(next_row, 1);
        __syncthreads();
        ridx = __builtin_amdgcn_readfirstlane(sd.ctrl[28]);
        __syncthreads();
    }
    RowMeta cur = load_meta(ridx), nxt = cur;
    bool have_first = false;                                       // this row's first chunk of columns was requested during the previous row's last chunk
    for (; ridx < nrows; ridx = nridx, cur = nxt) {                // persistent: see spgemm_symbolic_window_kernel
    if (next_row) { if (t == 0) sd.ctrl[29] = atomicAdd(next_row, 1); }   // read below, behind a barrier
    else { nridx = ridx + gridDim.x; nxt = load_meta(nridx); }
    const int a0 = cur.a0, a1 = cur.a1, off = cur.off, nz = cur.nz;
    if (nz <= nz_lo || nz > nz_hi) {                               // uniform: the whole workgroup skips the row
        if (next_row) { __syncthreads(); nridx = __builtin_amdgcn_readfirstlane(sd.ctrl[29]); __syncthreads(); nxt = load_meta(nridx); }
        have_first = false;
        continue;
    }

    BIG_PROF_DECL;
    // The row's sorted distinct columns (window ids) are in place when this kernel starts: carried from the symbolic phase of the one-shot call in
    // pre_cols[po …] (po >= 0), or written into ccol at the row's own offset by the emit pass in front of this launch (po < 0; round 4 — the mark-and-emit
    // phase used to be part of this kernel and cost it registers on the path every product takes).
    const long long po = cur.po;                                   // uniform
#ifdef G4S_PROFILE_BIG
    if (po + a0 + a1 + off + nz == -12345) sd.ctrl[30] = 1;        // (forces the row's metadata to have arrived)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BIG_PROF(3);
#endif
    __syncthreads();                                               // the next-row ticket (ctrl[29]) is visible; the previous row's last chunk is out of LDS
    BIG_PROF(0);
    if (next_row) { nridx = __builtin_amdgcn_readfirstlane(sd.ctrl[29]); nxt = load_meta(nridx); }   // (nobody writes the slot again before this row's last barrier)
    // (round 4 tried the three dependent levels of load_meta behind three different barriers of the row's first chunk: 14.83 ms against 14.66 — the scalar
    // loads share their counter with LDS, and spreading them only spreads the waits)

    // ---- phase 2: values, one chunk of the sorted columns at a time
    // A product finds its slot through a bucket index over the chunk's column span: bucket b = (col - first) >> shift holds
    // (first slot << 16 | last slot) of the columns that fall in it, so the binary search runs over one bucket (0–2 steps where the
    // row is dense or evenly spread) instead of the whole chunk (13 steps).
    double *V = reinterpret_cast<double *>(lds_i);                 // kBigChunk doubles
    int *KC = lds_i + 2 * kBigChunk;                                // kBigChunk ints
    unsigned *IDX = reinterpret_cast<unsigned *>(lds_i + 3 * kBigChunk);   // kBigChunk buckets: 4·kBigChunk ints = the bitmap's 128 KiB
    unsigned short *IDX16 = reinterpret_cast<unsigned short *>(IDX);       // (low half: last slot, high half: first slot)
    static_assert(4 * kBigChunk <= kBigWindowWords && kBigChunk <= 65536, "phase 2 reuses the bitmap region; slots are packed in 16 bits");
    constexpr int kU = kFlatUnitsPerRound;
    int kfirst = 0, klast = 0, shift = 0;
    constexpr bool exact = UNITS;                                  // unit lists carry exact chunk splits — every product visited lies in the chunk; the entry-pass walk runs on the window pieces
    // One round of a lane's products: their slots are found in lock-step (the dependent LDS reads of the kU searches interleave) for as many
    // halvings as the wave's deepest bucket needs — a unit is 64 consecutive entries of a sorted B row, so a wave's lanes sit in neighbouring
    // buckets — and the atomics come last. A round whose units all lie outside the chunk's column range is skipped by the whole wave.
    auto accumulate = [&](const int (&col)[kU], const double (&bv)[kU], const double (&av)[kU], const bool (&ok)[kU]) {
        int lo[kU], hi[kU], key[kU];
        bool in[kU], any_in = false;
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            in[q] = ok[q] && (exact || (col[q] >= kfirst && col[q] <= klast));
            any_in |= in[q];
        }
        if (G4S_KO & 16) {
#pragma unroll
            for (int q = 0; q < kU; ++q) { asm volatile("" :: "v"(col[q])); asm volatile("" :: "v"(bv[q])); }
            return;
        }
        if (!__any(any_in)) return;
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            key[q] = in[q] ? col[q] : kfirst;
            const unsigned w = IDX[(key[q] - kfirst) >> shift];    // the column is present, so its bucket is not empty
            lo[q] = (int)(w >> 16); hi[q] = (int)(w & 0xffffu);
        }
#ifdef G4S_PROFILE_BIG
        prof_acc[11] += 1;                                          // rounds (of the reporting wavefront) and halving steps: how deep the buckets are
#endif
        for (; !(G4S_KO & 1);) {
            bool more = false;
#pragma unroll
            for (int q = 0; q < kU; ++q) more |= lo[q] < hi[q];
            if (!__any(more)) break;
#ifdef G4S_PROFILE_BIG
            prof_acc[12] += 1;
#endif
            int mid[kU], km[kU];
#pragma unroll
            for (int q = 0; q < kU; ++q) { mid[q] = (lo[q] + hi[q]) >> 1; km[q] = KC[mid[q]]; }   // the kU reads in flight together
#pragma unroll
            for (int q = 0; q < kU; ++q) {
                const bool less = km[q] < key[q];
                lo[q] = less ? mid[q] + 1 : lo[q];
                hi[q] = less ? hi[q] : mid[q];
            }
        }
#pragma unroll
        for (int q = 0; q < kU; ++q)
            if (in[q] && !(G4S_KO & 2)) atomicAdd(&V[lo[q] & (kBigChunk - 1)], av[q] * bv[q]);   // (the mask: a stale bucket can only be read with a wrong crpt from the caller — stay inside the chunk)
    };
    // The chunk's sorted columns are issued a chunk ahead (they are consumed at the top of the next chunk, a whole accumulation pass later).
    auto fetch_chunk = [&](int q0) { fetch_from(po >= 0 ? pre_cols + po + q0 : ccol + off + q0, min(kBigChunk, nz - q0)); };
    auto open_chunk = [&](int qn) {                                // cc → LDS (the bucket index is not cleared: only buckets that hold a column of this chunk are ever looked up, and those are rewritten)
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = t + u * kBigThreads;
            if (i < qn) { KC[i] = cc[u]; V[i] = 0.0; }
        }
    };
    auto chunk_span = [&](int qn) {                                // after the barrier behind open_chunk
        kfirst = KC[0]; klast = KC[qn - 1];
        const int span = klast - kfirst;                           // < 2^31
        shift = span < kBigChunk ? 0 : 32 - __clz(span) - BigCfg<T>::kChunkBits;   // (span >> shift) < kBigChunk
        static_assert(kBigChunk == (1 << BigCfg<T>::kChunkBits), "bucket shift");
    };
    auto build_index = [&](int qn) {                               // bucket b = (col − first) >> shift → (first slot << 16 | last slot)
        int b[kPerThread], bp[kPerThread], bn[kPerThread];         // (all LDS reads first: read by read, each waits on its own)
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = min(t + u * kBigThreads, qn - 1);
            b[u] = KC[i]; bp[u] = KC[max(i - 1, 0)]; bn[u] = KC[min(i + 1, qn - 1)];
        }
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int i = t + u * kBigThreads;
            if (i < qn) {
                const int bb = (b[u] - kfirst) >> shift;               // the columns are sorted: a bucket's first and last slot sit where the bucket id changes — two 16-bit stores, no atomics
                if (i == 0 || ((bp[u] - kfirst) >> shift) != bb) IDX16[2 * bb + 1] = (unsigned short)i;
                if (i == qn - 1 || ((bn[u] - kfirst) >> shift) != bb) IDX16[2 * bb] = (unsigned short)i;
            }
        }
    };
    if (!have_first) {                                             // uniform: the first row of this workgroup, or a skipped row in between
        fetch_chunk(0);
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) asm volatile("" :: "v"(cc[u]));   // settled on this path too (see the store step): the first open needs them at once anyway
    }
    have_first = true;
    // UNITS: the chunk's products come from its unit list (unit_kernel): wave w takes units w·kU … w·kU + kU − 1 of every round, the descriptors read with scalar
    // loads. Round 0 is requested before the chunk opens — its B-row loads fly under the LDS set-up, the two barriers and the bucket index.
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int cu0 = cur.u0, cu1 = cur.u1, cu2 = 0, nu = 0;               // this chunk's units [cu0, cu1); cu2: end of the next chunk's
    const int na = a1 - a0;
    int rc[kU];
    double rv[kU], rav[kU];
    bool rok[kU];
    auto request_round = [&](int g) {
        // the round's kU descriptors first (uniform: one s_load_dwordx4 each, all in flight), then its vector loads — descriptor by descriptor the compiler waited
        // for each scalar load before it issued that unit's two vector loads: kU scalar round trips in a row in front of every round (ISA, round 4)
        UnitDesc d[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) d[q] = U[cu0 + min(g + q, nu - 1)];
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            rok[q] = g + q < nu && lane < d[q].len;
            const int kk = d[q].bpos + min(lane, d[q].len - 1);
            rc[q] = bcol[kk];
            rv[q] = bval[kk];
            rav[q] = __longlong_as_double(((long long)d[q].av_hi << 32) | (unsigned)d[q].av_lo);
        }
    };
    for (int q0 = 0; q0 < nz; q0 += kBigChunk) {
        const int qn = min(kBigChunk, nz - q0);
        const int qi = q0 / kBigChunk;
        if constexpr (UNITS) {
            nu = cu1 - cu0;
            request_round(wave * kU);
            const int nch = (nz + kBigChunk - 1) / kBigChunk;
            cu2 = uoff[min(cur.ioff + (long long)min(qi + 2, nch) * na, total_items)];   // (consumed a chunk later)
        }
        open_chunk(qn);
#ifdef G4S_PROFILE_BIG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q0 == 0) BIG_PROF(9); else BIG_PROF(1);               // the chunk's columns have arrived and are in LDS: first chunk of a row / later chunks
#endif
        auto fetch_next = [&]() {
            // the next chunk's columns — in the row's last chunk: the first chunk of the workgroup's NEXT row. One unconditional fetch from a selected pointer: a
            // load under a branch makes hipcc wait at the join for every load that might be in flight.
            const bool last = q0 + kBigChunk >= nz;                 // uniform
            const int *src = last ? (nxt.po >= 0 ? pre_cols + nxt.po : ccol + nxt.off)
                                  : (po >= 0 ? pre_cols + po + q0 + kBigChunk : ccol + off + q0 + kBigChunk);
            fetch_from(src, last ? max(1, min(kBigChunk, nxt.nz)) : min(kBigChunk, nz - q0 - kBigChunk));
        };
        fetch_next();
        __syncthreads();
        chunk_span(qn);
        BIG_PROF(6);
        if (!(G4S_KO & 4)) build_index(qn);
        __syncthreads();
        BIG_PROF(7);
        const int *clo = brpt, *chi = brpt + 1;
        if constexpr (!UNITS) window_bounds(brpt, wsplit, K, N, kfirst, klast, clo, chi);   // the part of each B row inside the windows this chunk spans
        // the chunk's columns as B's column ids (window ids → ids): a gather, started in front of the walk's last barrier so that it arrives under the barrier's skew
        int orig[kPerThread];
        auto gather_ids = [&]() {
            if (col_of) {
#pragma unroll
                for (int u = 0; u < kPerThread; ++u) orig[u] = col_of[KC[min(t + u * kBigThreads, qn - 1)]];
            } else {
#pragma unroll
                for (int u = 0; u < kPerThread; ++u) orig[u] = KC[min(t + u * kBigThreads, qn - 1)];
            }
        };
        if constexpr (UNITS) {
            accumulate(rc, rv, rav, rok);                          // round 0: requested at the top of the chunk
            for (int g = (wave + T / 64) * kU; g < nu; g += (T / 64) * kU) {   // (uniform per wave; most chunks hold at most one round per wave)
                request_round(g);
                accumulate(rc, rv, rav, rok);
            }
            gather_ids();
            __syncthreads();
            cu0 = cu1; cu1 = cu2;
        } else {
            flat_products<T, true, kU>(a0, a1, acol, aval, clo, chi, bcol, bval, sd, t, accumulate, gather_ids);
        }
        BIG_PROF(8);
        {   // the chunk's values and column ids (gathered above).
            // hipcc's wait-count book: a register whose load is consumed only under a per-lane branch (the stores below) stays "pending" on the path that skips the
            // branch, and the merged state then makes the NEXT chunk wait — at its very top, for the stores just issued — before it may reuse that register. A use
            // of the prefetched columns and of the gathered ids here, in front of the stores, settles them on every path: the next chunk opens without a wait.
#pragma unroll
            for (int u = 0; u < kPerThread; ++u) { asm volatile("" :: "v"(cc[u])); asm volatile("" :: "v"(orig[u])); }
            double val[kPerThread];
#pragma unroll
            for (int u = 0; u < kPerThread; ++u) val[u] = V[min(t + u * kBigThreads, qn - 1)];
#pragma unroll
            for (int u = 0; u < kPerThread; ++u) {
                const int i = t + u * kBigThreads;
                if (i < qn && !(G4S_KO & 8)) cval[off + q0 + i] = val[u];
            }
            if (col_of || po >= 0) {                               // (a numeric-only call without a column map has them in place already)
#pragma unroll
                for (int u = 0; u < kPerThread; ++u) {
                    const int i = t + u * kBigThreads;
                    if (i < qn && !(G4S_KO & 8)) ccol[off + q0 + i] = orig[u];
                }
            }
        }
        __syncthreads();
        BIG_PROF(10);
    }
    BIG_PROF_FLUSH;
    __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ bitmap-rank path (hub rows)
struct HubItem { int slot, row, j0, j1; };   // a workgroup's share of one hub row: A-entries [j0, j1)
constexpr int kHubJChunk = 32;

__global__ __launch_bounds__(256) void hub_mark_kernel(const HubItem *__restrict__ items, int W,
                                                        const int *__restrict__ acol, const int *__restrict__ brpt,
                                                        const int *__restrict__ bcol, unsigned *__restrict__ bitmap)
{
    __shared__ int4 longs[kLongCap + 1];                          // the only shared array of this kernel: 16-byte aligned
    const HubItem it = items[blockIdx.x];
    unsigned *bm = bitmap + (size_t)it.slot * W;
    const int t = threadIdx.x;
    if (t == 0) longs[0].x = 0;
    __syncthreads();
    for (int j = it.j0 + (t >> 5); j < it.j1; j += 8) {           // 32 lanes per A-entry; long B rows are deferred to the whole workgroup
        const int c = acol[j];
        const int b0 = brpt[c], b1 = brpt[c + 1];
        if (defer_long_b(longs, b0, b1, 0.0, t & 31, 31, 256)) continue;
        for (int k = b0 + (t & 31); k < b1; k += 32) {
            const int col = bcol[k];
            atomicOr(&bm[col >> 5], 1u << (col & 31));
        }
    }
    __syncthreads();
    const int nl = min(longs[0].x, kLongCap);
    for (int i = 0; i < nl; ++i) {
        const int4 e = longs[1 + i];
        for (int k = e.x + t; k < e.y; k += 256) {
            const int col = bcol[k];
            atomicOr(&bm[col >> 5], 1u << (col & 31));
        }
    }
}

// One workgroup per hub row: exclusive popcount prefix over the row's bitmap words; total = nz of the row.
__global__ __launch_bounds__(256) void hub_prefix_kernel(const int *__restrict__ slot_rows, int W, const unsigned *__restrict__ bitmap,
                                                          int *__restrict__ prefix, int *__restrict__ row_nz)
{
    __shared__ int s_part[256];
    __shared__ int s_carry;
    const int slot = blockIdx.x, t = threadIdx.x;
    const unsigned *bm = bitmap + (size_t)slot * W;
    int *pf = prefix + (size_t)slot * W;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < W; base += 256 * 8) {
        int loc[8], sum = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = base + t * 8 + u;
            loc[u] = w < W ? __popc(bm[w]) : 0;
            sum += loc[u];
        }
        s_part[t] = sum;
        __syncthreads();
        // exclusive scan of s_part by Hillis–Steele (256 entries)
        for (int off = 1; off < 256; off <<= 1) {
            const int v = t >= off ? s_part[t - off] : 0;
            __syncthreads();
            s_part[t] += v;
            __syncthreads();
        }
        int run = s_carry + s_part[t] - sum;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = base + t * 8 + u;
            if (w < W) pf[w] = run;
            run += loc[u];
        }
        __syncthreads();
        if (t == 255) s_carry += s_part[255];
        __syncthreads();
    }
    if (t == 0 && row_nz) row_nz[slot_rows[slot]] = s_carry;
}

// Column ids of a hub row in ascending order, values zeroed (products are accumulated afterwards).
__global__ __launch_bounds__(256) void hub_emit_kernel(const int *__restrict__ slot_rows, int W, const unsigned *__restrict__ bitmap,
                                                        const int *__restrict__ prefix, const int *__restrict__ crpt,
                                                        int *__restrict__ ccol, double *__restrict__ cval)
{
    const int slot = blockIdx.y;
    const int off = crpt[slot_rows[slot]];
    const unsigned *bm = bitmap + (size_t)slot * W;
    const int *pf = prefix + (size_t)slot * W;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < W; w += gridDim.x * blockDim.x) {
        unsigned bits = bm[w];
        int pos = off + pf[w];
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            ccol[pos] = (w << 5) + b;
            cval[pos] = 0.0;
            ++pos;
        }
    }
}

__global__ __launch_bounds__(256) void hub_accumulate_kernel(const HubItem *__restrict__ items, int W,
                                                              const int *__restrict__ acol, const double *__restrict__ aval,
                                                              const int *__restrict__ brpt, const int *__restrict__ bcol,
                                                              const double *__restrict__ bval, const unsigned *__restrict__ bitmap,
                                                              const int *__restrict__ prefix, const int *__restrict__ crpt,
                                                              double *__restrict__ cval)
{
    __shared__ int4 longs[kLongCap + 1];
    const HubItem it = items[blockIdx.x];
    const unsigned *bm = bitmap + (size_t)it.slot * W;
    const int *pf = prefix + (size_t)it.slot * W;
    double *out = cval + crpt[it.row];
    const int t = threadIdx.x;
    if (t == 0) longs[0].x = 0;
    __syncthreads();
    auto add = [&](int k, double av) {
        const int col = bcol[k];
        const int w = col >> 5;
        const int pos = pf[w] + __popc(bm[w] & ((1u << (col & 31)) - 1u));
        atomicAdd(&out[pos], av * bval[k]);
    };
    for (int j = it.j0 + (t >> 5); j < it.j1; j += 8) {
        const int c = acol[j];
        const double av = aval[j];
        const int b0 = brpt[c], b1 = brpt[c + 1];
        if (defer_long_b(longs, b0, b1, av, t & 31, 31, 256)) continue;
        for (int k = b0 + (t & 31); k < b1; k += 32) add(k, av);
    }
    __syncthreads();
    const int nl = min(longs[0].x, kLongCap);
    for (int i = 0; i < nl; ++i) {
        const int4 e = longs[1 + i];
        const double av = long_b_value(e);
        for (int k = e.x + t; k < e.y; k += 256) add(k, av);
    }
}

// ------------------------------------------------------------------------------------------------ exclusive scan → crpt
constexpr int kScanChunk = 2048; // elements per workgroup

__global__ __launch_bounds__(256) void scan_block_sums_kernel(int M, const int *__restrict__ row_nz, long long *__restrict__ block_sums)
{
    __shared__ long long s[4];
    long long v = 0;
    const int base = blockIdx.x * kScanChunk;
    for (int i = threadIdx.x; i < kScanChunk; i += 256) {
        const int r = base + i;
        if (r < M) v += row_nz[r];
    }
    v = wave_sum_ll(v);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void scan_write_kernel(int M, const int *__restrict__ row_nz, const long long *__restrict__ block_offs,
                                                          int *__restrict__ crpt)
{
    // out[0] = 0, out[i+1] = out[i] + in[i]  (seq_scan, mm/inc/utility.h:156-163)
    __shared__ long long s_scan[256];
    const int base = blockIdx.x * kScanChunk;
    constexpr int PER = kScanChunk / 256;
    int loc[PER];
    long long sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int r = base + threadIdx.x * PER + u;
        loc[u] = r < M ? row_nz[r] : 0;
        sum += loc[u];
    }
    s_scan[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const long long v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    long long run = block_offs[blockIdx.x] + s_scan[threadIdx.x] - sum;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int r = base + threadIdx.x * PER + u;
        if (r < M) crpt[r] = (int)run;
        run += loc[u];
        if (r == M - 1) crpt[M] = (int)run;
    }
}

// ------------------------------------------------------------------------------------------------ host orchestration
struct RowClasses {
    DevBuf cls, lists, hist;
    int count[CLS_COUNT] = {0};
    int offset[CLS_COUNT + 1] = {0};
    const int *list(int c) const { return lists.as<int>() + offset[c]; }
};

int classify_rows(int M, const long long *d_size, const ClassLimits &lim, int cols_clip, RowClasses &rc, hipStream_t s)
{
    G4S_TRY(rc.cls.alloc(sizeof(int) * (size_t)M));
    G4S_TRY(rc.lists.alloc(sizeof(int) * (size_t)M));
    G4S_TRY(rc.hist.alloc(sizeof(int) * 2 * CLS_COUNT));
    G4S_HIP_TRY(hipMemsetAsync(rc.hist.p, 0, sizeof(int) * 2 * CLS_COUNT, s));
    const int grid = std::min(kClassBlocks, (M + 255) / 256);
    hipLaunchKernelGGL(classify_kernel, dim3(grid), dim3(256), 0, s, M, d_size, lim, cols_clip, rc.cls.as<int>(), rc.hist.as<int>());
    G4S_HIP_TRY(g4s::read_small(rc.count, rc.hist.p, sizeof(int) * CLS_COUNT, s));
    int *cursor = rc.hist.as<int>() + CLS_COUNT;                   // (zeroed with the counts)
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, s, M, rc.cls.as<int>(), (const int *)rc.hist.as<int>(), cursor, rc.lists.as<int>());
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(g4s::reads_sync(s));                               // (the host's wait for the counts covers the scatter: it no longer needs anything from the host)
    rc.offset[0] = 0;
    for (int c = 0; c < CLS_COUNT; ++c) rc.offset[c + 1] = rc.offset[c] + rc.count[c];
    return G4S_OK;
}

template <typename Kernel>
int allow_lds(Kernel k, size_t bytes)
{
    if (bytes > 64 * 1024) G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return G4S_OK;
}

// Hub rows: mark → prefix (→ counts) [→ emit → accumulate], in batches bounded by the bitmap workspace.
int run_hub_rows(bool numeric, const std::vector<int> &hub_rows, const std::vector<int> &h_arpt_of_rows /*2 per row*/, int N,
                 const int *arpt, const int *acol, const double *aval, const int *brpt, const int *bcol, const double *bval,
                 int *row_nz, const int *crpt, int *ccol, double *cval, hipStream_t s)
{
    (void)arpt;
    if (hub_rows.empty()) return G4S_OK;
    const int W = (N + 31) / 32;
    const size_t per_row = (size_t)W * 8; // bitmap + prefix words
    const size_t budget = (size_t)1 << 30;
    const int max_slots = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hub_rows.size(), 32768), budget / per_row));
    DevBuf bitmap, prefix, slot_rows, items;
    G4S_TRY(bitmap.alloc((size_t)max_slots * W * 4));
    G4S_TRY(prefix.alloc((size_t)max_slots * W * 4));
    G4S_TRY(slot_rows.alloc(sizeof(int) * (size_t)max_slots));
    for (size_t b0 = 0; b0 < hub_rows.size(); b0 += max_slots) {
        const int nslots = (int)std::min<size_t>(max_slots, hub_rows.size() - b0);
        std::vector<HubItem> h_items;
        for (int sidx = 0; sidx < nslots; ++sidx) {
            const int a0 = h_arpt_of_rows[2 * (b0 + sidx)], a1 = h_arpt_of_rows[2 * (b0 + sidx) + 1];
            for (int j = a0; j < a1; j += kHubJChunk) h_items.push_back(HubItem{sidx, hub_rows[b0 + sidx], j, std::min(a1, j + kHubJChunk)});
        }
        G4S_TRY(items.alloc(sizeof(HubItem) * h_items.size()));
        G4S_HIP_TRY(hipMemcpyAsync(items.p, h_items.data(), sizeof(HubItem) * h_items.size(), hipMemcpyHostToDevice, s));
        G4S_HIP_TRY(hipMemcpyAsync(slot_rows.p, hub_rows.data() + b0, sizeof(int) * nslots, hipMemcpyHostToDevice, s));
        G4S_HIP_TRY(hipMemsetAsync(bitmap.p, 0, (size_t)nslots * W * 4, s));
        if (!h_items.empty())
            hipLaunchKernelGGL(hub_mark_kernel, dim3((unsigned)h_items.size()), dim3(256), 0, s, items.as<HubItem>(), W, acol, brpt, bcol, bitmap.as<unsigned>());
        hipLaunchKernelGGL(hub_prefix_kernel, dim3(nslots), dim3(256), 0, s, slot_rows.as<int>(), W, bitmap.as<unsigned>(), prefix.as<int>(),
                           numeric ? nullptr : row_nz);
        if (numeric) {
            const int gx = std::min(64, (W + 255) / 256);
            hipLaunchKernelGGL(hub_emit_kernel, dim3(gx, nslots), dim3(256), 0, s, slot_rows.as<int>(), W, bitmap.as<unsigned>(), prefix.as<int>(),
                               crpt, ccol, cval);
            if (!h_items.empty())
                hipLaunchKernelGGL(hub_accumulate_kernel, dim3((unsigned)h_items.size()), dim3(256), 0, s, items.as<HubItem>(), W, acol, aval, brpt,
                                   bcol, bval, bitmap.as<unsigned>(), prefix.as<int>(), crpt, cval);
        }
        G4S_HIP_TRY(hipGetLastError());
        G4S_HIP_TRY(g4s::reads_sync(s)); // h_items is reused; workspace is reused by the next batch
    }
    return G4S_OK;
}

int fetch_rows_and_ranges(const int *d_list, int n, const int *d_arpt, std::vector<int> &rows, std::vector<int> &ranges, hipStream_t s)
{
    rows.resize(n);
    ranges.resize(2 * (size_t)n);
    if (!n) return G4S_OK;
    G4S_HIP_TRY(hipMemcpyAsync(rows.data(), d_list, sizeof(int) * n, hipMemcpyDeviceToHost, s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    std::sort(rows.begin(), rows.end());
    DevBuf d_rows, d_rng;
    G4S_TRY(d_rows.alloc(sizeof(int) * (size_t)n));
    G4S_TRY(d_rng.alloc(sizeof(int) * 2 * (size_t)n));
    G4S_HIP_TRY(hipMemcpyAsync(d_rows.p, rows.data(), sizeof(int) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gather_ranges_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_rows.as<int>(), n, d_arpt, d_rng.as<int>());
    G4S_HIP_TRY(hipMemcpyAsync(ranges.data(), d_rng.p, sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost, s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    return G4S_OK;
}

int compute_row_flop(int M, const int *arpt, const int *acol, const int *brpt, long long *d_row_flop, int64_t *total, hipStream_t s, long long annz = -1)
{
    if (M > 0 && annz > 0 && annz < (1ll << 31)) {                  // entry-parallel form (the caller knows nnz(A))
        DevBuf f, P;
        G4S_TRY(f.alloc(sizeof(long long) * ((size_t)annz + 1)));
        G4S_TRY(P.alloc(sizeof(long long) * ((size_t)annz + 1)));
        hipLaunchKernelGGL(entry_flop_kernel, dim3((unsigned)((annz + 256) / 256)), dim3(256), 0, s, annz, acol, brpt, f.as<long long>());
        G4S_TRY(g4s::prims::exclusive_scan(f.as<long long>(), P.as<long long>(), annz + 1, s));
        hipLaunchKernelGGL(row_flop_from_scan_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, arpt, P.as<long long>(), d_row_flop);
        G4S_HIP_TRY(hipGetLastError());
        long long h = 0;
        G4S_HIP_TRY(g4s::read_small(&h, P.as<long long>() + annz, sizeof(h), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        if (total) *total = (int64_t)h;
        return G4S_OK;
    }
    DevBuf tot;
    G4S_TRY(tot.alloc(sizeof(unsigned long long)));
    G4S_HIP_TRY(hipMemsetAsync(tot.p, 0, sizeof(unsigned long long), s));
    if (M > 0) hipLaunchKernelGGL(row_flop_kernel, dim3((M + 31) / 32), dim3(256), 0, s, M, arpt, acol, brpt, d_row_flop, tot.as<unsigned long long>());
    G4S_HIP_TRY(hipGetLastError());
    unsigned long long h = 0;
    G4S_HIP_TRY(g4s::read_small(&h, tot.p, sizeof(h), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (total) *total = (int64_t)h;
    return G4S_OK;
}

int check_ids(const int *ids, long long n, int bound, const char *what, hipStream_t s)
{
    if (n <= 0) return G4S_OK;
    DevBuf flag;
    G4S_TRY(flag.alloc(sizeof(int)));
    G4S_HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), s));
    const int grid = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(check_range_kernel, dim3(grid), dim3(256), 0, s, ids, n, bound, flag.as<int>());
    int h = 0;
    G4S_HIP_TRY(g4s::read_small(&h, flag.p, sizeof(int), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (h) return g4s::set_error(G4S_ERR_INVALID, "SpGEMM: %s outside its valid range [0,%d)", what, bound);
    return G4S_OK;
}

// range check of B's column ids and the sorted-rows contract, one synchronisation
int check_b(const int *brpt, const int *bcol, int K, long long bnnz, int N, hipStream_t s, bool *unsorted)
{
    *unsorted = false;
    if (bnnz <= 0) return G4S_OK;
    DevBuf buf;
    G4S_TRY(buf.alloc(sizeof(unsigned long long) * 3));
    G4S_HIP_TRY(hipMemsetAsync(buf.p, 0, sizeof(unsigned long long) * 3, s));
    unsigned long long *d = buf.as<unsigned long long>();
    const int grid = (int)std::min<long long>((bnnz + 255) / 256, 512);   // (one atomic per workgroup on one counter: 4 096 of them were 20 of the pass's 28 µs on a 5.6e5-entry B)
    hipLaunchKernelGGL(check_descents_kernel, dim3(grid), dim3(256), 0, s, bcol, bnnz, N, reinterpret_cast<int *>(d), d + 1);
    hipLaunchKernelGGL(row_start_descents_kernel, dim3(std::min((K + 255) / 256, 512)), dim3(256), 0, s, K, brpt, bcol, d + 2);
    unsigned long long h[3] = {0, 0, 0};
    G4S_HIP_TRY(g4s::read_small(h, d, sizeof(h), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (h[0] & 0xffffffffull) return g4s::set_error(G4S_ERR_INVALID, "SpGEMM: a column id of B outside its valid range [0,%d)", N);
    *unsorted = h[1] != h[2];                                       // the caller sorts a private copy of B's rows (sort_b_rows)
    return G4S_OK;
}

// ---- B with unsorted rows (round 5). The reference takes any row order — the hash traversal never looks at it (mm/inc/hash_mult.h:579-600), HashSpGEMM<…, false>
// emits unsorted rows itself (:530-551) and a chained product feeds them back in as B; mkl_sparse_spmm takes them too (mm/inc/mkl_mult.h:58). The merge and window
// kernels here cut B's rows at column boundaries, so when the sortedness check fires the call sorts a PRIVATE copy of B's rows and works on that: two stable
// LSD radix sorts of the entry indices (prims.hpp), by column and then by row = by (row, column); the caller's arrays are not touched. Only then: the check
// itself rides on the opening pass of every call.
__global__ void sortb_col_keys_kernel(int n, const int *__restrict__ col, int N, int *__restrict__ key, int *__restrict__ idx)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { key[k] = N - 1 - col[k]; idx[k] = k; }            // (the sort is descending: ascending columns)
}
__global__ void sortb_row_keys_kernel(int n, int K, const int *__restrict__ rpt, const int *__restrict__ perm, int *__restrict__ key)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int e = perm[k];
    int lo = 0, hi = K;                                            // the row that holds entry e: the last r with rpt[r] <= e
    while (hi - lo > 1) {
        const int mid = lo + ((hi - lo) >> 1);
        if (rpt[mid] <= e) lo = mid; else hi = mid;
    }
    key[k] = K - 1 - lo;
}
__global__ void sortb_gather_kernel(int n, const int *__restrict__ perm, const int *__restrict__ col, const double *__restrict__ val, int *__restrict__ col_out, double *__restrict__ val_out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int e = perm[k];
    if (col_out) col_out[k] = col[e];
    if (val_out) val_out[k] = val[e];
}
inline int bits_for(int n) { return n > 1 ? 32 - __builtin_clz((unsigned)(n - 1)) : 1; }
// perm[k] = the entry of B that stands at position k once every row is sorted by column (ties keep their order)
int sort_b_rows(int K, int N, int bnnz, const int *brpt, const int *bcol, int *perm, hipStream_t s)
{
    DevBuf key, idx, key2, perm1, tk, tv;
    const size_t b = sizeof(int) * (size_t)bnnz;
    G4S_TRY(key.alloc(b)); G4S_TRY(idx.alloc(b)); G4S_TRY(key2.alloc(b)); G4S_TRY(perm1.alloc(b)); G4S_TRY(tk.alloc(b)); G4S_TRY(tv.alloc(b));
    const dim3 grid((unsigned)((bnnz + 255) / 256));
    hipLaunchKernelGGL(sortb_col_keys_kernel, grid, dim3(256), 0, s, bnnz, bcol, N, key.as<int>(), idx.as<int>());
    G4S_TRY(g4s::prims::sort_pairs_descending(key.as<int>(), idx.as<int>(), key2.as<int>(), perm1.as<int>(), tk.as<int>(), tv.as<int>(), bnnz, bits_for(N), s));
    hipLaunchKernelGGL(sortb_row_keys_kernel, grid, dim3(256), 0, s, bnnz, K, brpt, perm1.as<int>(), key.as<int>());
    G4S_TRY(g4s::prims::sort_pairs_descending(key.as<int>(), perm1.as<int>(), key2.as<int>(), perm, tk.as<int>(), tv.as<int>(), bnnz, bits_for(K), s));
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}
int gather_b(int bnnz, const int *perm, const int *bcol, const double *bval, int *col_out, double *val_out, hipStream_t s)
{
    if (bnnz > 0) hipLaunchKernelGGL(sortb_gather_kernel, dim3((unsigned)((bnnz + 255) / 256)), dim3(256), 0, s, bnnz, perm, bcol, bval, col_out, val_out);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

// ---- the key of a carried symbolic state: a position-dependent 64-bit sum over the five index arrays of the product. A numeric call takes the state
// over only when its arrays still hash to the value the symbolic call saw — the pointers alone match a caller that refilled the same buffers with another pattern.
__global__ __launch_bounds__(256) void pattern_hash_kernel(long long n0, const int *__restrict__ a0, long long n1, const int *__restrict__ a1, long long n2, const int *__restrict__ a2,
                                                           long long n3, const int *__restrict__ a3, long long n4, const int *__restrict__ a4, unsigned long long *__restrict__ out)
{
    const long long total = n0 + n1 + n2 + n3 + n4;
    unsigned long long sum = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long j = i;
        const int *p = a0;
        if (j >= n0) { j -= n0; p = a1; if (j >= n1) { j -= n1; p = a2; if (j >= n2) { j -= n2; p = a3; if (j >= n3) { j -= n3; p = a4; } } } }
        unsigned long long v = (unsigned long long)(unsigned)p[j] ^ ((unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull);
        v *= 0xff51afd7ed558ccdull; v ^= v >> 33;
        sum += v;
    }
    sum = (unsigned long long)wave_sum_ll((long long)sum);
    __shared__ unsigned long long s_p[4];
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_p[0] + s_p[1] + s_p[2] + s_p[3]);
}
// enqueues the hash of (arpt, acol, brpt, bcol, crpt) into *d_out (zeroed here); the caller reads it behind its next synchronisation
int enqueue_pattern_hash(int M, int K, long long annz, long long bnnz, const int *arpt, const int *acol, const int *brpt, const int *bcol, const int *crpt, unsigned long long *d_out, hipStream_t s)
{
    G4S_HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), s));
    const long long total = 2ll * (M + 1) + (K + 1) + annz + bnnz;
    const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(pattern_hash_kernel, dim3(grid), dim3(256), 0, s, (long long)M + 1, arpt, annz, acol, (long long)K + 1, brpt, bnnz, bcol, (long long)M + 1, crpt, d_out);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

// The opening of a one-shot or symbolic call in ONE host wait (it was three: A's ids, B's ids and order, the flop total): B's checks, the per-entry flop with
// the check of A's ids fused in, its scan, the per-row flop; flags and total come back together.
int checked_row_flop(int M, int K, int N, const int *arpt, const int *acol, long long annz, const int *brpt, const int *bcol, long long bnnz,
                     long long *d_row_flop, int64_t *total, hipStream_t s, bool *b_unsorted)
{
    *b_unsorted = false;
    if (!(M > 0 && annz > 0 && annz < (1ll << 31))) {              // the row-parallel form keeps its own sequence
        G4S_TRY(check_ids(acol, annz, K, "a column id of A", s));
        G4S_TRY(check_b(brpt, bcol, K, bnnz, N, s, b_unsorted));
        return compute_row_flop(M, arpt, acol, brpt, d_row_flop, total, s, annz);
    }
    DevBuf buf, f, P;
    G4S_TRY(buf.alloc(sizeof(unsigned long long) * 4));
    G4S_HIP_TRY(hipMemsetAsync(buf.p, 0, sizeof(unsigned long long) * 4, s));
    unsigned long long *d = buf.as<unsigned long long>();
    if (bnnz > 0) {
        const int grid = (int)std::min<long long>((bnnz + 255) / 256, 512);   // (one atomic per workgroup on one counter: 4 096 of them were 20 of the pass's 28 µs on a 5.6e5-entry B)
        hipLaunchKernelGGL(check_descents_kernel, dim3(grid), dim3(256), 0, s, bcol, bnnz, N, reinterpret_cast<int *>(d), d + 1);
        hipLaunchKernelGGL(row_start_descents_kernel, dim3(std::min((K + 255) / 256, 512)), dim3(256), 0, s, K, brpt, bcol, d + 2);
    }
    G4S_TRY(f.alloc(sizeof(long long) * ((size_t)annz + 1)));
    G4S_TRY(P.alloc(sizeof(long long) * ((size_t)annz + 1)));
    hipLaunchKernelGGL(entry_flop_checked_kernel, dim3((unsigned)((annz + 256) / 256)), dim3(256), 0, s, annz, acol, K, brpt, f.as<long long>(), d + 3);
    G4S_TRY(g4s::prims::exclusive_scan(f.as<long long>(), P.as<long long>(), annz + 1, s));
    hipLaunchKernelGGL(row_flop_from_scan_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, arpt, P.as<long long>(), d_row_flop);
    G4S_HIP_TRY(hipGetLastError());
    unsigned long long h[4] = {0, 0, 0, 0};
    long long tot = 0;
    G4S_HIP_TRY(g4s::read_small(h, d, sizeof(h), s));
    G4S_HIP_TRY(g4s::read_small(&tot, P.as<long long>() + annz, sizeof(tot), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (h[3]) return g4s::set_error(G4S_ERR_INVALID, "SpGEMM: a column id of A outside its valid range [0,%d)", K);
    if (h[0] & 0xffffffffull) return g4s::set_error(G4S_ERR_INVALID, "SpGEMM: a column id of B outside its valid range [0,%d)", N);
    *b_unsorted = h[1] != h[2];                                     // (the flop does not depend on the order inside B's rows: everything computed here stands)
    if (total) *total = (int64_t)tot;
    return G4S_OK;
}

// the last entries of two row-pointer arrays (nnz(A), nnz(B)) in one host wait
int read_last2(const int *a_rpt, int na, int *a_out, const int *b_rpt, int nb, int *b_out, hipStream_t s)
{
    G4S_HIP_TRY(g4s::read_small(a_out, a_rpt + na, sizeof(int), s));
    G4S_HIP_TRY(g4s::read_small(b_out, b_rpt + nb, sizeof(int), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    return G4S_OK;
}

int read_last(const int *d_rpt, int n, int *out, hipStream_t s)
{
    G4S_HIP_TRY(g4s::read_small(out, d_rpt + n, sizeof(int), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    return G4S_OK;
}

} // namespace

// ================================================================================================ C-ABI
namespace {
// Sorted columns carried from the symbolic to the numeric phase of the one-shot call (g4s_spgemm_csr_i32_f64): rows counted by the
// window kernel whose product bound lies in (1 K, 128 K] — the rows the numeric big-row kernel will take — also write their distinct
// columns, in order, to cols[off[row] …] (off[row] = −1 for every other row); the big-row kernel then skips its own mark-and-emit
// phase for them. The scratch is sized by the bound Σ min(flop_i, N), so it is only used while that fits comfortably in free HBM.
constexpr long long kPresortMinFlop = 512;    // every row that reaches a window kernel in the numeric phase (nz > 512) carries its sorted columns over
// Column compaction for the bitmap-window kernels. The windows span B's COLUMN RANGE, and a power-law B leaves much of it unused
// (R-MAT scale 21: 59 % of the columns hold no entry): renumbering the non-empty columns 0 … N2 − 1 in order shrinks every bitmap pass
// by that share (two 2^20 windows become one) and keeps sorted rows sorted. The window kernels then work on bcol2 (compact ids) and
// translate back through inv when they write ccol; the table and hub kernels keep the original ids. Left out when fewer than an
// eighth of the columns are empty.
struct ColumnMap {
    DevBuf bcol2, inv;           // int[nnz(B)] compact id per entry; int[N2] original id per compact id
    int n2 = 0;                  // 0: not in use
    bool keep = false;           // the map outlives the API call that builds it (g4s_spgemm_symbolic → g4s_spgemm_numeric): its arrays come from the block cache, not from the call's arena
    const int *cols(const int *bcol) const { return n2 ? bcol2.as<int>() : bcol; }
    int width(int N) const { return n2 ? n2 : N; }
    const int *inverse() const { return n2 ? inv.as<int>() : nullptr; }
};
// (Round 4: one BYTE per column, set with plain stores — every writer stores the same 1, so the race is benign — and packed into bitmap words by the count
// kernel; the atomicOr per entry on a bitmap whose popular words every wave hits cost 0.3 ms per product.)
__global__ void colmap_mark_kernel(long long nnz, int N, const int *__restrict__ bcol, unsigned char *__restrict__ seen)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int c = bcol[k];
    if (c < 0 || c >= N) return;                                   // the numeric-only call does not range-check B again: never write outside the map
    seen[c] = 1;
}
__global__ void colmap_popc_kernel(int W, int N, const unsigned char *__restrict__ seen, unsigned *__restrict__ bm, int *__restrict__ cnt)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > W) return;                                             // W + 1 items: the scan's last element is the total
    unsigned word = 0;
    if (w < W) {
        const uint4 *p = reinterpret_cast<const uint4 *>(seen + (size_t)w * 32);   // (the byte map is padded to a multiple of 32)
        const uint4 a = p[0], b = p[1];
        const unsigned v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) word |= ((v[q] >> (8 * j)) & 1u) << (4 * q + j);
        bm[w] = word;
    }
    cnt[w] = __popc(word);
}
__global__ void colmap_inverse_kernel(int W, const unsigned *__restrict__ bm, const int *__restrict__ prefix, int *__restrict__ inv)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    unsigned bits = bm[w];
    int pos = prefix[w];
    while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        inv[pos++] = (w << 5) + bit;
    }
}
__global__ void colmap_apply_kernel(long long nnz, const int *__restrict__ bcol, const unsigned *__restrict__ bm, const int *__restrict__ prefix,
                                    int *__restrict__ bcol2)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int c = bcol[k];
    bcol2[k] = prefix[c >> 5] + __popc(bm[c >> 5] & ((1u << (c & 31)) - 1u));
}
int build_column_map(int N, long long bnnz, const int *bcol, ColumnMap &cm, hipStream_t s)
{
    cm.n2 = 0;
    if (N < (1 << 16) || bnnz <= 0 || getenv("G4S_SPGEMM_NO_COLMAP")) return G4S_OK;   // a single small window either way
    const int W = (N + 31) >> 5;
    DevBuf bm, cnt, prefix, seen;
    G4S_TRY(bm.alloc(sizeof(unsigned) * (size_t)W));
    G4S_TRY(seen.alloc((size_t)W * 32));
    G4S_HIP_TRY(hipMemsetAsync(seen.p, 0, (size_t)W * 32, s));
    G4S_TRY(cnt.alloc(sizeof(int) * ((size_t)W + 1)));
    G4S_TRY(prefix.alloc(sizeof(int) * ((size_t)W + 1)));
    hipLaunchKernelGGL(colmap_mark_kernel, dim3((unsigned)((bnnz + 255) / 256)), dim3(256), 0, s, bnnz, N, bcol, seen.as<unsigned char>());
    hipLaunchKernelGGL(colmap_popc_kernel, dim3((W + 256) / 256), dim3(256), 0, s, W, N, seen.as<unsigned char>(), bm.as<unsigned>(), cnt.as<int>());
    G4S_TRY(g4s::prims::exclusive_scan(cnt.as<int>(), prefix.as<int>(), (long long)W + 1, s));
    int n2 = 0;
    G4S_HIP_TRY(g4s::read_small(&n2, prefix.as<int>() + W, sizeof(int), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (n2 <= 0 || (long long)n2 * 8 > (long long)N * 7) return G4S_OK;
    G4S_TRY(cm.bcol2.alloc(sizeof(int) * (size_t)bnnz, cm.keep));
    G4S_TRY(cm.inv.alloc(sizeof(int) * (size_t)n2, cm.keep));
    hipLaunchKernelGGL(colmap_inverse_kernel, dim3((W + 255) / 256), dim3(256), 0, s, W, bm.as<unsigned>(), prefix.as<int>(), cm.inv.as<int>());
    hipLaunchKernelGGL(colmap_apply_kernel, dim3((unsigned)((bnnz + 255) / 256)), dim3(256), 0, s, bnnz, bcol, bm.as<unsigned>(), prefix.as<int>(),
                       cm.bcol2.as<int>());
    G4S_HIP_TRY(hipGetLastError());                               // (bm / prefix / seen are released in stream order — or with the call's arena: no wait here)
    cm.n2 = n2;
    if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s column map: %d of %d columns of B hold entries\n", n2, N);
    return G4S_OK;
}

struct PreSorted {
    ColumnMap cmap;              // built by the symbolic phase of the one-shot call, reused by its numeric phase
    DevBuf wsplit_buf;           // likewise the window splits of B
    const int *wsplit = nullptr;
    bool has_wsplit = false;
    DevBuf off;                  // long long off[M]
    const long long *d_off = nullptr;
    int *d_cols = nullptr;       // int cols[total], borrowed from the cache below
    bool holds_cache = false;
    bool complete = false;       // every row the numeric window kernels will take carries its columns (no hub rows, no overflowed optimistic tables)
    long long flop = -1;         // the product's flop: bounds the unit lists of the numeric launches without a count read back
    // round 5 (spgemm_rank.hpp): the rows of the long symbolic classes carry CUTS instead of columns — cut_off[row] >= 0 marks them, cuts[cut_off[row] …] holds
    // nseg segment starts, then the count cuts
    DevBuf cut_off_buf, cuts_buf;
    const long long *d_cut_off = nullptr;
    int *d_cuts = nullptr;
    int nseg = 0;
    long long bnnz = -1, annz = -1;
    bool rank = false;
    // B arrived with unsorted rows (sort_b_rows): the sorted copy of its columns — what every kernel of the product reads — and the permutation behind it (the
    // numeric phase gathers the values through it)
    DevBuf b_cols_sorted, b_perm;
    bool b_unsorted = false;
    unsigned long long key_hash = 0;   // pattern_hash_kernel over the caller's five index arrays (two-call form)
    // A product whose rows all have at most 512 products (the reference's own examples: can_24, patents_main) needs no column map, no window splits, no column
    // scratch, and its numeric phase takes the symbolic phase's row lists (nz ≤ flop ≤ 512: the same rows go to the same kernel) — one-call form only, the lists
    // live in the call's arena.
    RowClasses sym_rc;
    bool sym_rc_valid = false;
    bool keep = false;           // carried from g4s_spgemm_symbolic to the g4s_spgemm_numeric call that follows it (see CarriedSymbolic): nothing of it may live in a call's arena
    ~PreSorted();
};

// The column scratch is the largest transient of a product (8 GB on config 3) and has the same size from call to call: one cached
// block per process, handed to one call at a time (a concurrent call simply runs without the scratch), released by g4s_shutdown.
// Leaving it to the pool made the allocator split the freed output blocks differently from call to call, and every other call then
// paid a fresh 15 GB driver allocation (1–2 s).
struct ColumnScratchCache {
    std::mutex m;
    void *p = nullptr;
    size_t bytes = 0;
    bool in_use = false;
} g_col_cache;

int *acquire_column_scratch(size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_col_cache.m);
    if (g_col_cache.in_use) return nullptr;
    if (g_col_cache.bytes < bytes) {
        if (g_col_cache.p) { (void)hipFree(g_col_cache.p); g_col_cache.p = nullptr; g_col_cache.bytes = 0; }
        if (hipMalloc(&g_col_cache.p, bytes) != hipSuccess) { (void)hipGetLastError(); g_col_cache.p = nullptr; return nullptr; }
        g_col_cache.bytes = bytes;
    }
    g_col_cache.in_use = true;
    return static_cast<int *>(g_col_cache.p);
}
void release_column_scratch()
{
    std::lock_guard<std::mutex> lock(g_col_cache.m);
    g_col_cache.in_use = false;
}
PreSorted::~PreSorted() { if (holds_cache) release_column_scratch(); }

// A class's rows, longest first, with the counter the persistent kernels hand them out through. The key of a row is size[row]
// (64-bit, clipped to int) or, without size, its output length crpt[row + 1] − crpt[row].
__global__ void row_size_keys_kernel(int n, const int *__restrict__ rows, const long long *__restrict__ size, const int *__restrict__ crpt, int *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    const long long v = size ? size[r] : (long long)crpt[r + 1] - crpt[r];
    // The order only decides which rows the persistent kernels take first (longest first, for the tail): an 11-bit logarithmic key —
    // position of the leading one, then the six bits behind it — is monotone in the size and sorts in 3 radix passes instead of 8.
    const unsigned u = v > INT_MAX ? (unsigned)INT_MAX : (v < 0 ? 0u : (unsigned)v);
    const int e = u ? 31 - __clz(u) : 0;
    keys[i] = u ? ((e + 1) << 6) | (int)(((unsigned long long)u << (32 - e)) >> 26 & 63u) : 0;
}
struct SortedRows {
    DevBuf keys, keys_sorted, rows, tmp, counter;
    int build(int n, const int *list, const long long *size, const int *crpt, long long key_bound, hipStream_t s)
    {
        G4S_TRY(keys.alloc(sizeof(int) * (size_t)n));
        G4S_TRY(keys_sorted.alloc(sizeof(int) * (size_t)n));
        G4S_TRY(rows.alloc(sizeof(int) * (size_t)n));
        G4S_TRY(counter.alloc(sizeof(int)));
        G4S_HIP_TRY(hipMemsetAsync(counter.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(row_size_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, list, size, crpt, keys.as<int>());
        (void)key_bound;
        if (n <= g4s::prims::kSortSmallMax)                       // a short list: one launch (the keys are < 2^11; the order among equal keys decides nothing but who takes a row first)
            return g4s::prims::sort_pairs_descending_small_unstable(keys.as<int>(), list, keys_sorted.as<int>(), rows.as<int>(), n, s);
        G4S_TRY(tmp.alloc(sizeof(int) * 2 * (size_t)n));
        G4S_TRY(g4s::prims::sort_pairs_descending(keys.as<int>(), list, keys_sorted.as<int>(), rows.as<int>(), tmp.as<int>(), tmp.as<int>() + n, n, 11, s));   // (31 + 1) << 6 | 63 < 2^11
        return G4S_OK;
    }
};
__global__ void presorted_need_kernel(int M, const int *__restrict__ cls, unsigned class_mask, const long long *__restrict__ row_flop, int N,
                                      long long min_flop, long long *__restrict__ need)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const long long f = row_flop[i];
    const bool take = ((class_mask >> cls[i]) & 1u) && f > min_flop;
    need[i] = take ? (f < N ? f : (long long)N) : 0;
}
__global__ void rank_need_kernel(int M, const int *__restrict__ cls, unsigned class_mask, const long long *__restrict__ row_flop, int N, int nseg, long long *__restrict__ need)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > M) return;
    long long v = 0;
    if (i < M && ((class_mask >> cls[i]) & 1u)) {
        const long long f = row_flop[i], nzb = f < N ? f : (long long)N;   // the row's outputs are at most min(flop, columns): so many count cuts at most
        v = nseg + nzb / kRankCut + 1;
    }
    need[i] = v;
}
__global__ void presorted_mark_kernel(int M, const long long *__restrict__ need, long long *__restrict__ off)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M && need[i] == 0) off[i] = -1;
}
} // namespace

namespace {
// wsplit for the window kernels (see window_splits_kernel); left empty (kernels then read whole rows) for a single window or when
// the table would be large (more than 16 windows).
int build_window_splits(int K, int N, const int *brpt, const int *bcol, DevBuf &buf, const int **out, hipStream_t s, bool keep = false)
{
    *out = nullptr;
    const int sb = split_bits(N), W = (N + (1 << sb) - 1) >> sb;
    if (W < 2 || W > 16 || K <= 0) return G4S_OK;
    const long long total = (long long)K * (W - 1);
    G4S_TRY(buf.alloc(sizeof(int) * (size_t)total, keep));
    hipLaunchKernelGGL(window_splits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, K, W, sb, brpt, bcol, buf.as<int>());
    G4S_HIP_TRY(hipGetLastError());
    *out = buf.as<int>();
    return G4S_OK;
}
} // namespace

namespace {
int spgemm_symbolic_impl(int32_t M, int32_t K, int32_t N, const int32_t *arpt, const int32_t *acol, const int32_t *brpt, const int32_t *bcol,
                         int32_t *crpt, int64_t *cnnz, void *stream, PreSorted *pre)
{
    G4S_REQUIRE(M >= 0 && K >= 0 && N >= 0, "negative dimension");
    G4S_REQUIRE(arpt && brpt && crpt && cnnz, "NULL argument");
    hipStream_t s = g4s::as_stream(stream);
    t_stream = s;
    t_idle = false;
    *cnnz = 0;
    if (M == 0) { G4S_HIP_TRY(hipMemsetAsync(crpt, 0, sizeof(int), s)); return G4S_OK; }
    DbgPhases dbg("symbolic");
    ArenaScope arena;
    int annz = 0, bnnz = 0;
    G4S_TRY(read_last2(arpt, M, &annz, brpt, K, &bnnz, s));

    DevBuf row_flop, row_nz, ovf_rows, ovf_count;
    G4S_TRY(row_flop.alloc(sizeof(long long) * (size_t)M));
    G4S_TRY(row_nz.alloc(sizeof(int) * ((size_t)M + 1)));
    G4S_HIP_TRY(hipMemsetAsync(row_nz.p, 0, sizeof(int) * ((size_t)M + 1), s));
    int64_t flop = 0;
    bool b_unsorted = false;
    const int32_t *bcol_caller = bcol;
    G4S_TRY(checked_row_flop(M, K, N, arpt, acol, annz, brpt, bcol, bnnz, row_flop.as<long long>(), &flop, s, &b_unsorted));
    DevBuf local_bs, local_perm;
    if (b_unsorted) {                                              // the product runs on a private copy of B with sorted rows (sort_b_rows)
        DevBuf &bs = pre ? pre->b_cols_sorted : local_bs, &pm = pre ? pre->b_perm : local_perm;
        const bool keep = pre && pre->keep;
        G4S_TRY(bs.alloc(sizeof(int) * (size_t)bnnz, keep));
        G4S_TRY(pm.alloc(sizeof(int) * (size_t)bnnz, keep));
        G4S_TRY(sort_b_rows(K, N, bnnz, brpt, bcol, pm.as<int>(), s));
        G4S_TRY(gather_b(bnnz, pm.as<int>(), bcol, nullptr, bs.as<int>(), nullptr, s));
        bcol = bs.as<int>();
        if (pre) pre->b_unsorted = true;
        if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s symbolic: the rows of B are not sorted by column — working on a sorted private copy\n");
    }

    dbg.mark("row_flop");
    RowClasses local_rc;
    RowClasses &rc = pre ? pre->sym_rc : local_rc;
    G4S_TRY(classify_rows(M, row_flop.as<long long>(), kSymLimits, N, rc, s));
    const bool only_short = rc.count[CLS_MEDIUM] + rc.count[CLS_LARGE] + rc.count[CLS_M2] + rc.count[CLS_M3] + rc.count[CLS_HUB] == 0;   // every row ≤ 512 products
    if (pre) pre->sym_rc_valid = only_short && !pre->keep;
    dbg.mark("classes");
    if (getenv("G4S_DEBUG"))
        fprintf(stderr, "g4s symbolic classes: empty %d tiny %d small %d medium %d large %d hub %d (flop %lld)\n", rc.count[CLS_EMPTY], rc.count[CLS_TINY],
                rc.count[CLS_SMALL], rc.count[CLS_MEDIUM], rc.count[CLS_LARGE], rc.count[CLS_HUB], (long long)flop);
    // the bitmap-window kernels work on B's non-empty columns, renumbered (N2 of them; the original ids when that would not pay)
    ColumnMap local_map;
    ColumnMap &cmap = pre ? pre->cmap : local_map;
    if (!only_short) G4S_TRY(build_column_map(N, bnnz, bcol, cmap, s));   // (left out, the map is the identity)
    const int *wcol = cmap.cols(bcol);
    const int N2 = cmap.width(N);
    DevBuf wsplit_local;
    const int *wsplit = nullptr;
    if (!only_short) {
        G4S_TRY(build_window_splits(K, N2, brpt, wcol, pre ? pre->wsplit_buf : wsplit_local, &wsplit, s, pre && pre->keep));
        if (pre) { pre->wsplit = wsplit; pre->has_wsplit = true; }
    }
    dbg.mark("colmap+splits");
    G4S_TRY(ovf_rows.alloc(sizeof(int) * (size_t)std::max(1, rc.count[CLS_LARGE])));
    G4S_TRY(ovf_count.alloc(sizeof(int)));
    G4S_HIP_TRY(hipMemsetAsync(ovf_count.p, 0, sizeof(int), s));
    int *nz = row_nz.as<int>();

    // rows of at most 512 products (the tiny and the small class: their lists are adjacent): one wavefront merges a row (spgemm_small_wave_kernel); what does not
    // fit it (more than 64 A-entries) lands on a list that the table kernel takes, its count read on the device
    DevBuf small_ovf, small_ovf_n;
    const bool use_wave = !getenv("G4S_SPGEMM_NO_WAVE_ROWS");
    if (int n = rc.count[CLS_TINY] + rc.count[CLS_SMALL]; n && use_wave) {
        G4S_TRY(small_ovf.alloc(sizeof(int) * (size_t)n));
        G4S_TRY(small_ovf_n.alloc(sizeof(int)));
        G4S_HIP_TRY(hipMemsetAsync(small_ovf_n.p, 0, sizeof(int), s));
        if (int nt = rc.count[CLS_TINY]) {                          // at most 32 products each
            auto k = spgemm_small_wave_kernel<false, kTinyCap>;
            constexpr size_t lds = sizeof(int) * 4 * small_wave_ints<false, kTinyCap>();
            hipLaunchKernelGGL(k, dim3((nt + 3) / 4), dim3(256), lds, s, rc.list(CLS_TINY), nt, arpt, acol, (const double *)nullptr, brpt, bcol,
                               (const double *)nullptr, nz, (const int *)nullptr, (int *)nullptr, (double *)nullptr, small_ovf.as<int>(), small_ovf_n.as<int>());
        }
        if (int ns = rc.count[CLS_SMALL]) {
            auto k = spgemm_small_wave_kernel<false>;
            hipLaunchKernelGGL(k, dim3((ns + 3) / 4), dim3(256), sizeof(int) * 4 * small_wave_ints<false>(), s, rc.list(CLS_SMALL), ns, arpt, acol, (const double *)nullptr, brpt, bcol,
                               (const double *)nullptr, nz, (const int *)nullptr, (int *)nullptr, (double *)nullptr, small_ovf.as<int>(), small_ovf_n.as<int>());
        }
        auto k2 = spgemm_symbolic_lds_kernel<256, 256, 1024, false>;
        hipLaunchKernelGGL(k2, dim3(std::min(n, 256)), dim3(256), sym_lds_bytes(1, 1024), s, small_ovf.as<int>(), 0, arpt, acol, brpt, bcol, row_flop.as<long long>(), nz, nullptr, nullptr,
                           (const int *)small_ovf_n.as<int>());
    } else {
    if (int n = rc.count[CLS_TINY]) {
        auto k = spgemm_symbolic_lds_kernel<256, 64, 64, false>;
        hipLaunchKernelGGL(k, dim3((n + 3) / 4), dim3(256), sym_lds_bytes(4, 64), s, rc.list(CLS_TINY), n, arpt, acol, brpt, bcol, row_flop.as<long long>(), nz, nullptr, nullptr, (const int *)nullptr);
    }
    if (int n = rc.count[CLS_SMALL]) {
        auto k = spgemm_symbolic_lds_kernel<256, 256, 1024, false>;
        hipLaunchKernelGGL(k, dim3(n), dim3(256), sym_lds_bytes(1, 1024), s, rc.list(CLS_SMALL), n, arpt, acol, brpt, bcol, row_flop.as<long long>(), nz, nullptr, nullptr, (const int *)nullptr);
    }
    }
    // Up to kWindowMaxN columns (4 bitmap windows) the window kernel beats the key tables for every row past 512 products (it has no
    // probe chains and cannot overflow); with more windows each row would re-walk its products once per window, so tables take over.
    const bool x_large = N2 <= window_max_n(), x_med = x_large && !mid_tables();
    const long long *pre_off = nullptr;
    int *pre_cols = nullptr;
    const bool use_units = !getenv("G4S_SPGEMM_NO_UNITS");
    // The long classes' shape follows B's width (round 5): the count-and-cut pass is barriers and round trips per window, not bitmap work, so four 256-thread
    // workgroups per CU that fill each other's waits beat one of 1 024 threads even with five windows per row instead of two (configs[2]: 26.55 against 27.5 ms,
    // 512 threads 27.1: profiles/r05_spgemm_ab.txt) — while the number of windows stays small.
    const int long_shape = N2 <= 6 * (1 << 18) ? 256 : N2 <= 6 * (1 << 19) ? 512 : 1024;
    const int t_med = shape_of("G4S_SPGEMM_T_SYM_MED", kShapeSymMedium), t_large = shape_of("G4S_SPGEMM_T_SYM_LARGE", long_shape),
              t_win = shape_of("G4S_SPGEMM_T_SYM_WINDOW", long_shape);
    const bool one_long_launch = x_large && t_large == t_win;   // LARGE and M2 share a shape and their lists are adjacent: one launch
    // Round 5: the long classes (more than 8 K products) write CUTS instead of columns and take the rank kernel in the numeric phase (spgemm_rank.hpp). Needs what the
    // carried columns needed (a PreSorted to carry them in) plus the unit lists and the one long launch.
    const int n_long = rc.count[CLS_LARGE] + rc.count[CLS_M2];
    bool use_rank = pre && one_long_launch && use_units && n_long > 0 && !getenv("G4S_SPGEMM_NO_RANK");
    const int nseg = rank_segments(N2);
    long long *cut_off = nullptr;
    int *cuts = nullptr;
    if (use_rank) {
        const long long cut_bound = (long long)n_long * (nseg + 1) + flop / kRankCut + 1;   // Σ (nseg + min(flop_i, N2) / chunk + 1) over the long rows: no count comes back
        DevBuf need;
        if (cut_bound >= (1ll << 30) || need.alloc(sizeof(long long) * ((size_t)M + 1)) != G4S_OK || pre->cut_off_buf.alloc(sizeof(long long) * ((size_t)M + 1), pre->keep) != G4S_OK ||
            pre->cuts_buf.alloc(sizeof(int) * (size_t)cut_bound, pre->keep) != G4S_OK) { (void)hipGetLastError(); use_rank = false; }
        else {
            const unsigned long_mask = (1u << CLS_LARGE) | (1u << CLS_M2);
            hipLaunchKernelGGL(rank_need_kernel, dim3((M + 256) / 256), dim3(256), 0, s, M, rc.cls.as<int>(), long_mask, row_flop.as<long long>(), N2, nseg, need.as<long long>());
            G4S_TRY(g4s::prims::exclusive_scan(need.as<long long>(), pre->cut_off_buf.as<long long>(), (long long)M + 1, s));
            hipLaunchKernelGGL(presorted_mark_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, need.as<long long>(), pre->cut_off_buf.as<long long>());
            G4S_HIP_TRY(hipGetLastError());
            cut_off = pre->cut_off_buf.as<long long>(); cuts = pre->cuts_buf.as<int>();
        }
    }
    if (pre && !only_short) {
        // which classes the window kernel counts AND EMITS in this call: M2 always, MEDIUM / LARGE while B is narrow enough — without the classes that write cuts
        const unsigned class_mask = use_rank ? (x_med ? (1u << CLS_MEDIUM) : 0u) : ((1u << CLS_M2) | (x_med ? (1u << CLS_MEDIUM) : 0u) | (x_large ? (1u << CLS_LARGE) : 0u));
        DevBuf need;
        G4S_TRY(need.alloc(sizeof(long long) * ((size_t)M + 1)));
        G4S_TRY(pre->off.alloc(sizeof(long long) * ((size_t)M + 1), pre->keep));
        G4S_HIP_TRY(hipMemsetAsync(need.p, 0, sizeof(long long) * ((size_t)M + 1), s));
        const long long min_flop = kPresortMinFlop;
        hipLaunchKernelGGL(presorted_need_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, rc.cls.as<int>(), class_mask, row_flop.as<long long>(), N2, min_flop, need.as<long long>());
        G4S_TRY(g4s::prims::exclusive_scan(need.as<long long>(), pre->off.as<long long>(), (long long)M + 1, s));
        long long total_cols = 0;
        G4S_HIP_TRY(g4s::read_small(&total_cols, pre->off.as<long long>() + M, sizeof(long long), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        size_t free_b = 0, total_b = 0;
        G4S_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        int *cols = nullptr;
        if (total_cols > 0 && ((size_t)total_cols * sizeof(int) <= free_b / 4 || (size_t)total_cols * sizeof(int) <= g_col_cache.bytes))
            cols = acquire_column_scratch(sizeof(int) * (size_t)total_cols);
        if (cols) {
            pre->holds_cache = true;
            hipLaunchKernelGGL(presorted_mark_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, need.as<long long>(), pre->off.as<long long>());
            G4S_HIP_TRY(hipGetLastError());                        // (need is released in stream order — or with the call's arena: no wait here)
            pre->d_off = pre_off = pre->off.as<long long>();
            pre->d_cols = pre_cols = cols;
            if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s symbolic: %.2f GB scratch for pre-sorted columns\n", total_cols * 4 / 1e9);
        }
    }
    dbg.mark("tables+presort");
    std::vector<std::unique_ptr<DevBuf>> unit_keep;                // unit lists of the window launches: live until the stream is synchronised below
    bool rank_done = false;                                        // the long launch ran in its cut-writing form
    auto window_t = [&](auto shape, const int *rows, int n, const long long *poff, int *pcols, int *next_row, bool cuts_mode) -> int {
        constexpr int T = decltype(shape)::value;
        if (!n) return G4S_OK;
        const size_t lds = big_lds_bytes<T>();
        const dim3 grid(big_grid(n, BigCfg<T>::kPerCu));
        // unit lists (sym_unit_kernel), sized by bounds the host already has — items <= nnz(A) · windows, units <= flop / 64 + items — so that no count has to
        // come back from the device in front of the launch
        const int nwin = (N2 + (1 << BigCfg<T>::kWindowBits) - 1) >> BigCfg<T>::kWindowBits;
        const long long ibound = (long long)annz * nwin, ubound = flop / 64 + ibound;
        if (use_units && ibound > 0 && ibound <= (1ll << 28) && ubound <= (1ll << 28)) {
            auto items = std::make_unique<DevBuf>(), ioff = std::make_unique<DevBuf>(), ucnt = std::make_unique<DevBuf>(), uoff = std::make_unique<DevBuf>(), ud = std::make_unique<DevBuf>();
            bool ok = items->alloc(sizeof(long long) * ((size_t)n + 1)) == G4S_OK && ioff->alloc(sizeof(long long) * ((size_t)n + 1)) == G4S_OK &&
                      ucnt->alloc(sizeof(int) * ((size_t)ibound + 1)) == G4S_OK && uoff->alloc(sizeof(int) * ((size_t)ibound + 1)) == G4S_OK &&
                      ud->alloc(sizeof(SymUnit) * ((size_t)ubound + 1)) == G4S_OK;
            if (!ok) (void)hipGetLastError();
            if (ok) {
                hipLaunchKernelGGL(sym_items_kernel, dim3((n + 256) / 256), dim3(256), 0, s, n, rows, arpt, nwin, items->as<long long>());
                G4S_TRY(g4s::prims::exclusive_scan(items->as<long long>(), ioff->as<long long>(), (long long)n + 1, s));
                const unsigned ugrid = (unsigned)((ibound + 256) / 256);
                hipLaunchKernelGGL(sym_unit_kernel<false>, dim3(ugrid), dim3(256), 0, s, ibound, n, rows, ioff->as<long long>(), arpt, acol, brpt, K, N2, wsplit, BigCfg<T>::kWindowBits,
                                   ucnt->as<int>(), (const int *)nullptr, (SymUnit *)nullptr);
                G4S_TRY(g4s::prims::exclusive_scan(ucnt->as<int>(), uoff->as<int>(), ibound + 1, s));
                hipLaunchKernelGGL(sym_unit_kernel<true>, dim3(ugrid), dim3(256), 0, s, ibound, n, rows, ioff->as<long long>(), arpt, acol, brpt, K, N2, wsplit, BigCfg<T>::kWindowBits,
                                   (int *)nullptr, uoff->as<int>(), ud->as<SymUnit>());
                auto rmeta = std::make_unique<DevBuf>();
                G4S_TRY(rmeta->alloc(sizeof(SymRowMeta) * (size_t)n));
                hipLaunchKernelGGL(sym_row_meta_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, rows, arpt, poff, ioff->as<long long>(), uoff->as<int>(), rmeta->as<SymRowMeta>());
                if (cuts_mode) {
                    auto k = spgemm_symbolic_units_kernel<T, true>;
                    G4S_TRY(allow_lds(k, lds));
                    hipLaunchKernelGGL(k, grid, dim3(T), lds, s, rows, n, next_row, N2, arpt, wcol, nz, poff, pcols, ioff->as<long long>(), uoff->as<int>(), ud->as<SymUnit>(), rmeta->as<SymRowMeta>());
                    rank_done = true;
                } else {
                    auto k = spgemm_symbolic_units_kernel<T, false>;
                    G4S_TRY(allow_lds(k, lds));
                    hipLaunchKernelGGL(k, grid, dim3(T), lds, s, rows, n, next_row, N2, arpt, wcol, nz, poff, pcols, ioff->as<long long>(), uoff->as<int>(), ud->as<SymUnit>(), rmeta->as<SymRowMeta>());
                }
                G4S_HIP_TRY(hipGetLastError());
                unit_keep.push_back(std::move(rmeta));
                unit_keep.push_back(std::move(items)); unit_keep.push_back(std::move(ioff)); unit_keep.push_back(std::move(ucnt)); unit_keep.push_back(std::move(uoff)); unit_keep.push_back(std::move(ud));
                return G4S_OK;
            }
        }
        if (cuts_mode) { poff = nullptr; pcols = nullptr; }       // (no unit lists: the rows are counted only, and the numeric phase emits their columns itself as in the two-call form)
        auto k = spgemm_symbolic_window_kernel<T>;
        G4S_TRY(allow_lds(k, lds));
        hipLaunchKernelGGL(k, grid, dim3(T), lds, s, rows, n, next_row, N2, K, wsplit, arpt, acol, brpt, wcol, row_flop.as<long long>(), nz, poff, pcols, (const int *)nullptr, 0, 0);
        return G4S_OK;
    };
    SortedRows sorted[3];                                          // live until the stream is synchronised below
    int n_sorted = 0;
    auto window = [&](int threads, const int *rows, int n, const long long *poff, int *pcols, bool longest_first = false, bool cuts_mode = false) -> int {
        int *next_row = nullptr;
        if (longest_first && n > 1 && !getenv("G4S_SPGEMM_STATIC_ROWS")) {
            SortedRows &sr = sorted[n_sorted++];
            G4S_TRY(sr.build(n, rows, row_flop.as<long long>(), nullptr, INT_MAX, s));
            rows = sr.rows.as<int>(); next_row = sr.counter.as<int>();
        }
        if (threads == 256) return window_t(std::integral_constant<int, 256>{}, rows, n, poff, pcols, next_row, cuts_mode);
        if (threads == 512) return window_t(std::integral_constant<int, 512>{}, rows, n, poff, pcols, next_row, cuts_mode);
        return window_t(std::integral_constant<int, 1024>{}, rows, n, poff, pcols, next_row, cuts_mode);
    };
    // (Round 5 measured the long classes' launch on a side stream, its pre-passes beside the mid-size class's kernel, and the same for the numeric phase: 31.9 / 31.0 ms
    // against 31.3 / 31.6 on one stream — nothing. The persistent kernels run four waves of 128 registers per SIMD: no wave of another kernel fits beside them.)
    if (x_med) { G4S_TRY(window(t_med, rc.list(CLS_MEDIUM), rc.count[CLS_MEDIUM], pre_off, pre_cols, true)); }
    else if (int n = rc.count[CLS_MEDIUM]) {
        auto k = spgemm_symbolic_lds_kernel<256, 256, 16384, false>;
        G4S_TRY(allow_lds(k, sym_lds_bytes(1, 16384)));
        hipLaunchKernelGGL(k, dim3(n), dim3(256), sym_lds_bytes(1, 16384), s, rc.list(CLS_MEDIUM), n, arpt, acol, brpt, bcol, row_flop.as<long long>(), nz, nullptr, nullptr, (const int *)nullptr);
    }
    if (one_long_launch && use_rank) { G4S_TRY(window(t_large, rc.list(CLS_LARGE), n_long, cut_off, cuts, true, true)); }
    else if (one_long_launch) { G4S_TRY(window(t_large, rc.list(CLS_LARGE), rc.count[CLS_LARGE] + rc.count[CLS_M2], pre_off, pre_cols, true)); }
    else if (x_large) { G4S_TRY(window(t_large, rc.list(CLS_LARGE), rc.count[CLS_LARGE], pre_off, pre_cols, true)); }
    else if (int n = rc.count[CLS_LARGE]) {
        auto k = spgemm_symbolic_lds_kernel<1024, 1024, 32768, true>;
        G4S_TRY(allow_lds(k, sym_lds_bytes(1, 32768)));
        hipLaunchKernelGGL(k, dim3(n), dim3(1024), sym_lds_bytes(1, 32768), s, rc.list(CLS_LARGE), n, arpt, acol, brpt, bcol, row_flop.as<long long>(), nz, ovf_rows.as<int>(),
                           ovf_count.as<int>(), (const int *)nullptr);
    }
    G4S_HIP_TRY(hipGetLastError());

    // rows too wide for a key table in LDS: the rows whose optimistic table filled up, and the window class → LDS bitmap windows
    int n_ovf = 0;
    if (!x_large) {                                                // (only the optimistic table kernel can overflow)
        G4S_HIP_TRY(g4s::read_small(&n_ovf, ovf_count.p, sizeof(int), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s symbolic: %d optimistic tables overflowed, %d window-class rows\n", n_ovf, rc.count[CLS_M2]);
        G4S_TRY(window(1024, ovf_rows.as<int>(), n_ovf, nullptr, nullptr));   // rows of the optimistic table class are not in the scratch
    }
    if (pre) {
        pre->rank = rank_done; pre->nseg = nseg; pre->bnnz = bnnz; pre->annz = annz;
        if (rank_done) { pre->d_cut_off = cut_off; pre->d_cuts = cuts; }
        // complete: every row the numeric WINDOW kernels will take carries its columns (the rows with cuts take the rank kernel and need none)
        pre->complete = pre_off != nullptr && n_ovf == 0 && rc.count[CLS_HUB] == 0 && x_med && x_large && (!use_rank || rank_done);
        pre->flop = flop;
    }
    if (!one_long_launch) G4S_TRY(window(t_win, rc.list(CLS_M2), rc.count[CLS_M2], pre_off, pre_cols, true));
    G4S_HIP_TRY(hipGetLastError());
    // hub rows (flop > 2 M): many workgroups per row on a bitmap in HBM
    std::vector<int> hub, ranges;
    G4S_TRY(fetch_rows_and_ranges(rc.list(CLS_HUB), rc.count[CLS_HUB], arpt, hub, ranges, s));
    G4S_TRY(run_hub_rows(false, hub, ranges, N, arpt, acol, nullptr, brpt, bcol, nullptr, nz, nullptr, nullptr, nullptr, s));

    dbg.mark("windows+hub(+sync)");
    // scan(bin.row_nz, crpt, nrow+1); *nnz = crpt[nrow]   (hash_mult.h:506-507)
    const int nblocks = (M + kScanChunk - 1) / kScanChunk;
    DevBuf block_sums;
    G4S_TRY(block_sums.alloc(sizeof(long long) * ((size_t)nblocks + 1)));
    G4S_HIP_TRY(hipMemsetAsync(block_sums.as<long long>() + nblocks, 0, sizeof(long long), s));   // the slot behind the sums becomes the total
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblocks), dim3(256), 0, s, M, nz, block_sums.as<long long>());
    // block sums → block offsets by one workgroup in parallel (a single thread walking the ≈ 1 000 sums took 116 µs of every call)
    hipLaunchKernelGGL(g4s::prims::scan_tile_offsets_kernel<long long>, dim3(1), dim3(g4s::prims::kScanThreads), 0, s, nblocks + 1, block_sums.as<long long>());
    long long h_total = 0;
    G4S_HIP_TRY(g4s::read_small(&h_total, block_sums.as<long long>() + nblocks, sizeof(long long), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    *cnnz = h_total;
    if (h_total > INT32_MAX)
        return g4s::set_error(G4S_ERR_OVERFLOW, "nnz(C) = %lld exceeds the reference's int32 row pointer (mm/inc/define.h:14)", h_total);
    hipLaunchKernelGGL(scan_write_kernel, dim3(nblocks), dim3(256), 0, s, M, nz, block_sums.as<long long>(), crpt);
    G4S_HIP_TRY(hipGetLastError());
    DevBuf hash_buf;
    unsigned long long h_hash = 0;
    if (pre && pre->keep) {                                        // the two-call form: what the numeric call must find unchanged to take this state over
        G4S_TRY(hash_buf.alloc(sizeof(unsigned long long)));
        G4S_TRY(enqueue_pattern_hash(M, K, annz, bnnz, arpt, acol, brpt, bcol_caller, crpt, hash_buf.as<unsigned long long>(), s));
        G4S_HIP_TRY(g4s::read_small(&h_hash, hash_buf.p, sizeof(h_hash), s));
    }
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (pre) pre->key_hash = h_hash;
    t_idle = true;
    dbg.mark("scan");
    return G4S_OK;
}
} // namespace

// The two-call form carries what the one-call form carries between its phases (round 4): g4s_spgemm_symbolic leaves the sorted columns of the window classes, the
// column map and the window splits behind, keyed by its arguments; the g4s_spgemm_numeric call that follows WITH THE SAME ARRAYS takes them over instead of
// marking and emitting every row again (33 → 20 ms on configs[2]; the symbolic call pays the emit: 7 → 11.6 ms) — and so does every further numeric call on
// that pattern (new values, same structure). One product at a time per process: the next
// symbolic or one-call product, g4s_trim and g4s_shutdown drop whatever is still held (the column scratch is one cached block per process). The arrays must not
// change between the two calls — the contract of the two-call form anyway: crpt describes THIS product. G4S_SPGEMM_NO_CARRY=1: as before.
namespace {
struct CarriedSymbolic {
    std::mutex m;
    std::shared_ptr<PreSorted> pre;                                // (shared: a numeric call in progress keeps it alive when another thread's product drops it)
    int M = 0, K = 0, N = 0;
    const void *arpt = nullptr, *acol = nullptr, *brpt = nullptr, *bcol = nullptr, *crpt = nullptr;
};
// (never destroyed: a product still carried when the process ends must not release device blocks from a static destructor, after the HIP runtime's own teardown)
CarriedSymbolic &g_carried = *new CarriedSymbolic();
void drop_carried()
{
    std::shared_ptr<PreSorted> old;
    { std::lock_guard<std::mutex> lock(g_carried.m); old = std::move(g_carried.pre); }
    // (destroyed outside the lock: its blocks go back to the cache behind a device synchronisation)
}
} // namespace

G4S_API g4s_status g4s_spgemm_symbolic(int32_t M, int32_t K, int32_t N,
                                       const int32_t *arpt, const int32_t *acol, const int32_t *brpt, const int32_t *bcol,
                                       int32_t *crpt, int64_t *cnnz, void *stream)
{
    drop_carried();
    if (getenv("G4S_SPGEMM_NO_CARRY")) return spgemm_symbolic_impl(M, K, N, arpt, acol, brpt, bcol, crpt, cnnz, stream, nullptr);
    auto pre = std::make_shared<PreSorted>();
    pre->keep = true; pre->cmap.keep = true;
    const int st = spgemm_symbolic_impl(M, K, N, arpt, acol, brpt, bcol, crpt, cnnz, stream, pre.get());
    if (st != G4S_OK) return st;
    std::lock_guard<std::mutex> lock(g_carried.m);
    g_carried.pre = std::move(pre);
    g_carried.M = M; g_carried.K = K; g_carried.N = N;
    g_carried.arpt = arpt; g_carried.acol = acol; g_carried.brpt = brpt; g_carried.bcol = bcol; g_carried.crpt = crpt;
    return G4S_OK;
}

namespace {
int spgemm_numeric_impl(int32_t M, int32_t K, int32_t N,
                        const int32_t *arpt, const int32_t *acol, const double *aval,
                        const int32_t *brpt, const int32_t *bcol, const double *bval,
                        const int32_t *crpt, int32_t *ccol, double *cval, unsigned flags, void *stream, const PreSorted *pre)
{
    const long long *pre_off = pre ? pre->d_off : nullptr;
    const int *pre_cols = pre ? pre->d_cols : nullptr;
    (void)flags; // rows always come out sorted by column: the sorted form is the only ordering contract (hash_mult.h:530-551)
    G4S_REQUIRE(M >= 0 && N >= 0, "negative dimension");
    G4S_REQUIRE(arpt && brpt && crpt, "NULL argument");
    hipStream_t s = g4s::as_stream(stream);
    t_stream = s;
    t_idle = false;
    if (M == 0) return G4S_OK;
    DbgPhases dbg("numeric");
    ArenaScope arena;
    // B with unsorted rows: the kernels read a sorted private copy (sort_b_rows) — the symbolic phase's columns and permutation when they were carried over, else made here.
    // (Done in front of the short-row kernels: every kernel of the phase reads the same B.)
    DevBuf local_perm, local_bs, local_bv;
    bool sort_here = false;
    int bnnz_local = 0;
    if (!pre) {
        // the numeric-only call may be handed a B that is not the one a symbolic call checked: the column map and the window kernels index by column id, so the
        // ids are range-checked here too (the one-shot call checked them in its symbolic phase) — and B's rows are sorted here if they are not
        G4S_TRY(read_last(brpt, K, &bnnz_local, s));
        G4S_TRY(check_b(brpt, bcol, K, bnnz_local, N, s, &sort_here));
        if (sort_here) {
            G4S_TRY(local_perm.alloc(sizeof(int) * (size_t)bnnz_local)); G4S_TRY(local_bs.alloc(sizeof(int) * (size_t)bnnz_local)); G4S_TRY(local_bv.alloc(sizeof(double) * (size_t)bnnz_local));
            G4S_TRY(sort_b_rows(K, N, bnnz_local, brpt, bcol, local_perm.as<int>(), s));
            G4S_TRY(gather_b(bnnz_local, local_perm.as<int>(), bcol, bval, local_bs.as<int>(), local_bv.as<double>(), s));
            bcol = local_bs.as<int>(); bval = local_bv.as<double>();
        }
    }
    if (pre && pre->b_unsorted) {
        G4S_TRY(local_bv.alloc(sizeof(double) * (size_t)std::max<long long>(pre->bnnz, 1)));
        G4S_TRY(gather_b((int)pre->bnnz, pre->b_perm.as<int>(), nullptr, bval, nullptr, local_bv.as<double>(), s));
        bcol = pre->b_cols_sorted.as<int>(); bval = local_bv.as<double>();
    }
    // lanes per A-entry are sized from the row's average B-row length; the exact nz of the output row (known here) stands in for
    // the flop count of the symbolic phase (they differ by the row's compression ratio), which saves a pass over A
    DevBuf row_size;
    DevBuf &row_flop = row_size;
    G4S_TRY(row_size.alloc(sizeof(long long) * (size_t)M));
    const bool rank = pre && pre->rank && pre->d_cut_off && !getenv("G4S_SPGEMM_NO_RANK");   // the rows with cuts take the rank kernel (spgemm_rank.hpp)
    hipLaunchKernelGGL(nz_to_ll_kernel, dim3((M + 255) / 256), dim3(256), 0, s, M, crpt, rank ? pre->d_cut_off : (const long long *)nullptr, row_size.as<long long>());
    RowClasses local_rc;
    const bool reuse_rc = pre && pre->sym_rc_valid;                // (see PreSorted::sym_rc)
    // With the rank path on, what is left for the window kernels past 4 096 outputs is the rows of at most 8 192 products (and symbolic hub rows): those of up to
    // 8 192 outputs join the mid-size launch (four 2 048-output chunks at most) instead of being a launch — a row sort, unit lists, a persistent kernel — of their own.
    ClassLimits num_limits = kNumLimits;
    if (rank) num_limits.lim[4] = 8192;
    if (!reuse_rc) G4S_TRY(classify_rows(M, row_size.as<long long>(), num_limits, 0, local_rc, s));
    const RowClasses &rc = reuse_rc ? pre->sym_rc : local_rc;
    const bool num_only_short = rc.count[CLS_MEDIUM] + rc.count[CLS_LARGE] + rc.count[CLS_M2] + rc.count[CLS_M3] + rc.count[CLS_HUB] + rc.count[CLS_RANK] == 0;
    if (getenv("G4S_DEBUG"))
        fprintf(stderr, "g4s numeric classes: empty %d <=32 %d <=512 %d <=1024 %d <=2048 %d <=4096 %d <=1M %d hub %d rank %d\n", rc.count[CLS_EMPTY], rc.count[CLS_TINY],
                rc.count[CLS_SMALL], rc.count[CLS_MEDIUM], rc.count[CLS_LARGE], rc.count[CLS_M2], rc.count[CLS_M3], rc.count[CLS_HUB], rc.count[CLS_RANK]);

    dbg.mark("classes");
    ColumnMap local_map;                                           // see spgemm_symbolic_impl; the one-shot call hands its map over
    if (!pre && !num_only_short) {
        G4S_TRY(build_column_map(N, bnnz_local, bcol, local_map, s));
    }
    const ColumnMap &cmap = pre ? pre->cmap : local_map;
    const int *wcol = cmap.cols(bcol), *winv = cmap.inverse();
    const int N2 = cmap.width(N);
    DevBuf wsplit_buf;
    const int *wsplit = pre && pre->has_wsplit ? pre->wsplit : nullptr;
    if (!(pre && pre->has_wsplit) && !num_only_short) G4S_TRY(build_window_splits(K, N2, brpt, wcol, wsplit_buf, &wsplit, s));
    std::vector<std::unique_ptr<DevBuf>> ct_keep;                  // transients of the launches below: live until the end of this call
    // ---- the rows with cuts (spgemm_rank.hpp), longest first: chunk lists from the cuts, exact splits and unit lists per chunk, then the rank kernel. In two stages:
    // stage 1 (row order, chunk counts, their scans, B's records) is enqueued before the short rows and the mid-size classes, so that its three totals have long
    // arrived when stage 2 (chunk lists, splits, unit lists, the kernel) is enqueued behind them — the host never waits for the rank launch's counts.
    const int rank_n = rc.count[CLS_RANK];
    SortedRows rank_sr;
    auto mk = [&]() { ct_keep.push_back(std::make_unique<DevBuf>()); return ct_keep.back().get(); };
    DevBuf *tasks = mk(), *toff = mk(), *items = mk(), *ioff = mk(), *nch = mk(), *choff = mk(), *chunks = mk(), *ctoff = mk(), *ctb = mk(), *ucnt = mk(), *uoff = mk(), *ud = mk(), *bpack = mk();
    long long rank_totals[2] = {0, 0};
    int rank_nchunks = 0;
    hipEvent_t rank_totals_ready = nullptr;
    auto rank_stage1 = [&]() -> int {
        const int n = rank_n;
        G4S_TRY(rank_sr.build(n, rc.list(CLS_RANK), nullptr, crpt, N, s));
        const int *rows = rank_sr.rows.as<int>();
        G4S_TRY(tasks->alloc(sizeof(long long) * ((size_t)n + 1))); G4S_TRY(toff->alloc(sizeof(long long) * ((size_t)n + 1)));
        G4S_TRY(items->alloc(sizeof(long long) * ((size_t)n + 1))); G4S_TRY(ioff->alloc(sizeof(long long) * ((size_t)n + 1)));
        G4S_TRY(nch->alloc(sizeof(int) * ((size_t)n + 1))); G4S_TRY(choff->alloc(sizeof(int) * ((size_t)n + 1)));
        G4S_TRY(ctoff->alloc(sizeof(long long) * ((size_t)n + 1)));
        hipLaunchKernelGGL(rank_chunks_kernel<false>, dim3((n + 256) / 256), dim3(256), 0, s, n, rows, arpt, crpt, pre->d_cut_off, pre->d_cuts, pre->nseg, tasks->as<long long>(), items->as<long long>(),
                           nch->as<int>(), (const int *)nullptr, (RankChunk *)nullptr);
        G4S_TRY(g4s::prims::exclusive_scan(tasks->as<long long>(), toff->as<long long>(), (long long)n + 1, s));
        G4S_TRY(g4s::prims::exclusive_scan(items->as<long long>(), ioff->as<long long>(), (long long)n + 1, s));
        G4S_TRY(g4s::prims::exclusive_scan(nch->as<int>(), choff->as<int>(), (long long)n + 1, s));
        G4S_HIP_TRY(g4s::read_small(&rank_totals[0], toff->as<long long>() + n, sizeof(long long), s));
        G4S_HIP_TRY(g4s::read_small(&rank_totals[1], ioff->as<long long>() + n, sizeof(long long), s));
        G4S_HIP_TRY(g4s::read_small(&rank_nchunks, choff->as<int>() + n, sizeof(int), s));
        {   // an event behind the three copies: stage 2 waits for IT, not for the stream — the mid-size kernel enqueued in between is still running then
            thread_local hipEvent_t ev = nullptr;
            if (!ev) G4S_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            G4S_HIP_TRY(hipEventRecord(ev, s));
            rank_totals_ready = ev;
        }
        G4S_REQUIRE(pre->bnnz >= 0, "carried state without nnz(B)");
        G4S_TRY(bpack->alloc(sizeof(BPack) * (size_t)std::max<long long>(pre->bnnz, 1)));
        if (pre->bnnz > 0)
            hipLaunchKernelGGL(pack_b_kernel, dim3((unsigned)((pre->bnnz + 255) / 256)), dim3(256), 0, s, pre->bnnz, wcol, bcol, bval, bpack->as<BPack>());
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    };
    auto rank_stage2 = [&]() -> int {
        constexpr int T = kRankT;
        const int n = rank_n, nseg = pre->nseg;
        const int *rows = rank_sr.rows.as<int>();
        G4S_HIP_TRY(g4s::reads_sync_event(rank_totals_ready, s));        // stage 1's totals: there long ago (the mid-size launch in between has synchronised the stream)
        const int nchunks = rank_nchunks;
        const long long ntask = rank_totals[0], nitem = rank_totals[1], nct = nitem - ntask;
        if (ntask <= 0 || nitem <= 0 || nitem > (1ll << 28) || nchunks <= 0)
            return g4s::set_error(G4S_ERR_INVALID, "SpGEMM: the chunk lists of the carried symbolic state do not fit this product (%lld tasks, %lld items, %d chunks)", ntask, nitem, nchunks);
        const long long ubound = (pre->flop >= 0 ? pre->flop : 0) / 64 + nitem + 1;
        G4S_REQUIRE(pre->flop >= 0 && ubound <= (1ll << 28), "unit list of the rank launch past its cap");
        G4S_TRY(chunks->alloc(sizeof(RankChunk) * (size_t)nchunks));
        G4S_TRY(ctb->alloc(sizeof(int) * (size_t)std::max<long long>(nct, 1)));
        G4S_TRY(ucnt->alloc(sizeof(int) * ((size_t)nitem + 1))); G4S_TRY(uoff->alloc(sizeof(int) * ((size_t)nitem + 1)));
        G4S_TRY(ud->alloc(sizeof(UnitDesc) * (size_t)ubound));
        hipLaunchKernelGGL(rank_chunks_kernel<true>, dim3((n + 256) / 256), dim3(256), 0, s, n, rows, arpt, crpt, pre->d_cut_off, pre->d_cuts, nseg, (long long *)nullptr, (long long *)nullptr,
                           (int *)nullptr, (const int *)choff->as<int>(), chunks->as<RankChunk>());
        hipLaunchKernelGGL(diff_ll_kernel, dim3((n + 256) / 256), dim3(256), 0, s, n + 1, ioff->as<long long>(), toff->as<long long>(), ctoff->as<long long>());
        if (nct > 0)
            hipLaunchKernelGGL(chunk_splits_kernel, dim3((unsigned)((nct + 255) / 256)), dim3(256), 0, s, nct, n, rows, K, N2, wsplit, arpt, acol, brpt, wcol, crpt, (const long long *)nullptr, (const int *)nullptr,
                               (const int *)nullptr, kRankChunk, ctoff->as<long long>(), ctb->as<int>(), (const int *)choff->as<int>(), (const RankChunk *)chunks->as<RankChunk>());
        const unsigned tgrid = (unsigned)((ntask + 255) / 256);
        hipLaunchKernelGGL(unit_task_kernel<false>, dim3(tgrid), dim3(256), 0, s, ntask, n, rows, toff->as<long long>(), ioff->as<long long>(), arpt, acol, aval, brpt, wcol, crpt,
                           (const long long *)nullptr, (const int *)nullptr, (const int *)nullptr, kRankChunk, ctb->as<int>(), ucnt->as<int>(), (const int *)nullptr, (UnitDesc *)nullptr, 1, (const int *)choff->as<int>());
        G4S_TRY(g4s::prims::exclusive_scan(ucnt->as<int>(), uoff->as<int>(), nitem + 1, s));
        hipLaunchKernelGGL(unit_task_kernel<true>, dim3(tgrid), dim3(256), 0, s, ntask, n, rows, toff->as<long long>(), ioff->as<long long>(), arpt, acol, aval, brpt, wcol, crpt,
                           (const long long *)nullptr, (const int *)nullptr, (const int *)nullptr, kRankChunk, ctb->as<int>(), (int *)nullptr, (const int *)uoff->as<int>(), ud->as<UnitDesc>(), 1, (const int *)choff->as<int>());
        constexpr size_t lds = sizeof(int) * (3 * (size_t)kRankChunk + 2 * (size_t)kRankWords + 64);
        // the flat chunk list (spgemm_numeric_rank2_kernel): one self-contained item per chunk, dealt round-robin
        DevBuf *ritems = mk();
        G4S_TRY(ritems->alloc(sizeof(RankItem) * (size_t)nchunks));
        hipLaunchKernelGGL(rank_items_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, rows, arpt, crpt, (const long long *)ioff->as<long long>(), (const int *)uoff->as<int>(), (const int *)choff->as<int>(),
                           (const RankChunk *)chunks->as<RankChunk>(), ritems->as<RankItem>());
        // split by size (spgemm_rank.hpp): the chunks of at most kRankSmallCap outputs take the 512-thread shape, two workgroups per CU
        const bool two_shapes = !getenv("G4S_SPGEMM_RANK_ONE_SHAPE");
        auto *sflag = mk(), *spos = mk(), *rbig = mk(), *rsmall = mk(), *rcounts = mk();
        const RankItem *big_items = ritems->as<RankItem>();
        const int *n_big = nullptr;
        if (two_shapes) {
            G4S_TRY(sflag->alloc(sizeof(int) * ((size_t)nchunks + 1))); G4S_TRY(spos->alloc(sizeof(int) * ((size_t)nchunks + 1)));
            G4S_TRY(rbig->alloc(sizeof(RankItem) * (size_t)nchunks)); G4S_TRY(rsmall->alloc(sizeof(RankItem) * (size_t)nchunks)); G4S_TRY(rcounts->alloc(sizeof(int) * 2));
            hipLaunchKernelGGL(rank_small_flags_kernel, dim3((nchunks + 256) / 256), dim3(256), 0, s, nchunks, (const RankItem *)ritems->as<RankItem>(), sflag->as<int>());
            G4S_TRY(g4s::prims::exclusive_scan(sflag->as<int>(), spos->as<int>(), (long long)nchunks + 1, s));
            hipLaunchKernelGGL(rank_split_items_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, nchunks, (const RankItem *)ritems->as<RankItem>(), (const int *)spos->as<int>(), rbig->as<RankItem>(),
                               rsmall->as<RankItem>(), rcounts->as<int>());
            big_items = rbig->as<RankItem>(); n_big = rcounts->as<int>();
        }
        auto k = spgemm_numeric_rank2_kernel<T, kRankChunk, G4S_SPGEMM_RANK_UPR>;
        G4S_TRY(allow_lds(k, lds));
        hipLaunchKernelGGL(k, dim3(big_grid(nchunks, 1)), dim3(T), lds, s, nchunks, n_big, big_items, wcol, (const BPack *)bpack->as<BPack>(), (const UnitDesc *)ud->as<UnitDesc>(), ccol, cval);
        if (two_shapes) {
            auto ks = spgemm_numeric_rank2_kernel<kRankSmallT, kRankSmallCap, 8>;
            constexpr size_t lds_s = rank_lds_bytes(kRankSmallCap);
            G4S_TRY(allow_lds(ks, lds_s));
            hipLaunchKernelGGL(ks, dim3(big_grid(nchunks, 2)), dim3(kRankSmallT), lds_s, s, nchunks, (const int *)(rcounts->as<int>() + 1), (const RankItem *)rsmall->as<RankItem>(), wcol,
                               (const BPack *)bpack->as<BPack>(), (const UnitDesc *)ud->as<UnitDesc>(), ccol, cval);
        }
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    };
    if (rank_n) G4S_TRY(rank_stage1());
    // rows of at most 512 outputs: one wavefront merges a row (spgemm_small_wave_kernel); a row of more than 512 PRODUCTS (the classes are cut by output length) or
    // more than 64 A-entries lands on a list that the table kernel takes, its count read on the device
    DevBuf small_ovf, small_ovf_n;
    if (int n = rc.count[CLS_TINY] + rc.count[CLS_SMALL]; n && !getenv("G4S_SPGEMM_NO_WAVE_ROWS")) {
        G4S_TRY(small_ovf.alloc(sizeof(int) * (size_t)n));
        G4S_TRY(small_ovf_n.alloc(sizeof(int)));
        G4S_HIP_TRY(hipMemsetAsync(small_ovf_n.p, 0, sizeof(int), s));
        // (the lists of a carried symbolic classification are cut by PRODUCTS: its tiny class fits the 64-product form; the numeric classes are cut by outputs)
        const int nt = reuse_rc ? rc.count[CLS_TINY] : 0;
        if (nt) {
            auto k = spgemm_small_wave_kernel<true, kTinyCap>;
            constexpr size_t lds = sizeof(int) * 4 * small_wave_ints<true, kTinyCap>();
            hipLaunchKernelGGL(k, dim3((nt + 3) / 4), dim3(256), lds, s, rc.list(CLS_TINY), nt, arpt, acol, aval, brpt, bcol, bval, (int *)nullptr, crpt, ccol,
                               cval, small_ovf.as<int>(), small_ovf_n.as<int>());
        }
        if (n - nt) {
            auto k = spgemm_small_wave_kernel<true>;
            hipLaunchKernelGGL(k, dim3((n - nt + 3) / 4), dim3(256), sizeof(int) * 4 * small_wave_ints<true>(), s, rc.list(CLS_TINY) + nt, n - nt, arpt, acol, aval, brpt, bcol, bval, (int *)nullptr, crpt,
                               ccol, cval, small_ovf.as<int>(), small_ovf_n.as<int>());
        }
        auto k2 = spgemm_numeric_lds_kernel<256, 256, 1024>;
        hipLaunchKernelGGL(k2, dim3(std::min(n, 256)), dim3(256), num_lds_bytes(1024), s, small_ovf.as<int>(), 0, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval,
                           (const int *)small_ovf_n.as<int>());
    } else {
    if (int n = rc.count[CLS_TINY]) {
        auto k = spgemm_numeric_lds_kernel<256, 64, 64>;
        hipLaunchKernelGGL(k, dim3((n + 3) / 4), dim3(256), 4 * 64 * 12, s, rc.list(CLS_TINY), n, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval, (const int *)nullptr);
    }
    if (int n = rc.count[CLS_SMALL]) {
        auto k = spgemm_numeric_lds_kernel<256, 256, 1024>;
        hipLaunchKernelGGL(k, dim3(n), dim3(256), num_lds_bytes(1024), s, rc.list(CLS_SMALL), n, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval, (const int *)nullptr);
    }
    }
    // Rows past 1 K entries: bitmap windows + bucketed slots beat table + in-LDS bitonic sort while the column range is <= 4 windows.
    const bool xn_wide = N2 <= window_max_n(), xn_large = xn_wide && !(mid_tables() & 2), xn_m2 = xn_wide && !(mid_tables() & 4);
    // exact chunk splits (chunk_splits_kernel) for the rows of one launch that hold more than one value chunk: one-shot call only (the sorted columns must
    // exist before the numeric kernel runs); the buffers live until the end of this call
    const bool use_units = !getenv("G4S_SPGEMM_NO_UNITS");
    // Unit lists of one launch, by (row, A-entry) tasks (unit_task_kernel): splits, counts and descriptors in two passes over the tasks, ONE count read back. Never a
    // reason to fail: without them (a table past its cap, no memory) the kernel walks its rows with the per-chunk entry pass and the window pieces (no exact splits).
    struct UnitLists { const long long *item_off = nullptr; const int *uoff = nullptr; const UnitDesc *U = nullptr; };
    const long long class_flop_bound = pre ? pre->flop : -1;       // units <= flop / 64 + items; the one-shot call knows the product's flop (−1: unknown → the count is read back)
    auto unit_lists_by_tasks = [&](int threads, const int *rows, int n, int nz_lo, int nz_hi, UnitLists *out) -> int {
        *out = UnitLists{};
        const int chunk = 8 * threads;
        auto tasks = std::make_unique<DevBuf>(), toff = std::make_unique<DevBuf>(), items = std::make_unique<DevBuf>(), ioff = std::make_unique<DevBuf>(), ctb = std::make_unique<DevBuf>(),
             ucnt = std::make_unique<DevBuf>(), uoff = std::make_unique<DevBuf>(), ud = std::make_unique<DevBuf>();
        G4S_TRY(tasks->alloc(sizeof(long long) * ((size_t)n + 1))); G4S_TRY(toff->alloc(sizeof(long long) * ((size_t)n + 1)));
        G4S_TRY(items->alloc(sizeof(long long) * ((size_t)n + 1))); G4S_TRY(ioff->alloc(sizeof(long long) * ((size_t)n + 1)));
        hipLaunchKernelGGL(unit_rows_kernel, dim3((n + 256) / 256), dim3(256), 0, s, n, rows, arpt, crpt, chunk, nz_lo, nz_hi, tasks->as<long long>(), items->as<long long>());
        G4S_TRY(g4s::prims::exclusive_scan(tasks->as<long long>(), toff->as<long long>(), (long long)n + 1, s));
        G4S_TRY(g4s::prims::exclusive_scan(items->as<long long>(), ioff->as<long long>(), (long long)n + 1, s));
        long long totals[2] = {0, 0};
        G4S_HIP_TRY(g4s::read_small(&totals[0], toff->as<long long>() + n, sizeof(long long), s));
        G4S_HIP_TRY(g4s::read_small(&totals[1], ioff->as<long long>() + n, sizeof(long long), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        const long long ntask = totals[0], nitem = totals[1];
        if (ntask <= 0 || nitem <= 0 || nitem > (1ll << 28)) return G4S_OK;
        if (ctb->alloc(sizeof(int) * (size_t)std::max<long long>(nitem - ntask, 1)) != G4S_OK || ucnt->alloc(sizeof(int) * ((size_t)nitem + 1)) != G4S_OK ||
            uoff->alloc(sizeof(int) * ((size_t)nitem + 1)) != G4S_OK) { (void)hipGetLastError(); return G4S_OK; }
        const unsigned grid = (unsigned)((ntask + 255) / 256);
        // the splits: one thread per (entry, interior chunk boundary) — chunk_splits_kernel, its offsets = item_off − task_off (entries · (chunks − 1) per row)
        const long long nct = nitem - ntask;
        auto ctoff = std::make_unique<DevBuf>();
        G4S_TRY(ctoff->alloc(sizeof(long long) * ((size_t)n + 1)));
        hipLaunchKernelGGL(diff_ll_kernel, dim3((n + 256) / 256), dim3(256), 0, s, n + 1, ioff->as<long long>(), toff->as<long long>(), ctoff->as<long long>());
        if (nct > 0)
            hipLaunchKernelGGL(chunk_splits_kernel, dim3((unsigned)((nct + 255) / 256)), dim3(256), 0, s, nct, n, rows, K, N2, wsplit, arpt, acol, brpt, wcol, crpt, pre_off, pre_cols, ccol, chunk,
                               ctoff->as<long long>(), ctb->as<int>());
        hipLaunchKernelGGL(unit_task_kernel<false>, dim3(grid), dim3(256), 0, s, ntask, n, rows, toff->as<long long>(), ioff->as<long long>(), arpt, acol, aval, brpt, wcol, crpt,
                           pre_off, pre_cols, ccol, chunk, ctb->as<int>(), ucnt->as<int>(), (const int *)nullptr, (UnitDesc *)nullptr, 1);
        G4S_TRY(g4s::prims::exclusive_scan(ucnt->as<int>(), uoff->as<int>(), nitem + 1, s));
        long long ubound = class_flop_bound >= 0 ? class_flop_bound / 64 + nitem : -1;
        if (ubound < 0 || ubound > (1ll << 27)) {                   // no bound from the caller (or a loose one): read the count
            int total_units = 0;
            G4S_HIP_TRY(g4s::read_small(&total_units, uoff->as<int>() + nitem, sizeof(int), s));
            G4S_HIP_TRY(g4s::reads_sync(s));
            if (total_units <= 0 || total_units > (1 << 27)) return G4S_OK;      // (a sum past 2^31 shows up as a negative total)
            ubound = total_units;
        }
        if (ud->alloc(sizeof(UnitDesc) * (size_t)ubound) != G4S_OK) { (void)hipGetLastError(); return G4S_OK; }
        hipLaunchKernelGGL(unit_task_kernel<true>, dim3(grid), dim3(256), 0, s, ntask, n, rows, toff->as<long long>(), ioff->as<long long>(), arpt, acol, aval, brpt, wcol, crpt,
                           pre_off, pre_cols, ccol, chunk, ctb->as<int>(), (int *)nullptr, uoff->as<int>(), ud->as<UnitDesc>(), 1);
        G4S_HIP_TRY(hipGetLastError());
        out->item_off = ioff->as<long long>(); out->uoff = uoff->as<int>(); out->U = ud->as<UnitDesc>();
        for (auto *b : {&tasks, &toff, &items, &ioff, &ctb, &ucnt, &uoff, &ud, &ctoff}) ct_keep.push_back(std::move(*b));
        return G4S_OK;
    };
    // The window kernels need every row's sorted distinct columns (window ids) before they start. The one-shot call carries them over from its symbolic
    // phase; whatever is missing — every row in the two-call form, hub rows and overflowed optimistic tables in the one-shot form — is marked and emitted
    // into ccol at the row's own offset by the symbolic window kernel in its emit-only mode, in front of the launch that needs it (round 4: this used to be
    // phase 1 inside the numeric kernel itself).
    const bool carried_complete = pre && pre->complete;
    auto emit_pass = [&](auto shape, const int *rows, int n, int nz_lo, int nz_hi) -> int {
        constexpr int T = decltype(shape)::value;
        auto k = spgemm_symbolic_window_kernel<T>;
        const size_t lds = big_lds_bytes<T>();
        G4S_TRY(allow_lds(k, lds));
        hipLaunchKernelGGL(k, dim3(big_grid(n, BigCfg<T>::kPerCu)), dim3(T), lds, s, rows, n, (int *)nullptr, N2, K, wsplit, arpt, acol, brpt, wcol, (const long long *)nullptr, (int *)nullptr,
                           pre_off, ccol, crpt, nz_lo, nz_hi);
        return G4S_OK;
    };
    auto big_t = [&](auto shape, const int *rows, int n, int nz_lo, int nz_hi, int *next_row) -> int {
        constexpr int T = decltype(shape)::value;
        if (!n) return G4S_OK;
        const size_t lds = big_lds_bytes<T>();   // value chunk (fp64 sums, columns, bucket index) + scan scratch + the flat lists
        UnitLists ul;
        if (!carried_complete) G4S_TRY(emit_pass(shape, rows, n, nz_lo, nz_hi));
        if (use_units) G4S_TRY(unit_lists_by_tasks(T, rows, n, nz_lo, nz_hi, &ul));
        const dim3 grid(big_grid(n, BigCfg<T>::kPerCu));
        auto rmeta = std::make_unique<DevBuf>();
        G4S_TRY(rmeta->alloc(sizeof(NumRowMeta) * (size_t)n));
        hipLaunchKernelGGL(num_row_meta_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, rows, arpt, crpt, pre_off, ul.U ? ul.item_off : (const long long *)nullptr, ul.uoff, rmeta->as<NumRowMeta>());
        if (ul.U) {
            auto k = spgemm_numeric_big_kernel<T, true>;
            G4S_TRY(allow_lds(k, lds));
            hipLaunchKernelGGL(k, grid, dim3(T), lds, s, rows, n, nz_lo, nz_hi, next_row, N2, K, wsplit, arpt, acol, aval, brpt, wcol, winv, bval, row_flop.as<long long>(), crpt, ccol, cval, pre_off, pre_cols,
                               ul.item_off, ul.uoff, ul.U, rmeta->as<NumRowMeta>());
        } else {
            auto k = spgemm_numeric_big_kernel<T, false>;
            G4S_TRY(allow_lds(k, lds));
            hipLaunchKernelGGL(k, grid, dim3(T), lds, s, rows, n, nz_lo, nz_hi, next_row, N2, K, wsplit, arpt, acol, aval, brpt, wcol, winv, bval, row_flop.as<long long>(), crpt, ccol, cval, pre_off, pre_cols,
                               (const long long *)nullptr, (const int *)nullptr, (const UnitDesc *)nullptr, rmeta->as<NumRowMeta>());
        }
        ct_keep.push_back(std::move(rmeta));
        return G4S_OK;
    };
    auto big = [&](int threads, const int *rows, int n, int nz_lo = 0, int nz_hi = INT_MAX, int *next_row = nullptr) -> int {
        if (threads == 256) return big_t(std::integral_constant<int, 256>{}, rows, n, nz_lo, nz_hi, next_row);
        if (threads == 512) return big_t(std::integral_constant<int, 512>{}, rows, n, nz_lo, nz_hi, next_row);
        return big_t(std::integral_constant<int, 1024>{}, rows, n, nz_lo, nz_hi, next_row);
    };
    const int t_large = shape_of("G4S_SPGEMM_T_NUM_LARGE", kShapeNumLarge), t_m2 = shape_of("G4S_SPGEMM_T_NUM_M2", kShapeNumM2),
              t_m3 = shape_of("G4S_SPGEMM_T_NUM_M3", kShapeNumM3);
    const int m3_cut = getenv("G4S_SPGEMM_M3_CUT") ? atoi(getenv("G4S_SPGEMM_M3_CUT")) : kNumM3Cut;
    const char *e_med = getenv("G4S_SPGEMM_T_NUM_MED");
    const int t_med = (mid_tables() & 1) ? 0 : e_med ? (atoi(e_med) ? shape_of("G4S_SPGEMM_T_NUM_MED", 1024) : 0) : kShapeNumMedium;   // 0: the table kernel
    dbg.mark("table-kernels+maps");
    const bool one_mid_launch = t_med && xn_large && xn_m2 && t_med == t_large && t_large == t_m2;
    if (one_mid_launch) {
        // the three mid-size classes (512 < nz <= 4 096) share a shape and their lists are adjacent: one launch, one set of unit lists, one tail instead of three
        G4S_TRY(big(t_med, rc.list(CLS_MEDIUM), rc.count[CLS_MEDIUM] + rc.count[CLS_LARGE] + rc.count[CLS_M2]));
    } else {
    if (t_med && xn_large) { G4S_TRY(big(t_med, rc.list(CLS_MEDIUM), rc.count[CLS_MEDIUM])); }
    else if (int n = rc.count[CLS_MEDIUM]) {
        auto k = spgemm_numeric_lds_kernel<256, 256, 2048>;
        hipLaunchKernelGGL(k, dim3(n), dim3(256), num_lds_bytes(2048), s, rc.list(CLS_MEDIUM), n, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval, (const int *)nullptr);
    }
    if (xn_large) { G4S_TRY(big(t_large, rc.list(CLS_LARGE), rc.count[CLS_LARGE])); }
    else if (int n = rc.count[CLS_LARGE]) {
        auto k = spgemm_numeric_lds_kernel<512, 512, 4096>;
        G4S_TRY(allow_lds(k, num_lds_bytes(4096)));
        hipLaunchKernelGGL(k, dim3(n), dim3(512), num_lds_bytes(4096), s, rc.list(CLS_LARGE), n, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval, (const int *)nullptr);
    }
    if (xn_m2) { G4S_TRY(big(t_m2, rc.list(CLS_M2), rc.count[CLS_M2])); }
    else if (int n = rc.count[CLS_M2]) {
        auto k = spgemm_numeric_lds_kernel<1024, 1024, 8192>;
        G4S_TRY(allow_lds(k, num_lds_bytes(8192)));
        hipLaunchKernelGGL(k, dim3(n), dim3(1024), num_lds_bytes(8192), s, rc.list(CLS_M2), n, arpt, acol, aval, brpt, bcol, bval, row_flop.as<long long>(), crpt, ccol, cval, (const int *)nullptr);
    }
    }
    dbg.mark("mid-launch");
    if (rank_n) G4S_TRY(rank_stage2());                            // (its totals have arrived while the mid-size classes were being enqueued)
    dbg.mark("rank-launch");
    if (int n = rc.count[CLS_M3]) {
        // the class spans 4 K … 128 K entries: its short rows go to the many-workgroups shape, the long ones keep 1 024 threads
        if (t_m3 != 1024 && m3_cut > 0) { G4S_TRY(big(t_m3, rc.list(CLS_M3), n, 0, m3_cut)); G4S_TRY(big(1024, rc.list(CLS_M3), n, m3_cut, INT_MAX)); }
        else if (getenv("G4S_SPGEMM_STATIC_ROWS")) G4S_TRY(big(1024, rc.list(CLS_M3), n));
        else {
            // The class spans 4 K … 1 M outputs per row (a 250-fold range of work): longest rows first, handed out one at a time.
            SortedRows sr;
            // (round 4 tried a clustered order instead — the rows past 64 K outputs longest first, the rest by their first column id, i.e. by the long B row most of
            // their products come from, so that consecutive tickets find it in L2: 14.62 against 14.74 ms for the kernel, nothing for the call. B-row misses are
            // not what this kernel waits for.)
            G4S_TRY(sr.build(n, rc.list(CLS_M3), nullptr, crpt, N, s));   // a row has at most N outputs
            G4S_TRY(big(m3_cut <= 0 ? t_m3 : 1024, sr.rows.as<int>(), n, 0, INT_MAX, sr.counter.as<int>()));   // (G4S_SPGEMM_M3_CUT=0 + G4S_SPGEMM_T_NUM_M3: the whole class in another shape)
            // (the sorted list and the counter are released in stream order — or with the call's arena: no wait here)
        }
    }
    G4S_HIP_TRY(hipGetLastError());
    dbg.mark("big-launch(+sync)");
    std::vector<int> hub, ranges;
    G4S_TRY(fetch_rows_and_ranges(rc.list(CLS_HUB), rc.count[CLS_HUB], arpt, hub, ranges, s));
    G4S_TRY(run_hub_rows(true, hub, ranges, N, arpt, acol, aval, brpt, bcol, bval, nullptr, crpt, ccol, cval, s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    t_idle = true;                                                 // (cleared by the next call on this thread)
    dbg.mark("hub+sync");
    return G4S_OK;
}
} // namespace

G4S_API g4s_status g4s_spgemm_numeric(int32_t M, int32_t K, int32_t N,
                                      const int32_t *arpt, const int32_t *acol, const double *aval,
                                      const int32_t *brpt, const int32_t *bcol, const double *bval,
                                      const int32_t *crpt, int32_t *ccol, double *cval, unsigned flags, void *stream)
{
    std::shared_ptr<PreSorted> pre;                                // the state the symbolic call of THIS product left behind, if any — it stays for further numeric
    {                                                              // calls on the same pattern (new values of A or B, the time-stepping case) until something drops it
        std::lock_guard<std::mutex> lock(g_carried.m);
        if (g_carried.pre && g_carried.M == M && g_carried.K == K && g_carried.N == N && g_carried.arpt == arpt && g_carried.acol == acol && g_carried.brpt == brpt &&
            g_carried.bcol == bcol && g_carried.crpt == crpt)
            pre = g_carried.pre;
    }
    if (pre) {
        // The same pointers do not make it the same product: a caller may have refilled the buffers with another pattern, or brought its own crpt. The
        // state is taken over only while the five index arrays still hash to what the symbolic call saw (one pass over them, ≈ 30 µs on configs[2], one host wait);
        // otherwise this is an ordinary numeric call that works everything out again.
        hipStream_t s = g4s::as_stream(stream);
        unsigned long long *d_hash = nullptr, h_hash = 0;
        G4S_TRY(g4s::scratch_alloc(reinterpret_cast<void **>(&d_hash), sizeof(unsigned long long), s));
        int st = enqueue_pattern_hash(M, K, pre->annz, pre->bnnz, arpt, acol, brpt, bcol, crpt, d_hash, s);
        if (st == G4S_OK && (g4s::read_small(&h_hash, d_hash, sizeof(h_hash), s) != hipSuccess || g4s::reads_sync(s) != hipSuccess))
            st = g4s::set_error(G4S_ERR_HIP, "g4s_spgemm_numeric: reading the pattern hash failed");
        g4s::scratch_free(d_hash, s);
        G4S_TRY(st);
        if (h_hash != pre->key_hash) {
            if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s numeric: the index arrays changed since the symbolic call — its carried state is not used\n");
            pre.reset();
        }
    }
    return spgemm_numeric_impl(M, K, N, arpt, acol, aval, brpt, bcol, bval, crpt, ccol, cval, flags, stream, pre.get());
}

G4S_API g4s_status g4s_spgemm_flop(int32_t M, const int32_t *arpt, const int32_t *acol, const int32_t *brpt,
                                   int64_t *flop, int64_t *row_flop, unsigned flags)
{
    G4S_REQUIRE(M >= 0 && arpt && brpt && flop, "bad argument");
    *flop = 0;
    if (M == 0) return G4S_OK;
    t_stream = nullptr;
    t_idle = false;
    DevBuf rf;
    if (flags & G4S_DEVICE_POINTERS) {
        long long *d_rf = reinterpret_cast<long long *>(row_flop);
        if (!d_rf) { G4S_TRY(rf.alloc(sizeof(long long) * (size_t)M)); d_rf = rf.as<long long>(); }
        return compute_row_flop(M, arpt, acol, brpt, d_rf, flop, nullptr);
    }
    // host pointers: brpt's length is implied by the largest column id of A
    const int annz = arpt[M];
    G4S_REQUIRE(annz >= 0, "arpt[M] is negative");
    int maxc = -1;
    for (int k = 0; k < annz; ++k) { G4S_REQUIRE(acol[k] >= 0, "negative column id in A"); if (acol[k] > maxc) maxc = acol[k]; }
    DevBuf d_arpt, d_acol, d_brpt;
    G4S_TRY(d_arpt.alloc(sizeof(int) * ((size_t)M + 1)));
    G4S_TRY(d_acol.alloc(sizeof(int) * (size_t)annz));
    G4S_TRY(d_brpt.alloc(sizeof(int) * ((size_t)maxc + 2)));
    G4S_TRY(rf.alloc(sizeof(long long) * (size_t)M));
    G4S_HIP_TRY(hipMemcpy(d_arpt.p, arpt, sizeof(int) * ((size_t)M + 1), hipMemcpyHostToDevice));
    if (annz) G4S_HIP_TRY(hipMemcpy(d_acol.p, acol, sizeof(int) * (size_t)annz, hipMemcpyHostToDevice));
    G4S_HIP_TRY(hipMemcpy(d_brpt.p, brpt, sizeof(int) * ((size_t)maxc + 2), hipMemcpyHostToDevice));
    G4S_TRY(compute_row_flop(M, d_arpt.as<int>(), d_acol.as<int>(), d_brpt.as<int>(), rf.as<long long>(), flop, nullptr));
    if (row_flop) G4S_HIP_TRY(hipMemcpy(row_flop, rf.p, sizeof(long long) * (size_t)M, hipMemcpyDeviceToHost));
    return G4S_OK;
}

G4S_API g4s_status g4s_spgemm_csr_i32_f64(const int32_t *arpt, const int32_t *acol, const double *aval,
                                          const int32_t *brpt, const int32_t *bcol, const double *bval,
                                          int32_t **crpt_out, int32_t **ccol_out, double **cval_out,
                                          int32_t M, int32_t K, int32_t N, int64_t *cnnz_out,
                                          g4s_timings *timings, unsigned flags)
{
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    G4S_REQUIRE(crpt_out && ccol_out && cval_out && cnnz_out, "NULL output argument");
    G4S_REQUIRE(M >= 0 && K >= 0 && N >= 0 && arpt && brpt, "bad argument");
    *crpt_out = nullptr; *ccol_out = nullptr; *cval_out = nullptr; *cnnz_out = 0;
    g4s_timings t{};
    drop_carried();                                                // (a two-call product left unfinished holds the column scratch)
    const auto t_total = clk::now();
    ArenaScope arena;                                              // spans both phases: the carried state of the symbolic phase lives in it
    const bool dev = (flags & G4S_DEVICE_POINTERS) != 0;
    t_stream = nullptr;                                            // the one-call form runs on the default stream
    t_idle = false;

    // ---- create: inputs to the device (mkl_sparse_d_create_csr ×2 in the reference's timed path, mkl_mult.h:50-52)
    auto t0 = clk::now();
    DevBuf d_arpt, d_acol, d_aval, d_brpt, d_bcol, d_bval;
    const int *p_arpt = arpt, *p_acol = acol, *p_brpt = brpt, *p_bcol = bcol;
    const double *p_aval = aval, *p_bval = bval;
    if (!dev) {
        const int annz = M ? arpt[M] : 0, bnnz = K ? brpt[K] : 0;
        G4S_REQUIRE(annz >= 0 && bnnz >= 0, "negative nnz");
        auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
            G4S_TRY(b.alloc(bytes));
            if (bytes) G4S_HIP_TRY(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
            return G4S_OK;
        };
        G4S_TRY(up(d_arpt, arpt, sizeof(int) * ((size_t)M + 1)));
        G4S_TRY(up(d_acol, acol, sizeof(int) * (size_t)annz));
        G4S_TRY(up(d_aval, aval, sizeof(double) * (size_t)annz));
        G4S_TRY(up(d_brpt, brpt, sizeof(int) * ((size_t)K + 1)));
        G4S_TRY(up(d_bcol, bcol, sizeof(int) * (size_t)bnnz));
        G4S_TRY(up(d_bval, bval, sizeof(double) * (size_t)bnnz));
        p_arpt = d_arpt.as<int>(); p_acol = d_acol.as<int>(); p_aval = d_aval.as<double>();
        p_brpt = d_brpt.as<int>(); p_bcol = d_bcol.as<int>(); p_bval = d_bval.as<double>();
    }
    t.create = ms_since(t0);

    // ---- spmm: symbolic + numeric (rows come out sorted: the reference's separate `order` stage is folded in)
    t0 = clk::now();
    DevBuf d_crpt;
    G4S_TRY(d_crpt.alloc(sizeof(int) * ((size_t)M + 1), dev));
    int64_t cnnz = 0;
    const auto t_sym = clk::now();
    PreSorted pre;
    G4S_TRY(spgemm_symbolic_impl(M, K, N, p_arpt, p_acol, p_brpt, p_bcol, d_crpt.as<int>(), &cnnz, nullptr, &pre));
    *cnnz_out = cnnz;
    DevBuf d_ccol, d_cval;
    const double ms_sym = ms_since(t_sym);
    const auto t_alloc = clk::now();
    G4S_TRY(d_cval.alloc(sizeof(double) * (size_t)cnnz, dev));
    G4S_TRY(d_ccol.alloc(sizeof(int) * (size_t)cnnz, dev));
    const double ms_alloc = ms_since(t_alloc);
    const auto t_num = clk::now();
    G4S_TRY(spgemm_numeric_impl(M, K, N, p_arpt, p_acol, p_aval, p_brpt, p_bcol, p_bval, d_crpt.as<int>(), d_ccol.as<int>(), d_cval.as<double>(),
                                flags, nullptr, &pre));
    if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s one-call: symbolic %.1f ms, output allocation %.1f ms, numeric %.1f ms\n", ms_sym, ms_alloc, ms_since(t_num));
    t.spmm = ms_since(t0);
    t.convert = 0.0;
    t.order = 0.0;

    // ---- export_csr: hand the result to the caller (mkl_sparse_d_export_csr + memcpy, mkl_mult.h:79-99)
    t0 = clk::now();
    if (dev) {
        *crpt_out = d_crpt.as<int>(); d_crpt.p = nullptr;
        *ccol_out = d_ccol.as<int>(); d_ccol.p = nullptr;
        *cval_out = d_cval.as<double>(); d_cval.p = nullptr;
    } else {
        int *h_rpt = (int *)g4s_malloc(sizeof(int) * ((size_t)M + 1));
        int *h_col = (int *)g4s_malloc(sizeof(int) * (size_t)cnnz);
        double *h_val = (double *)g4s_malloc(sizeof(double) * (size_t)cnnz);
        if (!h_rpt || !h_col || !h_val) { g4s_free(h_rpt); g4s_free(h_col); g4s_free(h_val); return g4s::set_error(G4S_ERR_NOMEM, "host allocation of C failed"); }
        hipError_t e1 = hipMemcpy(h_rpt, d_crpt.p, sizeof(int) * ((size_t)M + 1), hipMemcpyDeviceToHost);
        hipError_t e2 = cnnz ? hipMemcpy(h_col, d_ccol.p, sizeof(int) * (size_t)cnnz, hipMemcpyDeviceToHost) : hipSuccess;
        hipError_t e3 = cnnz ? hipMemcpy(h_val, d_cval.p, sizeof(double) * (size_t)cnnz, hipMemcpyDeviceToHost) : hipSuccess;
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { g4s_free(h_rpt); g4s_free(h_col); g4s_free(h_val); return g4s::set_error(G4S_ERR_HIP, "D2H copy of C failed"); }
        *crpt_out = h_rpt; *ccol_out = h_col; *cval_out = h_val;
    }
    t.export_csr = ms_since(t0);

    t0 = clk::now();
    d_arpt.release(); d_acol.release(); d_aval.release(); d_brpt.release(); d_bcol.release(); d_bval.release();
    d_crpt.release(); d_ccol.release(); d_cval.release();
    t.destroy = ms_since(t0);
    t.total = ms_since(t_total);
    if (timings) *timings = t;
    if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s one-call: create %.2f spmm %.2f export %.2f destroy %.2f total %.2f ms (the carried state of the two phases is released after this line)\n",
                                     t.create, t.spmm, t.export_csr, t.destroy, t.total);
    return G4S_OK;
}

#ifdef G4S_PROFILE_BIG
extern "C" __attribute__((visibility("default"))) int g4s_debug_big_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_prof), sizeof(unsigned long long) * 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_big_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

// called by g4s_shutdown (runtime.cpp)
namespace g4s {
void spgemm_release_cache()
{
    drop_carried();
    std::lock_guard<std::mutex> lock(g_col_cache.m);
    if (g_col_cache.p && !g_col_cache.in_use) { (void)hipFree(g_col_cache.p); g_col_cache.p = nullptr; g_col_cache.bytes = 0; }
}
} // namespace g4s
