// spgemm_rank.hpp — the long-row path of the SpGEMM since round 5 (included by spgemm.hip inside its anonymous namespace; uses its UnitDesc, wave_inclusive_sum, bm_slot).
//
// What it replaces. Until round 4 the rows of more than 8 K products went through (symbolic) bitmap windows that also EMITTED the row's sorted distinct columns
// into an 8 GB scratch — 71 % of the symbolic window kernels was that emit step — and (numeric) value chunks that read the columns back, built a bucket index
// over them, found every product's slot with a search (4.3 halvings per round) and wrote the same columns out again as ccol. With a compression of 1.28 on
// configs[2] the sorted column list is nearly as long as the product list, so carrying it cost more than it saved.
//
// Now (reference arithmetic unchanged: hash_symbolic / hash_numeric, mm/inc/hash_mult.h:65-109,559-608 — distinct columns per row, products added per column):
//   symbolic   the window kernel marks and COUNTS, and writes only the row's CUTS: the compact column of every kRankChunk-th output (count_and_cut_window)
//              and the row's output count at every multiple of kRankWin columns — a few ints per row instead of the whole list;
//   pre-pass   rank_chunks_kernel merges the two kinds of cuts into the row's chunk list: a chunk is at most kRankChunk consecutive outputs whose columns lie in
//              one kRankWin-column segment; the exact B-row splits and the unit lists are built per chunk as before (chunk_splits_kernel, unit_task_kernel);
//   numeric    spgemm_numeric_rank2_kernel, one chunk at a time: a chunk's units are dealt to the 16 waves; a product is one 16-byte record {place, column,
//              value} of B (pack_b_kernel). The first kU units of a wave are requested a step ahead and stay in registers, the rest is streamed in two
//              alternating groups (stream_unit_groups). The products' places are marked in an LDS bitmap of 48-column words, the owner threads turn the top 16
//              bits of every word into the word's exclusive rank, and a product's slot is then rank(word) + popcount(bits below) — one LDS read, no search, no
//              bucket index, no column list; the product adds its value into V[slot] (ds_add_f64) and stores its column into KC[slot] (every product of a slot
//              stores the same id), and the chunk leaves as two coalesced streams. LDS per output: 8 B sum + 4 B column + 7 B of bitmap share = 152 KiB.
// Rows from 4 096 products take this path (G4S_SPGEMM_SYM_MEDIUM, spgemm.hip); DESIGN.md §4.2 has the step table of the round.
#pragma once

constexpr int kRankT = 1024;                               // threads of the rank kernel's workgroup, one per CU (512 threads with half-size chunks, two per CU: 47.5 ms against 30.5)
#ifndef G4S_SPGEMM_RANK_PER
#define G4S_SPGEMM_RANK_PER 8
#endif
#ifndef G4S_SPGEMM_RANK_WPT
#define G4S_SPGEMM_RANK_WPT 7
#endif
constexpr int kRankChunk = G4S_SPGEMM_RANK_PER * kRankT; // outputs per value chunk: 8 192 (6 144 / 5 120 with wider bitmaps: 30.9 / 31.1 ms against 28.0 on the final kernel)
constexpr int kRankWordCols = 48;                          // columns per 64-bit LDS word: bits 0–47 presence, bits 48–63 the word's exclusive rank within the chunk (< 8192)
constexpr int kRankWords = G4S_SPGEMM_RANK_WPT * kRankT; // words of a chunk's bitmap: 7 per thread (7 168 words, 56 KiB; 8 per thread is a 64-byte stride: 32-way bank conflicts)
static_assert(sizeof(int) * (3 * (size_t)kRankChunk + 2 * (size_t)kRankWords + 64) <= 160 * 1024, "the chunk's sums, columns and bitmap must fit one CU's LDS");
constexpr int kRankWin = kRankWords * kRankWordCols;       // 344 064 columns per segment = 336 symbolic threads of 1 024 columns each
constexpr int kRankSegThreads = kRankWin / 1024;
#ifndef G4S_SPGEMM_RANK_CUT
#define G4S_SPGEMM_RANK_CUT (G4S_SPGEMM_RANK_PER * 1024)
#endif
#ifndef G4S_SPGEMM_RANK_SPLIT_MAX
#define G4S_SPGEMM_RANK_SPLIT_MAX 0
#endif
// The symbolic phase cuts a row every kRankCut outputs (= the chunk size: every cut is a chunk boundary). A/B builds (-DG4S_SPGEMM_RANK_CUT=2000
// -DG4S_SPGEMM_RANK_SPLIT_MAX=6000): finer cuts of which the chunk builder takes every kRankMerge-th, and all of them inside a stretch of at most kRankSplitMax
// outputs between two segment boundaries, so that mid-size stretches leave as chunks for the kernel's two-per-CU shape — measured 27.9 against 27.3 ms without
// the splitting and 26.5 with plain 8 192-output cuts (profiles/r05_spgemm_ab.txt): every chunk more is a pass of the (chunk, A-entry) pre-passes over its row.
constexpr int kRankCut = G4S_SPGEMM_RANK_CUT, kRankMerge = kRankChunk / kRankCut, kRankSplitMax = G4S_SPGEMM_RANK_SPLIT_MAX;
static_assert(kRankWin % 1024 == 0 && kRankCut >= 1024 && kRankMerge >= 1 && kRankMerge * kRankCut <= kRankChunk,
              "a symbolic thread (1 024 columns) holds at most one count cut and never straddles a segment; merged cuts fit a chunk");
#ifndef G4S_SPGEMM_RANK_UPR
#define G4S_SPGEMM_RANK_UPR 8                              /* 64-entry units a wave keeps in registers per chunk (16 waves × 8 × 64 = one chunk of products at compression 1); 10: equal, 12 / 14: register spills, 31–36 ms */
#endif
#ifndef G4S_SPGEMM_RANK_AHEAD
#define G4S_SPGEMM_RANK_AHEAD 0                            /* 1: the next chunk's records are requested behind the first barrier into registers of their own — measured 28.18 ms against 28.0 (profiles/r05_spgemm_ab.txt): the wait moves, the time stays */
#endif
#ifndef G4S_SPGEMM_RANK_GM
#define G4S_SPGEMM_RANK_GM 8                               /* units per streamed group in the mark step (one register each) */
#endif
#ifndef G4S_SPGEMM_RANK_GA
#define G4S_SPGEMM_RANK_GA 4                               /* … in the accumulate step (four registers each); 16 / 8 and 12 / 6: equal within the boxes' spread */
#endif
#ifndef G4S_SPGEMM_RANK_ROUNDS
#define G4S_SPGEMM_RANK_ROUNDS 1                           /* rounds of RANK_UPR units per wave whose columns and records are requested ahead and held in registers */
#endif

// A row's cuts live at cuts[cut_off[row] …]: nseg segment starts (the row's output count in front of column s·kRankWin), then the count cuts (the compact column of
// output b·kRankChunk, b = 1, 2, …).
__host__ __device__ __forceinline__ int rank_segments(int N2) { return (N2 + kRankWin - 1) / kRankWin; }

// ---- symbolic: count one marked LDS window and write the cuts that fall into it. Thread t owns the 32 consecutive words [32t, 32t + 32) = 1 024 columns (the
// layout of emit_window_columns). Returns the window's count (the same value in every thread); leaves the bitmap clean. Contains one barrier; the caller puts a
// barrier between this call and the next write to the bitmap or to s_scan.
template <int T>
__device__ __forceinline__ int count_and_cut_window(unsigned *bm, int wi /* window index */, int row_before /* the row's outputs in earlier windows (uniform) */, int nseg,
                                                    int *__restrict__ segstart, int *__restrict__ bcut, int *s_scan, int t)
{
    const int lane = t & 63, wave = t >> 6;
    const int kq = (t >> 1) & 7;                                    // bm_slot's XOR for this thread's block, in 4-word groups
    uint4 *blk = reinterpret_cast<uint4 *>(bm + t * 32);
    uint4 g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) g[q] = blk[q ^ kq];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) cnt += __popc(g[q].x) + __popc(g[q].y) + __popc(g[q].z) + __popc(g[q].w);
#pragma unroll
    for (int q = 0; q < 8; ++q) blk[q] = make_uint4(0u, 0u, 0u, 0u);   // the window leaves the bitmap clean
    const int incl = (int)wave_inclusive_sum((unsigned)cnt);
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    constexpr int kW = T / 64;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned sc = wave_inclusive_sum(lane < kW ? (unsigned)s_scan[lane] : 0u);
    const int total = (int)__builtin_amdgcn_readlane(sc, kW - 1);
    int p = incl - cnt;
    if (wv > 0) p += (int)__builtin_amdgcn_readlane(sc, wv - 1);
    const int g0 = row_before + p;                                  // the row-wide index of this thread's first output
    const int A = wi * T + t;                                       // the thread's 1 024-column block, counted over the whole column range
    if (A % kRankSegThreads == 0 && A / kRankSegThreads < nseg) segstart[A / kRankSegThreads] = g0;
    const int b = (g0 + kRankCut - 1) / kRankCut;                   // the first multiple of the chunk size at or behind g0
    int k = b * kRankCut - g0;
    if (b >= 1 && k < cnt) {                                        // output b·kRankChunk is one of this thread's bits: the k-th (at most one per thread: 1 024 < kRankChunk)
        auto word = [&](int i) { const uint4 &v = g[i >> 2]; return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w; };
        int found = -1;
        unsigned fw = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const unsigned w = word(i);
            const int c = __popc(w);
            if (found < 0) { if (k < c) { found = i; fw = w; } else k -= c; }
        }
        for (int j = 0; j < k; ++j) fw &= fw - 1;                   // k < 32
        bcut[b - 1] = A * 1024 + found * 32 + (__ffs(fw) - 1);
    }
    return total;
}

// ---- the chunk list of a row: the merge of its segment starts and its count cuts by (output index, column); cuts at the same output keep the later column
// (an empty segment, or a count cut that is also a segment's first output). A chunk = outputs [o_lo, o_lo + qn) of the row, columns from cstart on, all inside
// segment seg. Rows without outputs get one empty chunk (the numeric kernel then walks no special case).
struct __attribute__((aligned(16))) RankChunk { int o_lo, qn, cstart, seg; };
template <bool WRITE>
__global__ void rank_chunks_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const int *__restrict__ crpt, const long long *__restrict__ cut_off,
                                   const int *__restrict__ cuts, int nseg, long long *__restrict__ tasks, long long *__restrict__ items, int *__restrict__ nchunks /* !WRITE: out, n + 1 */,
                                   const int *__restrict__ choff /* WRITE */, RankChunk *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { if constexpr (!WRITE) { tasks[n] = 0; items[n] = 0; nchunks[n] = 0; } return; }   // (the scans run over n + 1 entries)
    const int row = rows[i], nz = crpt[row + 1] - crpt[row];
    const int *seg = cuts + cut_off[row], *bc = seg + nseg;
    const int ncut = nz > 0 ? (nz - 1) / kRankCut : 0;
    RankChunk *dst = WRITE ? out + choff[i] : nullptr;
    int s = 0, b = 1, cnt = 0, cur_o = -1, cur_c = 0, stretch_o = 0;   // stretch_o: the output count at the last segment boundary taken
    auto flush = [&](int next_o) {
        if (cur_o >= 0 && next_o > cur_o) {
            if constexpr (WRITE) dst[cnt] = RankChunk{cur_o, next_o - cur_o, cur_c, cur_c / kRankWin};
            ++cnt;
        }
    };
    while (s < nseg || b <= ncut) {
        const int os = s < nseg ? min(seg[s], nz) : INT_MAX, ob = b <= ncut ? b * kRankCut : INT_MAX;
        const int cs = s * kRankWin, cb = b <= ncut ? bc[b - 1] : 0;
        const bool take_seg = os != ob ? os < ob : cs <= cb;
        const int o = take_seg ? os : ob, c = take_seg ? cs : cb;
        if (take_seg) { ++s; stretch_o = o; }
        else {
            ++b;
            // a count cut inside the stretch [stretch_o, next segment boundary or the row's end): every kRankMerge-th one, or every one of a short stretch
            const int stretch_end = s < nseg ? min(seg[s], nz) : nz;
            if ((b - 1) % kRankMerge != 0 && stretch_end - stretch_o > kRankSplitMax) continue;
        }
        if (o != cur_o) { flush(o); cur_o = o; }
        cur_c = c;
    }
    flush(nz);
    if (cnt == 0) { if constexpr (WRITE) dst[0] = RankChunk{0, 0, 0, 0}; cnt = 1; }
    if constexpr (!WRITE) {
        const long long na = arpt[row + 1] - arpt[row];
        tasks[i] = na; items[i] = na * cnt; nchunks[i] = cnt;
    }
}

// B's entries as the rank kernel reads them: the column in both numberings and the value, one global_load_dwordx4 per product instead of two loads, and the
// chunk's ccol leaves LDS as final ids (the store step's gather through the column map was a dependent load in front of every chunk's stores).
struct __attribute__((aligned(16))) BPack { int place, col; double val; };   // place: the compact column c as (c / 48) << 6 | c % 48 — a segment's place word is this minus the segment's base, no division in the kernel
__global__ void pack_b_kernel(long long nnz, const int *__restrict__ c2, const int *__restrict__ col, const double *__restrict__ val, BPack *__restrict__ out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) { const int c = c2[k], w = c / kRankWordCols; out[k] = BPack{(w << 6) | (c - w * kRankWordCols), col[k], val[k]}; }
}

// ---- the rank kernel, on a FLAT chunk list. (Its first form took rows through a ticket counter like the round-4 kernels: ticket → row metadata → unit
// descriptors → loads, three dependent round trips at every row boundary, and a chunk's products requested only once the previous chunk's registers were free:
// +1.1 ms, profiles/r05_spgemm_ab.txt.) Every chunk of every
// row is one self-contained 32-byte item (where its outputs go, its segment, its units), the items stand in the launch's row order (longest rows first) and
// workgroup b takes items b, b + G, b + 2G, … — nothing to draw, so everything about a workgroup's next chunk is known a chunk ahead and is requested then:
//   item(g + 2G) and the unit descriptors of g + G                 at the top of chunk g (a scalar load; one vector load, lane q = unit q's descriptor),
//   the COMPACT columns of g + G (what its mark step needs)          behind chunk g's first barrier,
//   the {column, value} records of g + G (its accumulate step)       behind chunk g's accumulate step, when those registers are free.
// Chunks are bounded pieces of work (at most kRankChunk outputs), so the round-robin deal balances without tickets.
struct __attribute__((aligned(32))) RankItem { int out0, qn, wbase /* the segment's first compact column */, u0, u1, pbase /* … and its first place word */, pad1, pad2; };
__global__ void rank_items_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const int *__restrict__ crpt, const long long *__restrict__ item_off,
                                  const int *__restrict__ uoff, const int *__restrict__ choff, const RankChunk *__restrict__ chunks, RankItem *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i], na = arpt[r + 1] - arpt[r], off = crpt[r], c0 = choff[i], nch = choff[i + 1] - c0;
    const long long io = item_off[i], total_items = item_off[n];
    for (int q = 0; q < nch; ++q) {
        const RankChunk c = chunks[c0 + q];
        out[c0 + q] = RankItem{off + c.o_lo, c.qn, c.seg * kRankWin, uoff[io + (long long)q * na], uoff[min(io + (long long)(q + 1) * na, total_items)], c.seg * (kRankWords << 6), 0, 0};
    }
}

// A wave's units past its register rounds, in two alternating groups of G: while one group's loads are in flight the other is consumed, and nothing is copied
// between the two (a copy of a register that is still being loaded is a wait for the load). On configs[2] a third of all units take this route even with 256 units
// in registers (23 % of the chunks are more crowded than that, 525 units on average), so it is a main path, not an overflow path; the first form — descriptors,
// wait, columns, wait, four units at a time — was 29 % of the kernel (profiles/r05_rank_sections_fine.txt, r05_rank_hist.txt).
template <int G, typename P, typename Issue, typename Use>
__device__ __forceinline__ void stream_unit_groups(int ne /* ≥ 1, uniform */, Issue issue, Use use)
{
    P a[G], b[G];
    issue(a, 0);
    for (int e0 = 0; e0 < ne; e0 += 2 * G) {
        if (e0 + G < ne) issue(b, e0 + G);
        use(a, e0);
        if (e0 + 2 * G < ne) issue(a, e0 + 2 * G);
        if (e0 + G < ne) use(b, e0 + G);
    }
}

// Two shapes (round 5): 1 024 threads and 8 192 outputs, one workgroup per CU; and, for the chunks of at most kRankSmallCap outputs — the sparse tail segments of
// mid-size rows: 44 % of configs[2]'s chunks, 13.7 K cycles each whatever their size, all of it barriers and exposed round trips (profiles/r05_rank_chunk_sizes.txt)
// — 512 threads and 80 KiB of LDS (the same 344 064-column bitmap, sums and columns for 2 000 outputs), TWO workgroups per CU that fill each other's waits.
constexpr int kRankSmallT = 512, kRankSmallCap = 2000;
constexpr size_t rank_lds_bytes(int cap) { return sizeof(int) * (3 * (size_t)cap + 2 * (size_t)kRankWords + 64); }
static_assert(2 * rank_lds_bytes(kRankSmallCap) <= 160 * 1024 && kRankSmallCap % 2 == 0, "two small-chunk workgroups share a CU's LDS; the bitmap behind the columns is 8-byte aligned");
// the chunk items of a launch, split by size: flags → scan → scatter (the order inside either list stays the launch's row order)
__global__ void rank_small_flags_kernel(int n, const RankItem *__restrict__ items, int *__restrict__ flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) flag[i] = i < n && items[i].qn <= kRankSmallCap ? 1 : 0;
}
__global__ void rank_split_items_kernel(int n, const RankItem *__restrict__ items, const int *__restrict__ spos /* n + 1: exclusive scan of the flags */, RankItem *__restrict__ big,
                                        RankItem *__restrict__ small, int *__restrict__ counts /* [2]: big, small */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counts[0] = n - spos[n]; counts[1] = spos[n]; }
    if (i >= n) return;
    const RankItem it = items[i];
    if (it.qn <= kRankSmallCap) small[spos[i]] = it; else big[i - spos[i]] = it;
}

template <int T, int CAP, int KU>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void spgemm_numeric_rank2_kernel(
    int nitems_max, const int *__restrict__ nitems_dev /* not NULL: the list's length (the grid is sized by nitems_max) */, const RankItem *__restrict__ items, const int *__restrict__ bcol2 /* compact column per entry of B */, const BPack *__restrict__ bpack,
    const UnitDesc *__restrict__ U, int *__restrict__ ccol, double *__restrict__ cval)
{
    static_assert(kRankWords % T == 0 && CAP % 2 == 0 && CAP <= kRankChunk, "whole bitmap words per thread; an 8-byte aligned bitmap");
    // kR ROUNDS of kU units per wave live in registers (16 waves × 2 × 8 = 256 units: a typical chunk of configs[2] has 196). With one round the units past 128
    // went through the loops below the slow way — request, wait a full memory latency, use — once in the mark step and once in the accumulate step: 39 % of
    // the kernel (profiles/r05_rank_sections_fine.txt). A product is ONE 16-byte record {compact column, column, value} in four registers: requested when the
    // previous chunk's accumulate step has freed them (its store step and barriers cover the latency), the mark step turns .x into the place word, the
    // accumulate step uses the rest. (Columns and records requested separately, columns a whole chunk ahead: 96 registers of state with the 3-register tuples
    // padded to 4, spills, and a spill reload waits for EVERY load in flight — 38.9 ms against 31.2.)
    constexpr int kGM = G4S_SPGEMM_RANK_GM, kGA = G4S_SPGEMM_RANK_GA;   // group sizes of the streamed units: mark (one register a unit), accumulate (four)
    constexpr int kU = KU, kR = G4S_SPGEMM_RANK_ROUNDS, kWaves = T / 64, kPer = (CAP + T - 1) / T, kWPT = kRankWords / T;
    static_assert(kR >= 1 && kR * kU <= 64, "a round's descriptors are lanes of one register");
    extern __shared__ int lds_i[];                                 // [V: CAP fp64][KC: CAP int][BM: 7168 × 64 bit][ctrl: 64 int]
    double *V = reinterpret_cast<double *>(lds_i);
    int *KC = lds_i + 2 * CAP;
    unsigned long long *BM = reinterpret_cast<unsigned long long *>(lds_i + 3 * CAP);
    unsigned *BM32 = reinterpret_cast<unsigned *>(BM);
    int *ctrl = lds_i + 3 * CAP + 2 * kRankWords;
    const int nitems = nitems_dev ? min(*nitems_dev, nitems_max) : nitems_max;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x;
    BIG_PROF_DECL_RANK;
    auto item_at = [&](int g) { return items[min(g, nitems - 1)]; };   // uniform: one s_load_dwordx8 (an index past the end reads the last item: never used)
    // The chunk's units are DEALT to the waves (unit i → wave i % 16, the wave's k-th unit is k·16 + wave): in blocks of kU a chunk of 195 units gave eight waves
    // 16 units and seven waves 8, and every step between two barriers lasts as long as its busiest wave.
    auto desc_load = [&](const RankItem &it) {                    // lane k = r·kU + q: the descriptor of the wave's unit q of round r
        int4 d = make_int4(0, 1, 0, 0);
        const int nu = it.u1 - it.u0;
        if (lane < kR * kU) d = reinterpret_cast<const int4 *>(U)[it.u0 + min(lane * kWaves + wave, max(nu, 1) - 1)];
        return d;
    };
    auto place_of = [&](int col, int wbase, bool ok) {
        const unsigned rel = (unsigned)(col - wbase);
        const unsigned w = __umulhi(rel, 0xAAAAAAABu) >> 5;        // rel / 48
        return (ok && rel < (unsigned)kRankWin) ? (int)((w << 6) | (rel - w * 48u)) : -1;
    };
    auto place_rec = [&](int place, int pbase, bool ok) {         // the same from a record's place field
        const unsigned rel = (unsigned)(place - pbase);
        return (ok && rel < (unsigned)(kRankWords << 6)) ? (int)rel : -1;
    };
    auto mark = [&](int wr) { if (wr >= 0) atomicOr(&BM32[wr >> 5], 1u << (wr & 31)); };
    auto accumulate = [&](int wr, int col, double prod) {
        if (wr < 0) return;
        const unsigned long long w = BM[wr >> 6];
        const int slot = min((int)(w >> 48) + (int)__popcll(w & ((1ull << (wr & 63)) - 1ull)), (int)CAP - 1);   // (the clamp: stay inside the chunk whatever the arrays hold; all-int, or min() goes through fp64)
        atomicAdd(&V[slot], prod);
        KC[slot] = col;
    };
#pragma unroll
    for (int u = 0; u < kPer; ++u) if (t + u * T < CAP) { V[t + u * T] = 0.0; KC[t + u * T] = 0; }
#pragma unroll
    for (int j = 0; j < kWPT; ++j) BM[t + j * T] = 0ull;
    int g = blockIdx.x;
    if (g >= nitems) return;                                       // uniform
    // prologue: the first chunk's data the slow way, the second chunk's item
    RankItem cur = item_at(g), nxt = item_at(g + G);
    // The wave's kU descriptors of a round stay in ONE int4 per lane (lane q = unit q) and are read out with v_readlane where they are needed: as SGPR arrays
    // (position, length, value × current and next chunk) they alone took more scalar registers than a wave has.
    int4 dc = desc_load(cur);
    auto d_pos = [&](const int4 &d, int r, int q) { return __builtin_amdgcn_readlane(d.x, r * kU + q); };
    auto units_of_wave = [&](int nu) { return __builtin_amdgcn_readfirstlane((nu - wave + kWaves - 1) / kWaves); };   // (nu ≥ 0 > wave - 16: never negative)
    auto units_in = [&](int r, int nu) { return max(0, min(kU, units_of_wave(nu) - r * kU)); };   // the wave's units in round r, a scalar
    auto d_len = [&](const int4 &d, int r, int q, int nu) { return q < units_in(r, nu) ? __builtin_amdgcn_readlane(d.y, r * kU + q) : 0; };   // (a unit past the chunk's last: no lane is valid)
    auto d_val = [&](const int4 &d, int r, int q) { return __longlong_as_double(((long long)__builtin_amdgcn_readlane(d.w, r * kU + q) << 32) | (unsigned)__builtin_amdgcn_readlane(d.z, r * kU + q)); };
    auto entry_of = [&](const int4 &d, int r, int q, int nu) { return d_pos(d, r, q) + min(lane, max(d_len(d, r, q, nu), 1) - 1); };
    // … and the wave's units past the register rounds (k = kR·kU + e), a batch is 64 of them
    auto extra_unit = [&](int e) { return (kR * kU + e) * kWaves + wave; };
    auto extras_in = [&](int eb, int nu) { return max(0, min(64, units_of_wave(nu) - kR * kU - eb)); };   // uniform
    auto extras_load = [&](const RankItem &it, int eb) { return reinterpret_cast<const int4 *>(U)[it.u0 + min(extra_unit(eb + lane), max(it.u1 - it.u0, 1) - 1)]; };
    int4 dx = extras_load(cur, 0);
    int4 rec[kR][kU];                                              // the chunk's products in the register rounds
    auto load_recs = [&](int4 (&into)[kR][kU], const int4 &d, int nu) {
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int q = 0; q < kU; ++q) into[r][q] = reinterpret_cast<const int4 *>(bpack)[entry_of(d, r, q, nu)];
    };
    load_recs(rec, dc, cur.u1 - cur.u0);
#if defined(G4S_PROFILE_BIG) && defined(G4S_PROFILE_SIZES)
    unsigned long long chunk_t0 = __builtin_amdgcn_s_memtime(), size_cnt[4] = {0, 0, 0, 0}, size_ticks[4] = {0, 0, 0, 0};
#endif
    for (;;) {
        const int nu = cur.u1 - cur.u0;
        const bool more = g + G < nitems;                          // uniform
        // what the NEXT chunk needs first: its descriptors (the item arrived a chunk ago); the item behind it
        int4 dn = make_int4(0, 1, 0, 0), dxn = make_int4(0, 1, 0, 0);
        if (more) { dn = desc_load(nxt); dxn = extras_load(nxt, 0); }
        const RankItem nxt2 = item_at(g + 2 * G);
        // ---- mark
#ifdef G4S_PROFILE_BIG
#ifndef G4S_PROFILE_HIST
        BIG_PROF(13);                                              // (loop top: item / descriptor requests)
#endif
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int q = 0; q < kU; ++q) asm volatile("" :: "v"(rec[r][q].x));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef G4S_PROFILE_HIST
        BIG_PROF(12);                                              // waiting for this chunk's records (requested behind the previous chunk's accumulate step)
#endif
#endif
        int wr[kR][kU];                                            // (the record's .x is dead from here on: the compiler reuses that register of the tuple)
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int q = 0; q < kU; ++q) { wr[r][q] = place_rec(rec[r][q].x, cur.pbase, lane < d_len(dc, r, q, nu)); mark(wr[r][q]); }
#ifdef G4S_PROFILE_BIG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BIG_PROF(14);                                              // the register rounds' marks alone
#endif
        // crowded chunks: the wave's units past the register rounds, by column only, 64 per batch (lane e of dx = the descriptor of the batch's e-th unit)
        for (int eb = 0;; eb += 64) {
            const int ne = extras_in(eb, nu);
            if (ne <= 0) break;
            if (eb) dx = extras_load(cur, eb);
            stream_unit_groups<kGM, int>(ne,
                [&](int (&c)[kGM], int e0) {
#pragma unroll
                    for (int q = 0; q < kGM; ++q) { const int e = min(e0 + q, ne - 1); c[q] = bcol2[__builtin_amdgcn_readlane(dx.x, e) + min(lane, __builtin_amdgcn_readlane(dx.y, e) - 1)]; }
                },
                [&](int (&c)[kGM], int e0) {
#pragma unroll
                    for (int q = 0; q < kGM; ++q)
                        if (e0 + q < ne) mark(place_of(c[q], cur.wbase, lane < __builtin_amdgcn_readlane(dx.y, e0 + q)));
                });
            if (ne < 64) break;
        }
        BIG_PROF(0);
        __syncthreads();
        BIG_PROF(1);
#if G4S_SPGEMM_RANK_AHEAD
        // the next chunk's records, into registers of their own: they have this chunk's rank, accumulate and store steps to arrive (requested behind the accumulate
        // step, when this chunk's registers are free, the mark step still waited 1.7 K cycles per chunk for them: profiles/r05_spgemm_rank_sections.txt)
        int4 recn[kR][kU];
        load_recs(recn, dn, more ? nxt.u1 - nxt.u0 : 0);
#endif
        // ---- ranks
        {
            unsigned long long w7[kWPT];
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < kWPT; ++j) { w7[j] = BM[t * kWPT + j]; cnt += __popcll(w7[j]); }
            const int incl = (int)wave_inclusive_sum((unsigned)cnt);
            if (lane == 63) ctrl[wave] = incl;
            __syncthreads();
            BIG_PROF(2);
            const unsigned sc = wave_inclusive_sum(lane < kWaves ? (unsigned)ctrl[lane] : 0u);
            int run = incl - cnt;
            if (wave > 0) run += (int)__builtin_amdgcn_readlane(sc, wave - 1);
#pragma unroll
            for (int j = 0; j < kWPT; ++j) { BM[t * kWPT + j] = w7[j] | ((unsigned long long)run << 48); run += __popcll(w7[j]); }
        }
        __syncthreads();
        BIG_PROF(3);
        // ---- accumulate
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int q = 0; q < kU; ++q)                             // multop / addop, hash_mult.h:583-593
                accumulate(wr[r][q], rec[r][q].y, d_val(dc, r, q) * __longlong_as_double(((long long)rec[r][q].w << 32) | (unsigned)rec[r][q].z));
#ifdef G4S_PROFILE_BIG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BIG_PROF(15);                                              // the register rounds' accumulate alone
#endif
        for (int eb = 0;; eb += 64) {
            const int ne = extras_in(eb, nu);
            if (ne <= 0) break;
            if (eb) dx = extras_load(cur, eb); else if (extras_in(64, nu) > 0) dx = extras_load(cur, 0);   // (the mark step went through further batches: the first one again)
            stream_unit_groups<kGA, int4>(ne,
                [&](int4 (&c)[kGA], int e0) {
#pragma unroll
                    for (int q = 0; q < kGA; ++q) {
                        const int e = min(e0 + q, ne - 1);
                        c[q] = reinterpret_cast<const int4 *>(bpack)[__builtin_amdgcn_readlane(dx.x, e) + min(lane, __builtin_amdgcn_readlane(dx.y, e) - 1)];
                    }
                },
                [&](int4 (&c)[kGA], int e0) {
#pragma unroll
                    for (int q = 0; q < kGA; ++q)
                        if (e0 + q < ne) {
                            const double av = __longlong_as_double(((long long)__builtin_amdgcn_readlane(dx.w, e0 + q) << 32) | (unsigned)__builtin_amdgcn_readlane(dx.z, e0 + q));
                            accumulate(place_rec(c[q].x, cur.pbase, lane < __builtin_amdgcn_readlane(dx.y, e0 + q)), c[q].y, av * __longlong_as_double(((long long)c[q].w << 32) | (unsigned)c[q].z));
                        }
                });
            if (ne < 64) break;
        }
#ifdef G4S_PROFILE_BIG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BIG_PROF(4);
        prof_acc[9] += 1; prof_acc[10] += nu;
#ifdef G4S_PROFILE_HIST                                            // how crowded are the chunks? (slots 11-13: chunks past the register rounds, their units, the units past the rounds)
        prof_acc[11] += nu > kR * kWaves * kU; prof_acc[12] += nu > kR * kWaves * kU ? nu : 0; prof_acc[13] += max(nu - kR * kWaves * kU, 0);
#else
        prof_acc[11] += cur.qn;
#endif
#endif
        // the next chunk's records: the registers are free, and the store step and two barriers stand between here and the next mark step. (Unconditional —
        // past the last chunk the descriptors are the empty ones and entry 0 is read: under `if (more)` the records became phi nodes whose register shuffles
        // waited for the loads on the spot.)
#if !G4S_SPGEMM_RANK_AHEAD
        load_recs(rec, dn, more ? nxt.u1 - nxt.u0 : 0);
#endif
        BIG_PROF(5);
        __syncthreads();
        BIG_PROF(6);
        // ---- store, clean
        {
            constexpr int kH = (kPer + 1) / 2;                      // in two parts: the step's temporaries stay under the register count the products' state sets
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double val[kH];
                int col[kH];
#pragma unroll
                for (int u = 0; u < kH; ++u) { const int i = min(t + (h * kH + u) * T, max(cur.qn, 1) - 1); val[u] = V[i]; col[u] = KC[i]; }
#pragma unroll
                for (int u = 0; u < kH; ++u) {
                    const int i = t + (h * kH + u) * T;
                    if (h * kH + u < kPer && i < cur.qn) { cval[(long long)cur.out0 + i] = val[u]; ccol[(long long)cur.out0 + i] = col[u]; V[i] = 0.0; }
                }
            }
#pragma unroll
            for (int j = 0; j < kWPT; ++j) BM[t * kWPT + j] = 0ull;
        }
        BIG_PROF(7);
        __syncthreads();
        BIG_PROF(8);
#if defined(G4S_PROFILE_BIG) && defined(G4S_PROFILE_SIZES)          // (a third build: chunks and their clock by output count — slots 12 … 15 and 5 are reused, see tools/rank_prof.py --sizes)
        {
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();
            const int b_ = cur.qn <= 512 ? 0 : cur.qn <= 2048 ? 1 : cur.qn <= 6144 ? 2 : 3;
            size_cnt[b_] += 1; size_ticks[b_] += now_ - chunk_t0; chunk_t0 = now_;
        }
#endif
        if (!more) break;
        dc = dn; dx = dxn;
#if G4S_SPGEMM_RANK_AHEAD
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int q = 0; q < kU; ++q) rec[r][q] = recn[r][q];
#endif
        g += G; cur = nxt; nxt = nxt2;
    }
#if defined(G4S_PROFILE_BIG) && defined(G4S_PROFILE_SIZES)
    if (threadIdx.x == BIG_PROF_TID && (blockIdx.x & 15) == 0)
        for (int b_ = 0; b_ < 4; ++b_) { atomicAdd(&g_big_prof[48 + 2 * b_], size_cnt[b_]); atomicAdd(&g_big_prof[48 + 2 * b_ + 1], size_ticks[b_]); }
#endif
    BIG_PROF_FLUSH;
}
