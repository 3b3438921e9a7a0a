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
//   numeric    spgemm_numeric_rank_kernel, one chunk at a time: the chunk's products are loaded ONCE (the first round stays in registers), their columns are
//              marked in an LDS bitmap of 48-column words, the owner threads turn the top 16 bits of every word into the word's exclusive rank, and a product's
//              slot is then rank(word) + popcount(bits below) — one LDS read, no search, no bucket index, no column list; the product adds its value into
//              V[slot] (ds_add_f64) and stores its column into KC[slot] (every product of a slot stores the same id), and the chunk leaves as two coalesced
//              streams (cval, ccol through the column map). LDS per output: 8 B sum + 4 B column + 7 B of bitmap share = 152 KiB per workgroup.
#pragma once

#ifndef G4S_RANK_T
#define G4S_RANK_T 1024                                    /* threads of the rank kernel's workgroup: 1024 (one per CU, 152 KiB) or 512 (two per CU, 76 KiB each, half the chunk and segment) */
#endif
constexpr int kRankT = G4S_RANK_T;
constexpr int kRankChunk = 8 * kRankT;                     // outputs per value chunk (8 192)
constexpr int kRankWordCols = 48;                          // columns per 64-bit LDS word: bits 0–47 presence, bits 48–63 the word's exclusive rank within the chunk (< 8192)
constexpr int kRankWords = 7 * kRankT;                     // words of a chunk's bitmap: 7 per thread (7 168 words, 56 KiB)
constexpr int kRankWin = kRankWords * kRankWordCols;       // 344 064 columns per segment = 336 symbolic threads of 1 024 columns each
constexpr int kRankSegThreads = kRankWin / 1024;
static_assert(kRankWin % 1024 == 0 && kRankChunk >= 1024, "a symbolic thread (1 024 columns) holds at most one count cut and never straddles a segment");
#ifndef G4S_RANK_PACK
#define G4S_RANK_PACK 1                                    /* 1: B's entries as {compact column, column, value} records, one 16-byte load per product and no gather in the store step */
#endif
#ifndef G4S_SPGEMM_RANK_UPR
#define G4S_SPGEMM_RANK_UPR 8                              /* 64-entry units a wave keeps in registers per chunk (16 waves × 8 × 64 = one chunk of products at compression 1) */
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
    const int b = (g0 + kRankChunk - 1) / kRankChunk;               // the first multiple of the chunk size at or behind g0
    int k = b * kRankChunk - g0;
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
    const int ncut = nz > 0 ? (nz - 1) / kRankChunk : 0;
    RankChunk *dst = WRITE ? out + choff[i] : nullptr;
    int s = 0, b = 1, cnt = 0, cur_o = -1, cur_c = 0;
    auto flush = [&](int next_o) {
        if (cur_o >= 0 && next_o > cur_o) {
            if constexpr (WRITE) dst[cnt] = RankChunk{cur_o, next_o - cur_o, cur_c, cur_c / kRankWin};
            ++cnt;
        }
    };
    while (s < nseg || b <= ncut) {
        const int os = s < nseg ? min(seg[s], nz) : INT_MAX, ob = b <= ncut ? b * kRankChunk : INT_MAX;
        const int cs = s * kRankWin, cb = b <= ncut ? bc[b - 1] : 0;
        const bool take_seg = os != ob ? os < ob : cs <= cb;
        const int o = take_seg ? os : ob, c = take_seg ? cs : cb;
        if (take_seg) ++s; else ++b;
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
struct __attribute__((aligned(16))) BPack { int c2, col; double val; };
__global__ void pack_b_kernel(long long nnz, const int *__restrict__ c2, const int *__restrict__ col, const double *__restrict__ val, BPack *__restrict__ out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz) out[k] = BPack{c2[k], col[k], val[k]};
}

// A row of the rank launch, packed per list position (one 64-byte scalar load): its place in C, its units, its chunk list and the first chunk itself.
struct __attribute__((aligned(16))) RankRowMeta { int row, na, off, nch; int u0, u1, ch0, pad; long long ioff, pad2; RankChunk first; };
__global__ void rank_row_meta_kernel(int n, const int *__restrict__ rows, const int *__restrict__ arpt, const int *__restrict__ crpt, const long long *__restrict__ item_off,
                                     const int *__restrict__ uoff, const int *__restrict__ choff, const RankChunk *__restrict__ chunks, RankRowMeta *__restrict__ meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rows[i];
    const long long total_items = item_off[n];
    RankRowMeta m;
    m.row = r; m.na = arpt[r + 1] - arpt[r]; m.off = crpt[r]; m.nch = choff[i + 1] - choff[i]; m.ch0 = choff[i]; m.pad = 0; m.pad2 = 0;
    m.ioff = item_off[i]; m.u0 = uoff[m.ioff]; m.u1 = uoff[min(m.ioff + m.na, total_items)];
    m.first = chunks[m.ch0];
    meta[i] = m;
}

// ---- numeric: see the header of this file. One 1024-thread workgroup per CU (152 KiB of LDS), rows longest first through a counter, a row's metadata one row
// ahead, the NEXT chunk's first round of products requested before the current chunk is stored.
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4, 4))) void spgemm_numeric_rank_kernel(
    int nrows, int *__restrict__ next_row, const int *__restrict__ bcol /* compact ids when col_of is given */, const int *__restrict__ col_of, const double *__restrict__ bval,
    int *__restrict__ ccol, double *__restrict__ cval, const long long *__restrict__ item_off, const int *__restrict__ uoff, const UnitDesc *__restrict__ U,
    const RankRowMeta *__restrict__ meta, const RankChunk *__restrict__ chunks, const BPack *__restrict__ bpack /* G4S_RANK_PACK */)
{
    static_assert(T == kRankT, "7 bitmap words and 8 outputs per thread");
    constexpr int kU = G4S_SPGEMM_RANK_UPR, kWaves = T / 64, kPer = kRankChunk / T, kWPT = kRankWords / T;
    extern __shared__ int lds_i[];                                 // dynamic only (Guideline 17): [V: 8192 fp64][KC: 8192 int][BM: 7168 × 64 bit][ctrl: 64 int]
    double *V = reinterpret_cast<double *>(lds_i);
    int *KC = lds_i + 2 * kRankChunk;
    unsigned long long *BM = reinterpret_cast<unsigned long long *>(lds_i + 3 * kRankChunk);
    unsigned *BM32 = reinterpret_cast<unsigned *>(BM);
    int *ctrl = lds_i + 3 * kRankChunk + 2 * kRankWords;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const long long total_items = item_off[nrows];
    auto load_meta = [&](int idx) { return meta[min(idx, nrows - 1)]; };   // uniform index: one scalar load (a ticket past the end reads the last row: never used)

    // round data of the chunk in flight: kU units per wave, a lane's entry of each
    int rc[kU], ro[kU];                                            // compact column (bitmap), column as it goes to ccol
    double rv[kU], rav[kU];
    bool rok[kU];
    BIG_PROF_DECL_RANK;
    auto issue_round = [&](const UnitDesc (&d)[kU], int nu, int g) {   // a lane's entry of each of the wave's kU units
#pragma unroll
        for (int q = 0; q < kU; ++q) {
            const int len = nu > 0 ? d[q].len : 1, bpos = nu > 0 ? d[q].bpos : 0;
            rok[q] = g + q < nu && lane < len;
            const int kk = bpos + min(lane, len - 1);
#if G4S_RANK_PACK
            const int4 e = reinterpret_cast<const int4 *>(bpack)[kk];
            rc[q] = e.x; ro[q] = e.y;
            rv[q] = __longlong_as_double(((long long)e.w << 32) | (unsigned)e.z);
#else
            rc[q] = bcol[kk]; ro[q] = rc[q];
            rv[q] = bval[kk];
#endif
            rav[q] = __longlong_as_double(((long long)d[q].av_hi << 32) | (unsigned)d[q].av_lo);
        }
    };
    auto request_round = [&](int cu0, int nu, int g) {             // descriptors first (uniform: one s_load_dwordx4 each, all in flight), then the vector loads
        UnitDesc d[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) d[q] = U[cu0 + min(g + q, max(nu, 1) - 1)];
        issue_round(d, nu, g);
    };
    // (word << 6 | bit) of a column inside the chunk's segment; −1 for a column outside it (only with arrays that changed under a carried symbolic state)
    auto place_of = [&](int col, int wbase, bool ok) {
        const unsigned rel = (unsigned)(col - wbase);
        const unsigned w = __umulhi(rel, 0xAAAAAAABu) >> 5;        // rel / 48
        return (ok && rel < (unsigned)kRankWin) ? (int)((w << 6) | (rel - w * 48u)) : -1;
    };
    auto mark = [&](int wr) {
        if (wr >= 0) atomicOr(&BM32[2 * (wr >> 6) + ((wr >> 5) & 1)], 1u << (wr & 31));
    };
    auto accumulate = [&](int wr, int col, double prod) {
        if (wr < 0) return;
        const unsigned long long w = BM[wr >> 6];
        const int slot = ((int)(w >> 48) + __popcll(w & ((1ull << (wr & 63)) - 1ull))) & (kRankChunk - 1);   // (the mask: stay inside the chunk whatever the arrays hold)
        atomicAdd(&V[slot], prod);
        KC[slot] = col;
    };

    // clean slate once; every chunk leaves V and BM clean behind it
#pragma unroll
    for (int u = 0; u < kPer; ++u) { V[t + u * T] = 0.0; KC[t + u * T] = 0; }   // (KC: a slot no product writes — an empty chunk's slot 0 — must still hold a valid column id)
#pragma unroll
    for (int j = 0; j < kWPT; ++j) BM[t + j * T] = 0ull;
    if (t == 0) ctrl[32] = atomicAdd(next_row, 1);
    __syncthreads();
    int ridx = __builtin_amdgcn_readfirstlane(ctrl[32]);
    if (ridx >= nrows) return;                                      // uniform
    RankRowMeta cur = load_meta(ridx), nxt = cur;
    RankChunk rec = cur.first;
    int qi = 0, cu0 = cur.u0, cu1 = cur.u1, nridx = nrows;
    bool have_next_row = false;                                    // nxt / nridx are valid for the current row
    request_round(cu0, cu1 - cu0, wave * kU);
    for (;;) {
        const int nu = cu1 - cu0;
        const bool first_chunk = qi == 0, last_chunk = qi + 1 >= cur.nch;   // uniform
        if (first_chunk && t == 0) ctrl[33] = atomicAdd(next_row, 1);        // the next row's ticket: read behind this chunk's first barrier
        // the next chunk of this row: its units end where the chunk after it begins; its record
        const int cu2 = uoff[min(cur.ioff + (long long)min(qi + 2, cur.nch) * cur.na, total_items)];
        const RankChunk rec_n = chunks[cur.ch0 + min(qi + 1, cur.nch - 1)];
        const int wbase = rec.seg * kRankWin;
        // ---- mark: round 0 from the registers requested a chunk ago, further rounds (crowded chunks) by column only
        int wr[kU];
#pragma unroll
        for (int q = 0; q < kU; ++q) { wr[q] = place_of(rc[q], wbase, rok[q]); mark(wr[q]); }
        for (int g = (wave + kWaves) * kU; g < nu; g += kWaves * kU) {   // (uniform per wave)
            UnitDesc d[kU];
#pragma unroll
            for (int q = 0; q < kU; ++q) d[q] = U[cu0 + min(g + q, nu - 1)];
            int c2[kU];
#pragma unroll
            for (int q = 0; q < kU; ++q) c2[q] = bcol[d[q].bpos + min(lane, d[q].len - 1)];
#pragma unroll
            for (int q = 0; q < kU; ++q) mark(place_of(c2[q], wbase, g + q < nu && lane < d[q].len));
        }
        BIG_PROF(0);
        __syncthreads();
        BIG_PROF(1);
        if (first_chunk) { nridx = __builtin_amdgcn_readfirstlane(ctrl[33]); nxt = load_meta(nridx); have_next_row = true; }
        // ---- ranks: thread t owns the words [7t, 7t + 7); the top 16 bits of a word become the number of set bits in front of it
        unsigned long long w7[kWPT];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < kWPT; ++j) { w7[j] = BM[t * kWPT + j]; cnt += __popcll(w7[j]); }
        const int incl = (int)wave_inclusive_sum((unsigned)cnt);
        if (lane == 63) ctrl[wave] = incl;
        __syncthreads();
        BIG_PROF(2);
        {
            const unsigned sc = wave_inclusive_sum(lane < kWaves ? (unsigned)ctrl[lane] : 0u);
            int run = incl - cnt;
            if (wave > 0) run += (int)__builtin_amdgcn_readlane(sc, wave - 1);
#pragma unroll
            for (int j = 0; j < kWPT; ++j) { BM[t * kWPT + j] = w7[j] | ((unsigned long long)run << 48); run += __popcll(w7[j]); }
        }
        __syncthreads();
        BIG_PROF(3);
        // the next chunk (of this row, or the first one of the next row): its wave's kU descriptors are requested here, a whole accumulate step before they are
        // needed (requested at the point of use, the scalar loads of the descriptors were a round trip in front of the round's vector loads: 14 % of the kernel)
        const bool more = !last_chunk || nridx < nrows;            // uniform (nxt is valid: the next row's ticket was read behind the first barrier of this row's first chunk)
        const int ncu0 = last_chunk ? nxt.u0 : cu1, ncu1 = last_chunk ? nxt.u1 : cu2;
        int4 dq = make_int4(0, 1, 0, 0);
        if (more && lane < kU) dq = reinterpret_cast<const int4 *>(U)[ncu0 + min(wave * kU + lane, max(ncu1 - ncu0, 1) - 1)];
        // ---- accumulate: round 0 from registers, further rounds loaded again (columns and values)
#pragma unroll
        for (int q = 0; q < kU; ++q) accumulate(wr[q], ro[q], rav[q] * rv[q]);   // multop / addop, hash_mult.h:583-593
        for (int g = (wave + kWaves) * kU; g < nu; g += kWaves * kU) {
            request_round(cu0, nu, g);
#pragma unroll
            for (int q = 0; q < kU; ++q) accumulate(place_of(rc[q], wbase, rok[q]), ro[q], rav[q] * rv[q]);
        }
#ifdef G4S_PROFILE_BIG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BIG_PROF(4);
        prof_acc[9] += 1; prof_acc[10] += nu; prof_acc[11] += rec.qn;
#endif
        // ---- the next chunk's first round, requested now: it arrives under the barrier and the store step
        if (more) {
            UnitDesc d[kU];                                        // the descriptors came in as one vector load (lane q holds unit q's): no scalar round trip in front of the loads
#pragma unroll
            for (int q = 0; q < kU; ++q)
                d[q] = UnitDesc{__builtin_amdgcn_readlane(dq.x, q), __builtin_amdgcn_readlane(dq.y, q), __builtin_amdgcn_readlane(dq.z, q), __builtin_amdgcn_readlane(dq.w, q)};
            issue_round(d, ncu1 - ncu0, wave * kU);
        }
        BIG_PROF(5);
        __syncthreads();
        BIG_PROF(6);
        // ---- store: the chunk's sums and columns; leave V and the bitmap clean
        {
            double val[kPer];
            int col[kPer];
#pragma unroll
            for (int u = 0; u < kPer; ++u) { const int i = min(t + u * T, max(rec.qn, 1) - 1); val[u] = V[i]; col[u] = KC[i]; }
#if !G4S_RANK_PACK
            if (col_of) {
#pragma unroll
                for (int u = 0; u < kPer; ++u) col[u] = col_of[col[u]];
            }
#endif
            const long long o = (long long)cur.off + rec.o_lo;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = t + u * T;
                if (i < rec.qn) { cval[o + i] = val[u]; V[i] = 0.0; }
            }
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int i = t + u * T;
                if (i < rec.qn) ccol[o + i] = col[u];
            }
#pragma unroll
            for (int j = 0; j < kWPT; ++j) BM[t * kWPT + j] = 0ull;
        }
        BIG_PROF(7);
        __syncthreads();
        BIG_PROF(8);
        if (!more) break;
        if (last_chunk) { ridx = nridx; cur = nxt; rec = cur.first; qi = 0; have_next_row = false; }
        else { rec = rec_n; ++qi; }
        cu0 = ncu0; cu1 = ncu1;
    }
    BIG_PROF_FLUSH;
    (void)have_next_row; (void)ridx; (void)col_of; (void)bval; (void)bpack;
}
