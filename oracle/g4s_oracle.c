/*
 * g4s_oracle.c — CPU restatement of the reference's algorithms for the G4S sparse hot path.
 *
 * THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / reported CPU baseline. The product (libg4s_hip.so, g4s_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Pinning status:
 *   - graph gather/apply driver (oracle_graph_process): PINNED against the reference's own GraphProcess,
 *     compiled in place from /root/reference/deepmd/source/op/graph.h into oracle/_ref/ (oracle/Makefile).
 *   - SpGEMM / SpMV / MatrixMarket reader: PARITY UNPINNED. The reference holds no test, golden vector or
 *     fixture for mm/ or mv/ (SURVEY.md §4), its shipped binaries need Intel MKL, and its header library
 *     mm/inc cannot be compiled here without a stand-in for TBB's <scalable_allocator.h>
 *     (mm/inc/utility.h:11), which the rules of this build forbid. These functions restate the published
 *     source line by line (citations below) and are cross-checked against scipy.sparse in tests/.
 *
 * Every function names the reference file:line it follows. Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * SpMV  y = alpha·A·x + beta·y.
 * The reference has no CSR mat-vec (mv/mv.c:6-27 only times dense BLAS-2 on a dense copy). Definition used
 * here: per row, products accumulated left to right in stored order, multiply then add
 * (the multiply-add of hash_numeric, mm/inc/hash_mult.h:583-593, with B an n×1 matrix), no FMA contraction.
 * beta == 0 does not read y (BLAS-2 convention of the cblas_dgemv call at mv/mv.c:9). */
ORACLE_API void oracle_spmv_csr(int32_t rows, const int32_t *rowptr, const int32_t *colids, const double *values,
                                const double *x, double *y, double alpha, double beta)
{
    for (int32_t i = 0; i < rows; ++i) {
        double s = 0.0;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            double p = values[k] * x[colids[k]];
            s = s + p;
        }
        if (beta == 0.0) y[i] = alpha * s;
        else {
            double t = alpha * s;
            double u = beta * y[i];
            y[i] = t + u;
        }
    }
}

/* Same sum in long double with the absolute-value sum beside it, to bound the condition-dependent error
 * of any summation order:  |y_any_order − y_exact| ≤ rows_nnz · eps · abs_sum  (tests use it as the tolerance scale). */
ORACLE_API void oracle_spmv_csr_ld(int32_t rows, const int32_t *rowptr, const int32_t *colids, const double *values,
                                   const double *x, double *y_ld, double *abs_sum)
{
    for (int32_t i = 0; i < rows; ++i) {
        long double s = 0.0L, a = 0.0L;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            long double p = (long double)values[k] * (long double)x[colids[k]];
            s += p;
            a += p < 0 ? -p : p;
        }
        y_ld[i] = (double)s;
        abs_sum[i] = (double)a;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Exclusive prefix sum with the reference's calling convention: N outputs from N-1 inputs,
 * out[0] = 0, out[i+1] = out[i] + in[i]   (seq_scan, mm/inc/utility.h:156-163; callers pass rows+1). */
static void oracle_scan_i32(const int32_t *in, int32_t *out, int32_t N)
{
    out[0] = 0;
    for (int32_t i = 0; i < N - 1; ++i) out[i + 1] = out[i] + in[i];
}

/* flop = Σ_i Σ_{j∈A(i,:)} nnz(B(acol_j,:))  — compute_flop, mm/inc/mkl_mult.h:8-38; get_flop, hash_mult.h:46-62;
 * per-row counts as in BIN::set_intprod_num, BIN.h:78-95 (there an int; int64 here so that hubs cannot wrap). */
ORACLE_API int64_t oracle_spgemm_flop(int32_t M, const int32_t *arpt, const int32_t *acol, const int32_t *brpt,
                                      int64_t *row_flop /* may be NULL */)
{
    int64_t total = 0;
    for (int32_t i = 0; i < M; ++i) {
        int64_t f = 0;
        for (int32_t j = arpt[i]; j < arpt[i + 1]; ++j) f += brpt[acol[j] + 1] - brpt[acol[j]];
        if (row_flop) row_flop[i] = f;
        total += f;
    }
    return total;
}

/* bin_id[i]: 0 for an empty row, else 1 + min{ j : min(row_flop, cols) <= 8<<j } — BIN::set_bin_id, BIN.h:158-177
 * (min_ht_size = 8, hash_mult.h MIN_HT_S/MIN_HT_N). Table entries for bin b>0: 8 << (b-1). */
ORACLE_API void oracle_bin_id(int32_t rows, int32_t cols, const int64_t *row_flop, int8_t *bin_id)
{
    for (int32_t i = 0; i < rows; ++i) {
        int64_t nz = row_flop[i];
        if (nz > cols) nz = cols;
        if (nz == 0) { bin_id[i] = 0; continue; }
        int j = 0;
        while (nz > ((int64_t)8 << j)) j++;
        bin_id[i] = (int8_t)(j + 1);
    }
}

/* Equal-work contiguous partition: prefix-sum the per-row work, target = ceil(total/parts), boundary p is
 * lower_bound(prefix, target·p) — BIN::set_rows_offset, BIN.h:101-122. Used by the reference to split rows
 * over threads; the build uses the same rule to split rows over GPUs. offsets has parts+1 entries. */
ORACLE_API void oracle_rows_offset(int32_t rows, const int64_t *row_work, int32_t parts, int32_t *offsets)
{
    int64_t *ps = (int64_t *)malloc(sizeof(int64_t) * ((size_t)rows + 1));
    ps[0] = 0;
    for (int32_t i = 0; i < rows; ++i) ps[i + 1] = ps[i] + row_work[i];
    int64_t total = ps[rows];
    int64_t avg = (total + parts - 1) / parts;
    offsets[0] = 0;
    for (int32_t t = 0; t < parts; ++t) {
        int64_t target = avg * (t + 1);
        /* std::lower_bound over ps[0..rows] */
        int64_t lo = 0, hi = (int64_t)rows + 1;
        while (lo < hi) {
            int64_t mid = lo + (hi - lo) / 2;
            if (ps[mid] < target) lo = mid + 1; else hi = mid;
        }
        offsets[t + 1] = (int32_t)lo;
    }
    offsets[parts] = rows; /* BIN.h:120 */
    free(ps);
}

/* Symbolic phase: per output row, insert every B column reached into an open-addressing table
 * (hash = (key·107) & (size−1), linear probing, empty = −1) and count distinct keys; then the exclusive scan gives
 * crpt — hash_symbolic_kernel, mm/inc/hash_mult.h:65-109; hash_symbolic, :496-508; HASH_SCAL, define.h:12.
 * Returns nnz(C) as int64, or −1 if it does not fit the reference's int32 crpt. crpt has M+1 entries. */
ORACLE_API int64_t oracle_spgemm_symbolic(int32_t M, int32_t N, const int32_t *arpt, const int32_t *acol,
                                          const int32_t *brpt, const int32_t *bcol, int32_t *crpt)
{
    int64_t *row_flop = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
    int8_t *bin = (int8_t *)malloc((size_t)(M > 0 ? M : 1));
    int32_t *row_nz = (int32_t *)malloc(sizeof(int32_t) * ((size_t)M + 1));
    oracle_spgemm_flop(M, arpt, acol, brpt, row_flop);
    oracle_bin_id(M, N, row_flop, bin);
    int64_t max_ht = 8;
    for (int32_t i = 0; i < M; ++i)
        if (bin[i] > 0 && ((int64_t)8 << (bin[i] - 1)) > max_ht) max_ht = (int64_t)8 << (bin[i] - 1);
    int32_t *check = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_ht);
    int64_t total = 0;
    for (int32_t i = 0; i < M; ++i) {
        int32_t nz = 0;
        int bid = bin[i];
        if (bid > 0) {
            int64_t ht = (int64_t)8 << (bid - 1);
            for (int64_t j = 0; j < ht; ++j) check[j] = -1;
            for (int32_t j = arpt[i]; j < arpt[i + 1]; ++j) {
                int32_t t = acol[j];
                for (int32_t k = brpt[t]; k < brpt[t + 1]; ++k) {
                    int32_t key = bcol[k];
                    int64_t h = ((int64_t)(int32_t)((uint32_t)key * 107u)) & (ht - 1); /* IT arithmetic wraps as int32 */
                    for (;;) {
                        if (check[h] == key) break;
                        if (check[h] == -1) { check[h] = key; nz++; break; }
                        h = (h + 1) & (ht - 1);
                    }
                }
            }
        }
        row_nz[i] = nz;
        total += nz;
    }
    row_nz[M] = 0;
    if (total <= INT32_MAX) oracle_scan_i32(row_nz, crpt, M + 1);
    free(check); free(row_nz); free(bin); free(row_flop);
    return total <= INT32_MAX ? total : -1;
}

static int cmp_pair_col(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* Numeric phase: same traversal, t = a·b, hit → value = t + value, miss → insert; then compact the table and,
 * when sort_output, sort the row by column — hash_numeric, mm/inc/hash_mult.h:559-608;
 * sort_and_store_table2mat, :526-553. Accumulation order per C(i,c) is (j outer, k inner), multiply then add. */
ORACLE_API void oracle_spgemm_numeric(int32_t M, int32_t N, const int32_t *arpt, const int32_t *acol, const double *aval,
                                      const int32_t *brpt, const int32_t *bcol, const double *bval,
                                      const int32_t *crpt, int32_t *ccol, double *cval, int sort_output)
{
    int64_t *row_flop = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
    int8_t *bin = (int8_t *)malloc((size_t)(M > 0 ? M : 1));
    oracle_spgemm_flop(M, arpt, acol, brpt, row_flop);
    oracle_bin_id(M, N, row_flop, bin);
    int64_t max_ht = 8;
    for (int32_t i = 0; i < M; ++i)
        if (bin[i] > 0 && ((int64_t)8 << (bin[i] - 1)) > max_ht) max_ht = (int64_t)8 << (bin[i] - 1);
    int32_t *check = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_ht);
    double *value = (double *)malloc(sizeof(double) * (size_t)max_ht);
    struct pr { int32_t c; int32_t pad; double v; } *pv = (struct pr *)malloc(sizeof(struct pr) * (size_t)max_ht);
    for (int32_t i = 0; i < M; ++i) {
        int bid = bin[i];
        if (bid <= 0) continue;
        int64_t ht = (int64_t)8 << (bid - 1);
        int32_t off = crpt[i];
        for (int64_t j = 0; j < ht; ++j) check[j] = -1;
        for (int32_t j = arpt[i]; j < arpt[i + 1]; ++j) {
            int32_t t = acol[j];
            double av = aval[j];
            for (int32_t k = brpt[t]; k < brpt[t + 1]; ++k) {
                double tv = av * bval[k];
                int32_t key = bcol[k];
                int64_t h = ((int64_t)(int32_t)((uint32_t)key * 107u)) & (ht - 1);
                for (;;) {
                    if (check[h] == key) { value[h] = tv + value[h]; break; }
                    if (check[h] == -1) { check[h] = key; value[h] = tv; break; }
                    h = (h + 1) & (ht - 1);
                }
            }
        }
        int32_t idx = 0;
        for (int64_t j = 0; j < ht; ++j)
            if (check[j] != -1) { pv[idx].c = check[j]; pv[idx].pad = 0; pv[idx].v = value[j]; idx++; }
        if (sort_output) qsort(pv, (size_t)idx, sizeof(struct pr), cmp_pair_col); /* keys are distinct: any sort gives one order */
        for (int32_t j = 0; j < idx; ++j) { ccol[off + j] = pv[j].c; cval[off + j] = pv[j].v; }
    }
    free(pv); free(value); free(check); free(bin); free(row_flop);
}

/* ------------------------------------------------------------------------------------------------
 * Two more SpGEMM algorithms of the reference, restated as independent CPU cross-checks of the hash oracle above (SURVEY.md §8 a11:
 * "no GPU version needed") — never called by any main() of the reference, used here only by tests/test_oracle_cpu.py.
 *
 * (1) Outer-product SpGEMM — OuterSpGEMM / OuterSpGEMM_stage, mm/inc/outer_mult.h:271-542: for every inner index idx, every entry
 *     A(i, idx) of A's COLUMN idx meets every entry B(idx, k) of B's ROW idx (:316-330) and emits the triple (i, k, a·b); the triples
 *     are binned by blocks of rows (nrows_per_blocker :281, the two memcpy-staged passes :316-414 only regroup them), sorted by
 *     (row, col) (doRadixSort :420-431), equal keys merged by summation (doMerge :434-449) and compacted into CSR (:470-488).
 *     Restated single-threaded: triples generated in ascending idx, a stable counting sort by row block and a stable LSD radix sort by
 *     (row, col) inside a block keep, for equal keys, the generation order — so every c_ik is summed in ascending idx, which is also
 *     the order of the row-wise hash loop above (A's row entries ascend in column): values come out bit-identical to it.
 * (2) Heap SpGEMM — HeapSpGEMM, mm/inc/heap_mult.h:47-223: written for CSC (C's column i = k-way merge of the columns of A selected by
 *     B's column i, a heap keyed by row id: initial heap :133-143, pop / accumulate-if-same-key / refill :146-171). The reference's
 *     CSR data run through it with the operands swapped (Cᵀ = Bᵀ·Aᵀ); restated directly in CSR: C's row i = k-way merge of the rows of
 *     B selected by A's row i, heap keyed by column id. Rows come out sorted by column. Equal keys are added in pop order, which for
 *     ties depends on the heap's shape: values agree with the hash oracle to rounding, index arrays exactly.
 * Both return nnz(C); two-call protocol: crpt is always written (M+1 entries), ccol/cval only when non-NULL. */
typedef struct { int32_t r, c; double v; } otriple;

ORACLE_API int64_t oracle_spgemm_outer(int32_t M, int32_t K, int32_t N, const int32_t *arpt, const int32_t *acol, const double *aval,
                                       const int32_t *brpt, const int32_t *bcol, const double *bval, int32_t nblockers,
                                       int32_t *crpt, int32_t *ccol, double *cval)
{
    /* A as CSC (the reference takes a CSC<IT,NT> A): column pointers by counting, entries of a column in ascending row */
    int64_t annz = arpt[M];
    int32_t *cptr = (int32_t *)calloc((size_t)K + 2, sizeof(int32_t));
    int32_t *crow = (int32_t *)malloc(sizeof(int32_t) * (size_t)(annz > 0 ? annz : 1));
    double *cvalA = (double *)malloc(sizeof(double) * (size_t)(annz > 0 ? annz : 1));
    for (int64_t k = 0; k < annz; ++k) cptr[acol[k] + 2]++;
    for (int32_t j = 0; j < K; ++j) cptr[j + 2] += cptr[j + 1];
    for (int32_t i = 0; i < M; ++i)
        for (int32_t k = arpt[i]; k < arpt[i + 1]; ++k) { int32_t p = cptr[acol[k] + 1]++; crow[p] = i; cvalA[p] = aval[k]; }
    /* now column j of A is [cptr[j], cptr[j+1]) */
    int64_t flop = 0;
    for (int32_t j = 0; j < K; ++j) flop += (int64_t)(cptr[j + 1] - cptr[j]) * (brpt[j + 1] - brpt[j]);
    otriple *t = (otriple *)malloc(sizeof(otriple) * (size_t)(flop > 0 ? flop : 1)), *u = (otriple *)malloc(sizeof(otriple) * (size_t)(flop > 0 ? flop : 1));
    int64_t n = 0;
    for (int32_t idx = 0; idx < K; ++idx)                                   /* outer_mult.h:316-330 */
        for (int32_t j = cptr[idx]; j < cptr[idx + 1]; ++j)
            for (int32_t k = brpt[idx]; k < brpt[idx + 1]; ++k) { t[n].r = crow[j]; t[n].c = bcol[k]; t[n].v = cvalA[j] * bval[k]; n++; }
    /* row blocks (:281): a stable counting sort by block, then inside the whole array a stable LSD radix sort on the fused key would do
     * the same; the block pass is kept because it is the algorithm's structure */
    if (nblockers < 1) nblockers = 1;
    int32_t per = M <= nblockers * 64 ? 64 : (M + nblockers - 1) / nblockers;
    int32_t nb = (M + per - 1) / per + 1;
    int64_t *bstart = (int64_t *)calloc((size_t)nb + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) bstart[t[i].r / per + 1]++;
    for (int32_t b = 0; b < nb; ++b) bstart[b + 1] += bstart[b];
    { int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nb + 1)); memcpy(pos, bstart, sizeof(int64_t) * ((size_t)nb + 1));
      for (int64_t i = 0; i < n; ++i) u[pos[t[i].r / per]++] = t[i];
      free(pos); }
    /* stable LSD radix sort of every block by (row, col), 8 bits a pass: col bytes first, then row bytes */
    for (int32_t b = 0; b < nb; ++b) {
        otriple *src = u + bstart[b], *dst = t + bstart[b];
        int64_t m = bstart[b + 1] - bstart[b];
        if (m <= 1) { if (m == 1) dst[0] = src[0]; continue; }
        for (int pass = 0; pass < 8; ++pass) {
            int64_t cnt[257] = {0};
            for (int64_t i = 0; i < m; ++i) { uint32_t key = pass < 4 ? (uint32_t)src[i].c : (uint32_t)src[i].r; cnt[((key >> (8 * (pass & 3))) & 255u) + 1]++; }
            for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
            for (int64_t i = 0; i < m; ++i) { uint32_t key = pass < 4 ? (uint32_t)src[i].c : (uint32_t)src[i].r; dst[cnt[(key >> (8 * (pass & 3))) & 255u]++] = src[i]; }
            otriple *sw = src; src = dst; dst = sw;
        }
        /* 8 passes: the sorted block is back in `src` == u + bstart[b]; copy to t for the merge below */
        memcpy(t + bstart[b], u + bstart[b], sizeof(otriple) * (size_t)m);
    }
    /* merge equal keys (doMerge :434-449) and compact into CSR (:470-488) */
    for (int32_t i = 0; i <= M; ++i) crpt[i] = 0;
    int64_t out = 0;
    for (int64_t i = 0; i < n;) {
        int64_t j = i;
        double sum = t[i].v;
        for (j = i + 1; j < n && t[j].r == t[i].r && t[j].c == t[i].c; ++j) sum = sum + t[j].v;
        crpt[t[i].r + 1]++;
        if (ccol) { ccol[out] = t[i].c; cval[out] = sum; }
        out++;
        i = j;
    }
    for (int32_t i = 0; i < M; ++i) crpt[i + 1] += crpt[i];
    free(bstart); free(t); free(u); free(cptr); free(crow); free(cvalA);
    (void)N;
    return out;
}

typedef struct { int32_t key, runr, loc; double value; } hentry;
static void heap_sift_down(hentry *h, int32_t n, int32_t i)
{
    for (;;) {
        int32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l].key < h[m].key) m = l;
        if (r < n && h[r].key < h[m].key) m = r;
        if (m == i) return;
        hentry x = h[i]; h[i] = h[m]; h[m] = x;
        i = m;
    }
}

ORACLE_API int64_t oracle_spgemm_heap(int32_t M, int32_t N, const int32_t *arpt, const int32_t *acol, const double *aval,
                                      const int32_t *brpt, const int32_t *bcol, const double *bval,
                                      int32_t *crpt, int32_t *ccol, double *cval)
{
    int32_t maxdeg = 1;
    for (int32_t i = 0; i < M; ++i) if (arpt[i + 1] - arpt[i] > maxdeg) maxdeg = arpt[i + 1] - arpt[i];
    hentry *h = (hentry *)malloc(sizeof(hentry) * (size_t)maxdeg);
    int64_t out = 0;
    crpt[0] = 0;
    for (int32_t i = 0; i < M; ++i) {
        int32_t hs = 0;
        for (int32_t j = arpt[i]; j < arpt[i + 1]; ++j) {                   /* heap_mult.h:133-143: one entry per non-empty selected row of B */
            int32_t inner = acol[j];
            if (brpt[inner + 1] > brpt[inner]) { h[hs].loc = 1; h[hs].runr = j; h[hs].value = aval[j] * bval[brpt[inner]]; h[hs].key = bcol[brpt[inner]]; hs++; }
        }
        for (int32_t k = hs / 2 - 1; k >= 0; --k) heap_sift_down(h, hs, k);
        int64_t row0 = out;
        int32_t last = -1;
        while (hs > 0) {
            hentry top = h[0];
            if (out > row0 && last == top.key) { if (cval) cval[out - 1] = top.value + cval[out - 1]; }   /* :150-153 */
            else { if (ccol) { ccol[out] = top.key; cval[out] = top.value; } last = top.key; out++; }
            int32_t inner = acol[top.runr];
            if (brpt[inner + 1] - brpt[inner] > top.loc) {                  /* :160-167: refill from the same row of B */
                int32_t idx = brpt[inner] + top.loc;
                h[0].loc = top.loc + 1; h[0].runr = top.runr; h[0].value = aval[top.runr] * bval[idx]; h[0].key = bcol[idx];
            } else { h[0] = h[hs - 1]; hs--; }
            heap_sift_down(h, hs, 0);
        }
        crpt[i + 1] = (int32_t)out;
    }
    free(h);
    (void)N;
    return out;
}

/* ------------------------------------------------------------------------------------------------
 * MatrixMarket coordinate reader with the semantics of CSR<IT,NT>::construct, mm/inc/CSR.h:485-669
 * (banner rules :441-478): matrix/coordinate only; real|integer|pattern|complex (real part kept :544-554,
 * pattern → 1.0 :532); 1-based → 0-based :565-568; symmetric / skew-symmetric mirror of off-diagonals
 * :571-623 (hermitian rejected :477); sort by the fused key cols·I+J :640-651; counting row pointer :654-661.
 * Duplicates are kept (not merged). Two-call protocol: pass NULL arrays to get sizes, then call again.
 * Returns 0 or a negative error. */
static int next_line(FILE *f, char *buf, size_t n) { return fgets(buf, (int)n, f) != NULL; }

struct coo { int64_t key; double v; int64_t seq; };
static int cmp_coo(const void *a, const void *b)
{
    const struct coo *x = (const struct coo *)a, *y = (const struct coo *)b;
    if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
    return (x->seq > y->seq) - (x->seq < y->seq); /* stable among duplicates (std::sort leaves it unspecified) */
}

ORACLE_API int oracle_mtx_read(const char *path, int32_t *rows_out, int32_t *cols_out, int64_t *nnz_out,
                               int32_t *rowptr, int32_t *colids, double *values)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[4096];
    if (!next_line(f, line, sizeof line)) { fclose(f); return -2; }
    char t0[64], t1[64], t2[64], t3[64], t4[64], t5[64];
    int nt = sscanf(line, "%63s %63s %63s %63s %63s %63s", t0, t1, t2, t3, t4, t5);
    if (nt != 5 || strcmp(t0, "%%MatrixMarket") != 0 || strcmp(t1, "matrix") != 0) { fclose(f); return -3; }
    if (strcmp(t2, "coordinate") != 0) { fclose(f); return -4; }
    int is_pattern = !strcmp(t3, "pattern"), is_complex = !strcmp(t3, "complex");
    if (!is_pattern && !is_complex && strcmp(t3, "real") && strcmp(t3, "integer")) { fclose(f); return -5; }
    int sym = 0;
    if (!strcmp(t4, "general")) sym = 0;
    else if (!strcmp(t4, "symmetric")) sym = 1;
    else if (!strcmp(t4, "skew-symmetric")) sym = 2;
    else { fclose(f); return -6; } /* hermitian: not implemented in the reference either */
    do { if (!next_line(f, line, sizeof line)) { fclose(f); return -7; } } while (line[0] == '%');
    long long r, c, n;
    if (sscanf(line, "%lld %lld %lld", &r, &c, &n) != 3) { fclose(f); return -8; }
    struct coo *e = (struct coo *)malloc(sizeof(struct coo) * (size_t)(2 * n + 1));
    int64_t cnt = 0;
    for (long long k = 0; k < n; ++k) {
        long long i, j; double v = 1.0, im;
        if (fscanf(f, "%lld %lld", &i, &j) != 2) { free(e); fclose(f); return -9; }
        if (is_complex) { if (fscanf(f, "%lf %lf", &v, &im) != 2) { free(e); fclose(f); return -9; } }
        else if (!is_pattern) { if (fscanf(f, "%lf", &v) != 1) { free(e); fclose(f); return -9; } }
        i -= 1; j -= 1;
        e[cnt].key = (int64_t)c * i + j; e[cnt].v = v; e[cnt].seq = cnt; cnt++;
        if (sym && i != j) { e[cnt].key = (int64_t)c * j + i; e[cnt].v = (sym == 2) ? -v : v; e[cnt].seq = cnt; cnt++; }
    }
    fclose(f);
    *rows_out = (int32_t)r; *cols_out = (int32_t)c; *nnz_out = cnt;
    if (rowptr && colids && values) {
        qsort(e, (size_t)cnt, sizeof(struct coo), cmp_coo);
        memset(rowptr, 0, sizeof(int32_t) * ((size_t)r + 1));
        for (int64_t k = 0; k < cnt; ++k) {
            int64_t I = e[k].key / c, J = e[k].key % c;
            rowptr[I + 1]++;
            colids[k] = (int32_t)J; values[k] = e[k].v;
        }
        for (long long i = 1; i <= r; ++i) rowptr[i] += rowptr[i - 1];
    }
    free(e);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Graph gather/apply driver: for vi in [0,numNodes): for nb in [0,degree): gather(vi,nb,…); apply(vi,…)
 * — GraphProcess, deepmd/source/op/graph.h:21-32, and the contract of spmm_dense inferred from its call
 * site citcoms/lib/Element_calculations.c:500 (the symbol itself is defined nowhere in the reference).
 * Sequential vertex order (the reference's threadNum=1 case; CitcomS's gather is only race-free there). */
typedef void (*fun_gather)(int, int, const double **, const double *, double *); /* global_defs.h:48 */
typedef void (*fun_apply)(int, const double **, const double *, double *);       /* global_defs.h:49 */

ORACLE_API void oracle_spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight,
                                  const double *vertexStates, double *temp, double *result,
                                  fun_gather gather, fun_apply apply, double *time, int threadNum)
{
    (void)temp; (void)threadNum;
    for (uint32_t vi = 0; vi < numNodes; ++vi) {
        for (uint32_t nb = 0; nb < degree; ++nb) gather((int)vi, (int)nb, edgeWeight, vertexStates, result);
        apply((int)vi, edgeWeight, vertexStates, result);
    }
    if (time) *time = 0.0;
}

/* CitcomS element-by-element stiffness mat-vec, the arithmetic of gather(), Element_calculations.c:453-471, with the
 * process globals (tempE->IEN, ->ID) made explicit and 0-based: for element e, local node a, dof i:
 *   Au[id[ien[e][a]][i]] += Σ_b ( K_e[ii]·u[id[nb][0]] + K_e[ii+1]·u[id[nb][1]] + K_e[ii+2]·u[id[nb][2]] ),
 *   ii = (dof·a + i)·n + dof·b, n = npe·dof  (index algebra of :463).
 * The three products of one b are summed first, then added to Au, exactly as the source expression does.
 * elt_k[e] is a row pointer (edgeWeight), base = 1 reproduces CitcomS's unused slot 0 (Drive_solvers.c:52-55). */
ORACLE_API void oracle_element_matvec(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id,
                                      const double **elt_k, int32_t base, const double *u, double *Au, int32_t neq)
{
    const int32_t n = npe * dof;
    for (int32_t i = 0; i < neq; ++i) Au[i] = 0.0; /* Element_calculations.c:495-496 */
    for (int32_t e = 0; e < nel; ++e) {
        const double *K = elt_k[e + base];
        for (int32_t a = 0; a < npe; ++a) {
            int32_t node = ien[e * npe + a];
            for (int32_t i = 0; i < dof; ++i) {
                int32_t aa = id[node * dof + i];
                for (int32_t b = 0; b < npe; ++b) {
                    int32_t ii = (dof * a + i) * n + dof * b;
                    int32_t nb = ien[e * npe + b];
                    double t = 0.0;
                    /* source: K[ii]*u0 + K[ii+1]*u1 + K[ii+2]*u2 evaluated left to right */
                    for (int32_t d = 0; d < dof; ++d) {
                        double p = K[ii + d] * u[id[nb * dof + d]];
                        t = (d == 0) ? p : t + p;
                    }
                    Au[aa] = Au[aa] + t;
                }
            }
        }
    }
}

/* DeePMD OptMatmul gather: result[e·K + a] = Σ_k xx[e][k]·w[k·K + a], accumulated from 0 in k order
 * — the lambda at deepmd/source/op/opt_matmul.cc:52-58 (apply is empty :59-61). xx rows via row pointers. */
ORACLE_API void oracle_dense_rows_times_matrix(int32_t M, int32_t N, int32_t K, const double **xx_rows, const double *w,
                                               double *result)
{
    for (int32_t e = 0; e < M; ++e)
        for (int32_t a = 0; a < K; ++a) {
            double s = 0.0;
            for (int32_t k = 0; k < N; ++k) { double p = xx_rows[e][k] * w[k * K + a]; s = s + p; }
            result[e * K + a] = s;
        }
}

/* Gradient of OptMatmul — deepmd/source/op/_opt_matmul_grad.py:6-12: dxx = matmul(grad, w, transpose_b) [M×N],
 * dw = matmul(xx, grad, transpose_a) [N×K]. Plain triple loops, sums accumulated from 0 in index order. */
ORACLE_API void oracle_dense_rows_times_matrix_grad(int32_t M, int32_t N, int32_t K, const double *xx, const double *w,
                                                    const double *grad, double *dxx, double *dw)
{
    for (int32_t i = 0; i < M; ++i)
        for (int32_t n = 0; n < N; ++n) {
            double s = 0.0;
            for (int32_t k = 0; k < K; ++k) { double p = grad[(size_t)i * K + k] * w[(size_t)n * K + k]; s = s + p; }
            dxx[(size_t)i * N + n] = s;
        }
    for (int32_t n = 0; n < N; ++n)
        for (int32_t k = 0; k < K; ++k) {
            double s = 0.0;
            for (int32_t i = 0; i < M; ++i) { double p = xx[(size_t)i * N + n] * grad[(size_t)i * K + k]; s = s + p; }
            dw[(size_t)n * K + k] = s;
        }
}

/* Cantera mixing rule: strictly-lower-triangle gather plus diagonal apply
 * — gather1/apply1/GraphProcess1 and gather2/apply2/GraphProcess2, cantera/src/thermo/RedlichKwongMFTP.cpp:927-983,
 * single rank (myid=0, numprocs=1), sequential. numbers==1 → form 1 (result[1] += x_i·b_i), else form 2. */
ORACLE_API void oracle_sym_quadratic_form(int32_t m, int32_t numbers, const double *a, const double *x, const double *b,
                                          double *result)
{
    for (int32_t i = 0; i < m; ++i) {
        for (int32_t j = 0; j < i; ++j) {
            size_t c1 = (size_t)i + (size_t)m * j, c2 = (size_t)j + (size_t)m * i;
            double tmp = x[i] * x[j];
            if (numbers == 1) {
                result[0] += tmp * (a[c1] + a[c2]);
            } else {
                result[0] += tmp * (a[numbers * c1] + a[numbers * c2]);
                result[1] += tmp * (a[numbers * c1 + 1] + a[numbers * c2 + 1]);
            }
        }
        size_t c = (size_t)i + (size_t)m * i;
        double tmp = x[i] * x[i];
        if (numbers == 1) {
            result[0] += tmp * a[c];
            result[1] += x[i] * b[i];
        } else {
            result[0] += tmp * a[numbers * c];
            result[1] += tmp * a[numbers * c + 1];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs (the build's own definitions; no reference counterpart — the reference reads
 * .mtx files that are not shipped). Counter-based so that the device generator reproduces them bit for bit.
 *
 * mix64 = SplitMix64 finaliser. R-MAT edge e of stream `seed`: for attempt t = 0,1,…: draw `scale` quadrant
 * choices from successive mix64 words (16 bits each, 4 per word) with (a,b,c,d) = (0.57,0.19,0.19,0.05); accept the
 * first attempt with row < n and col < n. Value of entry (i,j) depends only on (seed, i, j): U(−1,1). */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

ORACLE_API uint64_t oracle_mix64(uint64_t z) { return mix64(z); }

ORACLE_API double oracle_entry_value(uint64_t seed, int64_t i, int64_t j, int64_t n)
{
    uint64_t h = mix64(seed ^ mix64((uint64_t)(i * n + j) + 0x5851F42D4C957F2Dull));
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0; /* 53 bits → [−1,1) */
}

ORACLE_API double oracle_vector_value(uint64_t seed, int64_t i)
{
    uint64_t h = mix64(seed + 0xD1B54A32D192ED03ull * (uint64_t)(i + 1));
    return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

/* keys[e] = row·n + col for e in [e0, e0+count) */
ORACLE_API void oracle_rmat_edges(uint64_t seed, int32_t scale, int64_t n, int64_t e0, int64_t count, int64_t *keys)
{
    /* thresholds on a 16-bit draw: a=0.57, a+b=0.76, a+b+c=0.95 */
    const uint32_t TA = 37356u, TB = 49807u, TC = 62259u;
    for (int64_t q = 0; q < count; ++q) {
        uint64_t e = (uint64_t)(e0 + q);
        for (uint64_t attempt = 0;; ++attempt) {
            uint64_t base = mix64(seed ^ (e * 0x9E3779B97F4A7C15ull)) + attempt * 0xC2B2AE3D27D4EB4Full;
            int64_t row = 0, col = 0;
            uint64_t w = 0;
            for (int32_t lvl = 0; lvl < scale; ++lvl) {
                if ((lvl & 3) == 0) w = mix64(base + (uint64_t)(lvl >> 2));
                uint32_t r16 = (uint32_t)(w & 0xFFFFu);
                w >>= 16;
                int rb, cb;
                if (r16 < TA) { rb = 0; cb = 0; }
                else if (r16 < TB) { rb = 0; cb = 1; }
                else if (r16 < TC) { rb = 1; cb = 0; }
                else { rb = 1; cb = 1; }
                row = (row << 1) | rb;
                col = (col << 1) | cb;
            }
            if (row < n && col < n) { keys[q] = row * n + col; break; }
        }
    }
}

/* 5-point Laplacian on an nx×ny grid, natural row-major order, diag 4, off −1, Dirichlet;
 * columns ascending within a row. Returns nnz; arrays may be NULL to query. */
ORACLE_API int64_t oracle_laplacian5(int32_t nx, int32_t ny, int32_t *rowptr, int32_t *colids, double *values)
{
    int64_t k = 0;
    for (int32_t j = 0; j < ny; ++j)
        for (int32_t i = 0; i < nx; ++i) {
            int64_t r = (int64_t)j * nx + i;
            if (rowptr) rowptr[r] = (int32_t)k;
#define EMIT(cc, vv) do { if (colids) { colids[k] = (int32_t)(cc); values[k] = (vv); } k++; } while (0)
            if (j > 0) EMIT(r - nx, -1.0);
            if (i > 0) EMIT(r - 1, -1.0);
            EMIT(r, 4.0);
            if (i < nx - 1) EMIT(r + 1, -1.0);
            if (j < ny - 1) EMIT(r + nx, -1.0);
        }
    if (rowptr) rowptr[(int64_t)nx * ny] = (int32_t)k;
    return k;
}

/* 7-point Laplacian on nx×ny×nz, diag 6, off −1, rows [r0, r1) only (slab), global columns. */
ORACLE_API int64_t oracle_laplacian7_rows(int32_t nx, int32_t ny, int32_t nz, int64_t r0, int64_t r1,
                                          int32_t *rowptr, int32_t *colids, double *values)
{
    int64_t k = 0;
    const int64_t pl = (int64_t)nx * ny;
    for (int64_t r = r0; r < r1; ++r) {
        int32_t z = (int32_t)(r / pl), y = (int32_t)((r % pl) / nx), x = (int32_t)(r % nx);
        if (rowptr) rowptr[r - r0] = (int32_t)k;
        if (z > 0) EMIT(r - pl, -1.0);
        if (y > 0) EMIT(r - nx, -1.0);
        if (x > 0) EMIT(r - 1, -1.0);
        EMIT(r, 6.0);
        if (x < nx - 1) EMIT(r + 1, -1.0);
        if (y < ny - 1) EMIT(r + nx, -1.0);
        if (z < nz - 1) EMIT(r + pl, -1.0);
    }
#undef EMIT
    if (rowptr) rowptr[r1 - r0] = (int32_t)k;
    (void)nz;
    return k;
}

/* Banded matrix: row i holds columns max(0,i−hb) … min(n−1,i+hb), values from oracle_entry_value. */
ORACLE_API int64_t oracle_banded(int32_t n, int32_t hb, uint64_t seed, int32_t *rowptr, int32_t *colids, double *values)
{
    int64_t k = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (rowptr) rowptr[i] = (int32_t)k;
        int32_t lo = i - hb < 0 ? 0 : i - hb, hi = i + hb > n - 1 ? n - 1 : i + hb;
        for (int32_t c = lo; c <= hi; ++c) {
            if (colids) { colids[k] = c; values[k] = oracle_entry_value(seed, i, c, n); }
            k++;
        }
    }
    if (rowptr) rowptr[n] = (int32_t)k;
    return k;
}

/* ------------------------------------------------------------------------------------------------
 * Multi-threaded SpMV used ONLY as bench.py's reported cpu_baseline ("port"): rows split by equal nnz with the
 * rule of BIN::set_rows_offset (BIN.h:101-122), one OpenMP thread per range, same per-row arithmetic as oracle_spmv_csr. */
#ifdef _OPENMP
#include <omp.h>
#endif
ORACLE_API int oracle_spmv_csr_mt(int32_t rows, const int32_t *rowptr, const int32_t *colids, const double *values,
                                  const double *x, double *y, double alpha, double beta, int threads)
{
#ifdef _OPENMP
    if (threads < 1) threads = 1;
    int64_t *work = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rows > 0 ? rows : 1));
    int32_t *off = (int32_t *)malloc(sizeof(int32_t) * ((size_t)threads + 1));
    for (int32_t i = 0; i < rows; ++i) work[i] = rowptr[i + 1] - rowptr[i] + 1;
    oracle_rows_offset(rows, work, threads, off);
#pragma omp parallel num_threads(threads)
    {
        int t = omp_get_thread_num();
        int32_t r0 = off[t], r1 = off[t + 1];
        oracle_spmv_csr(r1 - r0, rowptr + r0, colids, values, x, y + r0, alpha, beta);
    }
    free(off); free(work);
    return threads;
#else
    oracle_spmv_csr(rows, rowptr, colids, values, x, y, alpha, beta);
    (void)threads;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * CitcomS Jacobi-preconditioned conjugate gradient around the element mat-vec — conj_grad,
 * citcoms/lib/General_matrix_functions.c:307-424, single process (global_vdot = plain dot product,
 * Global_operations.c:534-562 with Skip_neq = 0 and one rank), with the update order of the source:
 *   r1 = F, d0 = 0; residual = sqrt(r1·r1)
 *   while ((residual > acc && count < steps) || count == 0):
 *     z1 = BI∘r1; dotr1z1 = r1·z1; p2 = z1 (first) | z1 + (dotr1z1/dotr0z0)·p1; dotr0z0 = dotr1z1
 *     Ap = K·p2 with boundary rows zeroed (assemble_del2_u(...,strip_bcs=1), Element_calculations.c:428-509; BC_util.c:89-102)
 *     dotprod = p2·Ap; alpha = dotprod == 0 ? 1e-3 : dotr1z1/dotprod
 *     d0 += alpha·p2; r2 = r1 − alpha·Ap; residual = sqrt(r2·r2); rotate (r,z,p); count++
 *   strip_bcs_from_residual(d0)
 * BI is the inverse diagonal (build_diagonal_of_K, Element_calculations.c:580-611, inverted at Construct_arrays.c:469).
 * Returns the final residual; *cycles in: max steps, out: iterations done. res_hist (may be NULL) gets the residual per iteration. */
ORACLE_API void oracle_element_inverse_diagonal(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id,
                                                const double *elt_k /* contiguous nel·n·n */, double *BI, int32_t neq)
{
    const int32_t n = npe * dof;
    for (int32_t i = 0; i < neq; ++i) BI[i] = 0.0;
    for (int32_t e = 0; e < nel; ++e)
        for (int32_t a = 0; a < npe; ++a)
            for (int32_t d = 0; d < dof; ++d) {
                int32_t p = a * dof + d;
                BI[id[ien[e * npe + a] * dof + d]] += elt_k[(size_t)e * n * n + (size_t)p * n + p];
            }
    for (int32_t i = 0; i < neq; ++i) BI[i] = 1.0 / BI[i];
}

ORACLE_API double oracle_conj_grad_elem(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id,
                                        const double *elt_k /* contiguous */, int32_t neq, const double *BI,
                                        const int32_t *zero_resid, int32_t n_zero, const double *F, double *d0,
                                        double acc, int32_t *cycles, double *res_hist)
{
    const int32_t n = npe * dof;
    const double **rows = (const double **)malloc(sizeof(double *) * (size_t)(nel > 0 ? nel : 1));
    for (int32_t e = 0; e < nel; ++e) rows[e] = elt_k + (size_t)e * n * n;
    double *r1 = malloc(sizeof(double) * neq), *r2 = malloc(sizeof(double) * neq), *z1 = malloc(sizeof(double) * neq);
    double *p1 = malloc(sizeof(double) * neq), *p2 = malloc(sizeof(double) * neq), *Ap = malloc(sizeof(double) * neq);
    const int32_t steps = *cycles;
    double dotr0z0 = 0.0, residual, t = 0.0;
    for (int32_t i = 0; i < neq; ++i) { r1[i] = F[i]; d0[i] = 0.0; p1[i] = 0.0; }
    for (int32_t i = 0; i < neq; ++i) t += r1[i] * r1[i];
    residual = sqrt(t);
    int32_t count = 0;
    while ((residual > acc && count < steps) || count == 0) {
        double dotr1z1 = 0.0, dotprod = 0.0, alpha;
        for (int32_t i = 0; i < neq; ++i) z1[i] = BI[i] * r1[i];
        for (int32_t i = 0; i < neq; ++i) dotr1z1 += r1[i] * z1[i];
        if (count == 0) for (int32_t i = 0; i < neq; ++i) p2[i] = z1[i];
        else {
            const double beta = dotr1z1 / dotr0z0;
            for (int32_t i = 0; i < neq; ++i) p2[i] = z1[i] + beta * p1[i];
        }
        dotr0z0 = dotr1z1;
        oracle_element_matvec(nel, npe, dof, ien, id, rows, 0, p2, Ap, neq);
        for (int32_t i = 0; i < n_zero; ++i) Ap[zero_resid[i]] = 0.0;
        for (int32_t i = 0; i < neq; ++i) dotprod += p2[i] * Ap[i];
        alpha = (dotprod == 0.0) ? 1.0e-3 : dotr1z1 / dotprod;
        t = 0.0;
        for (int32_t i = 0; i < neq; ++i) { d0[i] += alpha * p2[i]; r2[i] = r1[i] - alpha * Ap[i]; }
        for (int32_t i = 0; i < neq; ++i) t += r2[i] * r2[i];
        residual = sqrt(t);
        if (res_hist) res_hist[count] = residual;
        { double *s = r1; r1 = r2; r2 = s; s = p1; p1 = p2; p2 = s; }
        count++;
    }
    *cycles = count;
    for (int32_t i = 0; i < n_zero; ++i) d0[zero_resid[i]] = 0.0;
    free(r1); free(r2); free(z1); free(p1); free(p2); free(Ap); free(rows);
    return residual;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * CitcomS incompressibility (Uzawa) iteration around the velocity solve — SURVEY.md §8 f1.
 * Single cap (m = 1), 0-based arrays, pressure unknowns = elements (npno = nel), g[e][p] = elt_del[e].g[p][0].
 * ---------------------------------------------------------------------------------------------------------------------------- */

/* assemble_div_u, citcoms/lib/Element_calculations.c:701-729: a outer, e inner; each term is (g0·U1 + g1·U2 + g2·U3) added to divU[e]. */
ORACLE_API void oracle_assemble_div_u(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id, const double *g,
                                      const double *U, double *divU)
{
    const int32_t n = npe * dof;
    for (int32_t e = 0; e < nel; ++e) divU[e] = 0.0;
    for (int32_t a = 0; a < npe; ++a) {
        const int32_t p = a * dof;
        for (int32_t e = 0; e < nel; ++e) {
            const int32_t b = ien[e * npe + a];
            double t = 0.0;
            for (int32_t d = 0; d < dof; ++d) { double q = g[(size_t)e * n + p + d] * U[id[b * dof + d]]; t = (d == 0) ? q : t + q; }
            divU[e] = divU[e] + t;
        }
    }
}

/* assemble_grad_p, Element_calculations.c:737-779: zero, skip elements with P == 0, scatter g·P in element order, strip boundary rows. */
ORACLE_API void oracle_assemble_grad_p(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id, const double *g,
                                       int32_t neq, const int32_t *zero_resid, int32_t n_zero, const double *P, double *gradP)
{
    const int32_t n = npe * dof;
    for (int32_t i = 0; i < neq; ++i) gradP[i] = 0.0;
    for (int32_t e = 0; e < nel; ++e) {
        if (0.0 == P[e]) continue;
        for (int32_t a = 0; a < npe; ++a) {
            const int32_t b = ien[e * npe + a];
            for (int32_t d = 0; d < dof; ++d) {
                double q = g[(size_t)e * n + a * dof + d] * P[e];
                gradP[id[b * dof + d]] = gradP[id[b * dof + d]] + q;
            }
        }
    }
    for (int32_t i = 0; i < n_zero; ++i) gradP[zero_resid[i]] = 0.0;
}

/* build_diagonal_of_Ahat + assemble_dAhatp_entry, Element_calculations.c:613-644, 785-830 (control.precondition on). */
ORACLE_API void oracle_build_diagonal_of_Ahat(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id, const double *g,
                                              const double *BI, double *BPI)
{
    const int32_t n = npe * dof;
    for (int32_t e = 0; e < nel; ++e) {
        double divU = 0.0;
        for (int32_t a = 0; a < npe; ++a) {
            const int32_t node = ien[e * npe + a];
            for (int32_t d = 0; d < dof; ++d) {
                const double ge = g[(size_t)e * n + a * dof + d];
                const double gradP = BI[id[node * dof + d]] * ge;     /* gradP[p] starts at 0 and receives one term */
                double q = ge * gradP;
                divU = divU + q;
            }
        }
        BPI[e] = (divU != 0.0) ? 1.0 / divU : 1.0;
    }
}

/* global_v_norm2 / global_p_norm2 / global_div_norm2 / global_pdot, citcoms/lib/Global_operations.c:565-656 (one process). */
static double uz_v_norm2(int32_t nno, int32_t dof, const int32_t *id, const double *nmass, double volume, const double *V)
{
    double temp = 0.0;
    for (int32_t i = 0; i < nno; ++i) {
        double s = 0.0;
        for (int32_t d = 0; d < dof; ++d) { double q = V[id[i * dof + d]] * V[id[i * dof + d]]; s = (d == 0) ? q : s + q; }
        double w = s * nmass[i];
        temp = temp + w;
    }
    return temp / volume;
}
static double uz_p_norm2(int32_t nel, const double *area, double volume, const double *P)
{
    double temp = 0.0;
    for (int32_t i = 0; i < nel; ++i) { double q = P[i] * P[i] * area[i]; temp = temp + q; }
    return temp / volume;
}
static double uz_div_norm2(int32_t nel, const double *area, double volume, const double *A)
{
    double temp = 0.0;
    for (int32_t i = 0; i < nel; ++i) { double q = A[i] * A[i] / area[i]; temp = temp + q; }
    return temp / volume;
}
static double uz_pdot(int32_t nel, const double *A, const double *B)
{
    double temp = 0.0;
    for (int32_t i = 0; i < nel; ++i) { double q = A[i] * B[i]; temp = temp + q; }
    return temp;
}

/* solve_del2_u with the conjugate-gradient branch, General_matrix_functions.c:48-146: d0 = 0; residual = conj_grad(...,
 * cycles = v_steps_low); valid = residual < acc. */
static int uz_solve_del2_u(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id, const double *elt_k, int32_t neq,
                           const double *BI, const int32_t *zero_resid, int32_t n_zero, const double *F, double *d0, double acc,
                           int32_t v_steps_low, int64_t *inner_total)
{
    int32_t cycles = v_steps_low;
    double residual = oracle_conj_grad_elem(nel, npe, dof, ien, id, elt_k, neq, BI, zero_resid, n_zero, F, d0, acc, &cycles, NULL);
    if (inner_total) *inner_total += cycles;
    return residual < acc ? 1 : 0;
}

/* solve_Ahat_p_fhat_CG, citcoms/lib/Stokes_flow_Incomp.c:188-452, incompressible (inv_gruneisen == 0: initial_vel_residual
 * :839-881 runs), no rigid-rotation removal. hist (may be NULL) receives 5 doubles per printed line (v_norm, p_norm, dvelocity,
 * dpressure, incompressibility), line 0 = before the loop. Returns the final incompressibility; *steps_max: in cap, out count. */
ORACLE_API double oracle_solve_Ahat_p_fhat_CG(int32_t nel, int32_t npe, int32_t dof, const int32_t *ien, const int32_t *id, int32_t nno,
                                              int32_t neq, const double *elt_k, const double *g, const double *BI, const double *BPI,
                                              const double *nmass, const double *area, double volume,
                                              const int32_t *zero_resid, int32_t n_zero, const double *FF, double *V, double *P,
                                              double imp, double inner_accuracy_scale, double v_res, int32_t v_steps_low,
                                              int32_t check_continuity_convergence, int32_t check_pressure_convergence,
                                              int32_t *steps_max, double *hist, int64_t *inner_iterations)
{
    const int32_t n = npe * dof;
    const double **rows = (const double **)malloc(sizeof(double *) * (size_t)(nel > 0 ? nel : 1));
    for (int32_t e = 0; e < nel; ++e) rows[e] = elt_k + (size_t)e * n * n;
    double *F = malloc(sizeof(double) * neq), *u1 = malloc(sizeof(double) * neq), *tmp = malloc(sizeof(double) * neq);
    double *r1 = malloc(sizeof(double) * nel), *r2 = malloc(sizeof(double) * nel), *z1 = malloc(sizeof(double) * nel);
    double *s1 = malloc(sizeof(double) * nel), *s2 = malloc(sizeof(double) * nel), *Fp = malloc(sizeof(double) * (nel > neq ? nel : neq));
    const double inner_imp = imp * inner_accuracy_scale;
    int32_t count = 0, converging = 0, valid;
    double alpha, delta, r0dotz0 = 0.0, r1dotz1, vdotv, pdotp = 0.0, incompressibility, dvelocity = 1.0, dpressure = 1.0;
    if (inner_iterations) *inner_iterations = 0;
    for (int32_t j = 0; j < neq; ++j) F[j] = FF[j];
    for (int32_t j = 0; j < nel; ++j) s1[j] = 0.0;

    /* initial_vel_residual: F = F − grad(P) − K·V, stripped; solve K·u1 = F; strip; V += u1 */
    oracle_assemble_grad_p(nel, npe, dof, ien, id, g, neq, zero_resid, n_zero, P, u1);
    for (int32_t i = 0; i < neq; ++i) F[i] = F[i] - u1[i];
    oracle_element_matvec(nel, npe, dof, ien, id, rows, 0, V, u1, neq);
    for (int32_t i = 0; i < n_zero; ++i) u1[zero_resid[i]] = 0.0;
    for (int32_t i = 0; i < neq; ++i) F[i] = F[i] - u1[i];
    for (int32_t i = 0; i < n_zero; ++i) F[zero_resid[i]] = 0.0;
    valid = uz_solve_del2_u(nel, npe, dof, ien, id, elt_k, neq, BI, zero_resid, n_zero, F, u1, inner_imp * v_res, v_steps_low, inner_iterations);
    for (int32_t i = 0; i < n_zero; ++i) u1[zero_resid[i]] = 0.0;
    for (int32_t i = 0; i < neq; ++i) V[i] = V[i] + u1[i];

    oracle_assemble_div_u(nel, npe, dof, ien, id, g, V, r1);
    vdotv = uz_v_norm2(nno, dof, id, nmass, volume, V);
    incompressibility = sqrt(uz_div_norm2(nel, area, volume, r1) / (1e-32 + vdotv));
    if (hist) { hist[0] = sqrt(vdotv); hist[1] = sqrt(uz_p_norm2(nel, area, volume, P)); hist[2] = dvelocity; hist[3] = dpressure; hist[4] = incompressibility; }

    for (;;) {
        /* keep_iterating, Stokes_flow_Incomp.c:150-162 */
        const int keep = check_continuity_convergence ? ((incompressibility > imp) || (converging < 2)) : ((incompressibility > imp) && (converging < 2));
        if (!(count < *steps_max && keep)) break;
        for (int32_t j = 0; j < nel; ++j) z1[j] = BPI[j] * r1[j];
        r1dotz1 = uz_pdot(nel, r1, z1);
        if (count == 0) for (int32_t j = 0; j < nel; ++j) s2[j] = z1[j];
        else {
            delta = r1dotz1 / r0dotz0;
            for (int32_t j = 0; j < nel; ++j) s2[j] = z1[j] + delta * s1[j];
        }
        /* solve K·u1 = grad(s2) */
        oracle_assemble_grad_p(nel, npe, dof, ien, id, g, neq, zero_resid, n_zero, s2, tmp);
        valid = uz_solve_del2_u(nel, npe, dof, ien, id, elt_k, neq, BI, zero_resid, n_zero, tmp, u1, inner_imp * v_res, v_steps_low, inner_iterations);
        for (int32_t i = 0; i < n_zero; ++i) u1[zero_resid[i]] = 0.0;
        oracle_assemble_div_u(nel, npe, dof, ien, id, g, u1, Fp);
        alpha = r1dotz1 / uz_pdot(nel, s2, Fp);
        for (int32_t j = 0; j < nel; ++j) r2[j] = r1[j] - alpha * Fp[j];
        for (int32_t j = 0; j < nel; ++j) P[j] += alpha * s2[j];
        for (int32_t j = 0; j < neq; ++j) V[j] -= alpha * u1[j];
        vdotv = uz_v_norm2(nno, dof, id, nmass, volume, V);
        pdotp = uz_p_norm2(nel, area, volume, P);
        dvelocity = alpha * sqrt(uz_v_norm2(nno, dof, id, nmass, volume, u1) / (1e-32 + vdotv));
        dpressure = alpha * sqrt(uz_p_norm2(nel, area, volume, s2) / (1e-32 + pdotp));
        oracle_assemble_div_u(nel, npe, dof, ien, id, g, V, z1);
        incompressibility = sqrt(uz_div_norm2(nel, area, volume, z1) / (1e-32 + vdotv));
        count++;
        if (hist) { double *h = hist + 5 * count; h[0] = sqrt(vdotv); h[1] = sqrt(pdotp); h[2] = dvelocity; h[3] = dpressure; h[4] = incompressibility; }
        if (!valid) converging = 0;
        else if (check_pressure_convergence) converging = (dvelocity < imp && dpressure < imp) ? converging + 1 : 0;
        else converging = (dvelocity < imp) ? converging + 1 : 0;
        { double *sh = s1; s1 = s2; s2 = sh; sh = r1; r1 = r2; r2 = sh; }
        r0dotz0 = r1dotz1;
    }
    *steps_max = count;
    free(F); free(u1); free(tmp); free(r1); free(r2); free(z1); free(s1); free(s2); free(Fp); free(rows);
    return incompressibility;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * CitcomS node-assembled stiffness operator: Node_map / Eqn_k1..3 — SURVEY.md §8 f1.
 * max_eqn = 14·dims slots per node: slot group 0 = the node's own 3 equations, groups 1..13 = lower-numbered neighbours;
 * unused slots hold the dummy equation `neq` (Construct_arrays.c:254-328). 0-based nodes, dims = dof = 3.
 * ---------------------------------------------------------------------------------------------------------------------------- */

/* construct_node_ks, citcoms/lib/Construct_arrays.c:335-470: Eqn_k from the element matrices, symmetric half (node1 <= node),
 * boundary dofs weighted out (bcw[node·3 + d] = 0 where NODE & VBX/VBZ/VBY is set, 1 otherwise). Returns -1 if a slot is missing
 * (the source asserts). */
ORACLE_API int oracle_construct_node_ks(int32_t nel, int32_t npe, const int32_t *ien, const int32_t *id, int32_t nno, int32_t neq,
                                        int32_t max_eqn, const int32_t *node_map, const double *elt_k, const double *bcw,
                                        double *k1, double *k2, double *k3)
{
    const int32_t dims = 3, lms = npe * dims;
    (void)neq;
    for (int64_t i = 0; i < (int64_t)nno * max_eqn; ++i) { k1[i] = 0.0; k2[i] = 0.0; k3[i] = 0.0; }
    for (int32_t element = 0; element < nel; ++element) {
        const double *elt_K = elt_k + (size_t)element * lms * lms;
        for (int32_t i = 0; i < npe; ++i) {                /* i: the node we are storing to */
            const int32_t node = ien[element * npe + i], pp = i * dims;
            const int64_t loc0 = (int64_t)node * max_eqn;
            const double w1 = bcw[node * 3 + 0], w2 = bcw[node * 3 + 1], w3 = bcw[node * 3 + 2];
            for (int32_t j = 0; j < npe; ++j) {            /* j: the node we are receiving from */
                const int32_t node1 = ien[element * npe + j];
                if (node1 > node) continue;                /* only half of the matrix, because of the symmetry */
                const int32_t qq = j * dims;
                for (int32_t dir = 0; dir < 3; ++dir) {
                    const int32_t eqn = id[node1 * 3 + dir];
                    const double ww = bcw[node1 * 3 + dir];
                    int32_t index = -1;
                    for (int32_t k = 0; k < max_eqn; ++k)
                        if (node_map[loc0 + k] == eqn) { index = k; break; }
                    if (index < 0) return -1;
                    k1[loc0 + index] += w1 * ww * elt_K[pp * lms + qq + dir];
                    k2[loc0 + index] += w2 * ww * elt_K[(pp + 1) * lms + qq + dir];
                    k3[loc0 + index] += w3 * ww * elt_K[(pp + 2) * lms + qq + dir];
                }
            }
        }
    }
    return 0;
}

/* n_assemble_del2_u, citcoms/lib/Element_calculations.c:516-577: Au = K·u from the stored half, node by node; u and Au have
 * neq + 1 entries (index neq is the dummy equation). strip_bcs: zero the listed rows afterwards. */
ORACLE_API void oracle_n_assemble_del2_u(int32_t nno, int32_t neq, int32_t max_eqn, const int32_t *node_map, const int32_t *id,
                                         const double *k1, const double *k2, const double *k3, double *u /* [neq+1], u[neq] := 0 */,
                                         double *Au /* [neq+1] */, const int32_t *zero_resid, int32_t n_zero)
{
    for (int32_t e = 0; e <= neq; ++e) Au[e] = 0.0;
    u[neq] = 0.0;
    for (int32_t e = 0; e < nno; ++e) {
        const int32_t eqn1 = id[e * 3], eqn2 = id[e * 3 + 1], eqn3 = id[e * 3 + 2];
        const double U1 = u[eqn1], U2 = u[eqn2], U3 = u[eqn3];
        const int32_t *C = node_map + (int64_t)e * max_eqn;
        const double *B1 = k1 + (int64_t)e * max_eqn, *B2 = k2 + (int64_t)e * max_eqn, *B3 = k3 + (int64_t)e * max_eqn;
        for (int32_t i = 3; i < max_eqn; ++i) {
            const double UU = u[C[i]];
            Au[eqn1] += B1[i] * UU;
            Au[eqn2] += B2[i] * UU;
            Au[eqn3] += B3[i] * UU;
        }
        for (int32_t i = 0; i < max_eqn; ++i) Au[C[i]] += B1[i] * U1 + B2[i] * U2 + B3[i] * U3;
    }
    for (int32_t i = 0; i < n_zero; ++i) Au[zero_resid[i]] = 0.0;
}
