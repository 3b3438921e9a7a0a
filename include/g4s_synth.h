/*
 * g4s_synth.h — device-side generators of the synthetic inputs of SURVEY.md §8d (R-MAT power-law, 5-/7-point
 * Laplacians, banded). They exist so that bench.py and the full-size tests can build 1e8-nonzero matrices
 * directly in HBM; they are not part of the drop-in boundary (the reference reads MatrixMarket files that are
 * not shipped: mm/src/mkl_spgemm.cpp:18-37). Bit-for-bit twins of the oracle's generators
 * (oracle/g4s_oracle.c: oracle_rmat_edges, oracle_entry_value, oracle_vector_value, oracle_laplacian5/7, oracle_banded).
 * All pointers are device pointers; all calls are asynchronous on `stream`.
 */
#ifndef G4S_SYNTH_H
#define G4S_SYNTH_H
#include "g4s.h"
#ifdef __cplusplus
extern "C" {
#endif

/* keys[q] = row·n + col of R-MAT edge e0+q, (a,b,c,d) = (0.57,0.19,0.19,0.05), ids >= n redrawn. */
g4s_status g4s_synth_rmat_keys(uint64_t seed, int32_t scale, int64_t n, int64_t e0, int64_t count,
                               int64_t *keys_dev, void *stream);
/* From sorted, duplicate-free keys: colids[k] = key % n, values[k] = U(−1,1) of (seed,row,col), rowptr[0..rows]. */
g4s_status g4s_synth_csr_from_keys(uint64_t seed, int64_t n, int32_t rows, const int64_t *keys_dev, int64_t nnz,
                                   int32_t *rowptr_dev, int32_t *colids_dev, double *values_dev, void *stream);
/* x[i] = U(−1,1) of (seed, i0+i) for i in [0,count). */
g4s_status g4s_synth_vector(uint64_t seed, int64_t i0, int64_t count, double *x_dev, void *stream);

/* Rows [r0,r1) of the 7-point Laplacian on nx×ny×nz (diag 6, off −1), global column ids; nz==1 and diag 4
 * give the 5-point stencil (kind: 5 or 7). counts_dev[r-r0] = row length (pass 1, fill==0);
 * with fill!=0, rowptr_dev (local, r1-r0+1 entries, already scanned) is read and colids/values written. */
g4s_status g4s_synth_laplacian_rows(int32_t kind, int32_t nx, int32_t ny, int32_t nz, int64_t r0, int64_t r1,
                                    int32_t *counts_dev, const int32_t *rowptr_dev, int32_t *colids_dev,
                                    double *values_dev, int32_t fill, void *stream);
/* Banded matrix, half bandwidth hb, values U(−1,1) of (seed,row,col); rowptr/colids/values sized by the caller
 * (nnz = Σ_i (min(n−1,i+hb) − max(0,i−hb) + 1)). */
g4s_status g4s_synth_banded(int32_t n, int32_t hb, uint64_t seed, int32_t *rowptr_dev, int32_t *colids_dev,
                            double *values_dev, void *stream);

/* The library's own device-wide primitives (csrc/prims.hpp: exclusive prefix sum; stable key-value radix sort, descending keys, 4-bit
 * digits), exposed so that they can be tested on their own. out[i] = sum of in[0..i); keys are non-negative ints with at most key_bits
 * significant bits; inputs are left untouched. Device pointers, asynchronous on `stream`. */
g4s_status g4s_prim_exclusive_scan_i32(const int32_t *in_dev, int32_t *out_dev, int64_t n, void *stream);
g4s_status g4s_prim_exclusive_scan_i64(const int64_t *in_dev, int64_t *out_dev, int64_t n, void *stream);
g4s_status g4s_prim_sort_pairs_desc_i32(const int32_t *keys_in_dev, const int32_t *vals_in_dev, int32_t *keys_out_dev, int32_t *vals_out_dev,
                                        int32_t n, int32_t key_bits, void *stream);

#ifdef __cplusplus
}
#endif
#endif
