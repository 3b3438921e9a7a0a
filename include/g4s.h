/*
 * g4s.h — C-ABI of libg4s_hip.so: the MI355X (gfx950) drop-in for G4S's graph-as-sparse-matrix hot path.
 *
 * Plain C types only, no exceptions cross this boundary, every entry point returns a g4s_status
 * (0 = success, negative = error; g4s_last_error() gives the message of the calling thread's last failure).
 *
 * Each entry point names the reference interface (file:line under the G4S tree) it replaces:
 *
 *   B1  raw-pointer CSR SpGEMM     mm/inc/mkl_mult.h:40-43      void mkl(arpt,acol,aval, brpt,bcol,bval, &crpt,&ccol,&cval, M,K,N,&cnnz, Timings&)
 *                                   mm/inc/hash_mult.h:1028-1057 HashSpGEMM<vectorProbing,sortOutput>(a,b,c,multop,addop)
 *   B2  CSR SpMV                   mv/mv.c:6-27                 void matrix_multiply_*(double*A,double*B,double*C,int dim)  (y = A·x; the
 *                                                                reference ships only dense BLAS-2 forms, the CSR form is defined in DESIGN.md)
 *   B3  graph gather/apply         citcoms/lib/global_defs.h:48-49,854-857  spmm_dense(numNodes,degree,edgeWeight,vertexStates,temp,result,gather,apply,time,threadNum)
 *                                   deepmd/source/op/graph.h:5-32            struct Graph, GraphProcess(graph,result,gather,apply)
 *                                   cantera/src/thermo/RedlichKwongMFTP.cpp:917-983  GraphProcess1/2
 *
 * Index type is int32 and value type fp64, as in the reference (mm/inc/define.h:14-15).
 * The library fails loudly (G4S_ERR_HIP) when no HIP device is usable; there is no CPU fallback in it.
 */
#ifndef G4S_H
#define G4S_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
typedef int g4s_status;
#define G4S_OK               0
#define G4S_ERR_INVALID     -1 /* bad argument (null pointer, negative size, unsorted rowptr …)           */
#define G4S_ERR_NOMEM       -2 /* host or device allocation failed                                        */
#define G4S_ERR_HIP         -3 /* a HIP runtime call failed / no device                                   */
#define G4S_ERR_OVERFLOW    -4 /* a result does not fit the reference's int32 index type (define.h:14)    */
#define G4S_ERR_UNSUPPORTED -5 /* e.g. an unregistered gather/apply pair handed to the device dispatcher  */

/* ------------------------------------------------------------------ flags */
#define G4S_HOST_POINTERS    0u /* array arguments are host memory (default; H2D/D2H happen inside)       */
#define G4S_DEVICE_POINTERS  1u /* array arguments are device memory, borrowed for the handle's lifetime  */
#define G4S_SORT_OUTPUT      2u /* SpGEMM: rows of C sorted by column (HashSpGEMM sortOutput, hash_mult.h:526-553) */
#define G4S_SPMV_NO_NT       4u /* SpMV: plain (cache-allocating) loads for the matrix stream instead of nontemporal */
#define G4S_SPMV_BLOCKED     8u /* SpMV: force the propagation-blocked path (x band / y band in LDS; for matrices without gather locality) */
#define G4S_SPMV_STREAM     16u /* SpMV: force the row-streaming CSR kernel (default: chosen per matrix — blocked for matrices without gather locality,
                                 * the index-free diagonal form for stencil / banded matrices, the CSR kernel otherwise) */

#define G4S_SPMV_UPDATABLE 128u /* SpMV: the values of this matrix will be replaced (g4s_csr_update_values): a plan that keeps them in another order also keeps the
                                 * map back to the CSR order (blocked path: 4 bytes per entry), so that an update is one pass instead of a new plan */

/* ------------------------------------------------------------------ runtime */
const char *g4s_version(void);
/* "spmv_kernel_sources_sha256=<hex>;variant=<name>": the SpMV kernel sources this library was built from (tools/kernel_hash.py) and the name of an A/B
 * variant build (empty for the regular one) — for tools that pair a stored measurement with the LOADED build (bench.py's roofline.traffic). */
const char *g4s_build_info(void);
/* Loads the device code of the plan builders, the SpMV kernels and the SpGEMM row classes now instead of inside the first g4s_csr_create / g4s_spmv /
 * g4s_spgemm_* of the process (HIP loads a translation
 * unit's code object at its first launch: 12–25 ms of a first create of configs[1] that takes 18.6 ms afterwards). Builds, runs and destroys one small matrix on
 * the path g4s_csr_create picks for it, on the blocked and on the streaming path, and squares it. Optional; synchronous; may be called again (≈ 25 ms then, 0.2 s in a
 * process that has not touched the device yet).
 * No reference counterpart (a CPU library has no such step). */
g4s_status g4s_warm_up(void);
const char *g4s_last_error(void);                 /* thread-local, never NULL                              */
g4s_status  g4s_device_count(int *count);
g4s_status  g4s_set_device(int device);           /* also honours HIP_VISIBLE_DEVICES                      */
g4s_status  g4s_device_synchronize(void);
g4s_status  g4s_shutdown(void);                   /* releases cached workspaces                            */
g4s_status  g4s_trim(void);                       /* returns the library's cached device blocks (freed SpGEMM outputs, column scratch) to the
                                                    * driver; for host frameworks whose own allocator is about to need the memory            */

/* Allocator that matches every callee-allocated output of this library (the reference pairs
 * my_malloc/my_free, mm/inc/utility.h:126-153; CSR::make_empty frees what mkl()/HashSpGEMM allocated,
 * mm/inc/CSR.h:50-62). */
void       *g4s_malloc(size_t bytes);
void        g4s_free(void *p);

/* Device buffers for callers without their own HIP code (the bench and the tests use torch tensors instead). */
/* Device blocks of the library's caching allocator (a freed block is kept and handed out again for a request it fits within 25 %:
 * fresh multi-GB allocations cost up to seconds on this stack; g4s_shutdown releases what is cached). g4s_dev_free synchronises the
 * device like hipFree, and also accepts a plain hipMalloc pointer. */
g4s_status  g4s_dev_alloc(void **dptr, size_t bytes);
g4s_status  g4s_dev_free(void *dptr);
g4s_status  g4s_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
g4s_status  g4s_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);

/* ------------------------------------------------------------------ B2: CSR SpMV  y = alpha·A·x + beta·y */

/* Opaque device-resident CSR matrix + its SpMV execution plan (row blocks, long-row chunks). */
typedef struct g4s_csr_s *g4s_csr_t;

typedef struct g4s_csr_info {
    int32_t rows, cols;
    int64_t nnz;
    int32_t stream_blocks;    /* row-aligned blocks of <= tile_nnz nonzeros handled by the LDS-staged path */
    int32_t long_rows;        /* rows longer than tile_nnz, split into chunks                             */
    int32_t long_chunks;
    int32_t tile_nnz, tile_rows, long_chunk_nnz;
    int64_t algorithmic_bytes;/* 12·nnz + 4·(rows+1) + 8·rows + 8·cols                      */
    int64_t plan_bytes;       /* extra device bytes the plan itself occupies                               */
    int32_t spmv_path;        /* 0 = row-streaming CSR kernel, 1 = propagation-blocked (regrouped copy of the matrix), 2 = unused (was round 2's tile-blocked experiment, removed),
                               * 3 = diagonal-structured, index-free (stencil / banded matrices: values by diagonal + a presence mask per row),
                               * 4 = block-row (assembled FE matrices: aligned b×b blocks, one block-column id per block, one lane per row) */
    int32_t reserved;
} g4s_csr_info;

/* Create a handle. rowptr has rows+1 entries, zero-based, non-decreasing, rowptr[rows] == nnz; colids in [0,cols).
 * With G4S_HOST_POINTERS the three arrays are copied to the device (the handle owns the copies);
 * with G4S_DEVICE_POINTERS they are borrowed and must outlive the handle. Replaces the container role of
 * CSR<int,double> (mm/inc/CSR.h:22-113) for device residency.
 * The handle is a SNAPSHOT of the matrix: the execution plan (and, on three of the four paths, a copy of the values in the plan's own order) is
 * built here; the PATTERN is fixed for the handle's life, new VALUES go in through g4s_csr_update_values.
 * Concurrency: a handle supports ONE product in flight at a time — g4s_spmv uses per-handle workspaces (the partial sums of split long
 * rows on the streaming path, the product buffer and the gathered hot columns on the blocked path), so two g4s_spmv calls on the same
 * handle must be ordered (same stream, or an event between them); different handles are independent.
 * Stream order of create: g4s_csr_create reads the arrays and builds the plan on the NULL (legacy default) stream and returns after it
 * has synchronised; with G4S_DEVICE_POINTERS the arrays must be complete with respect to that stream — a caller that filled them on a
 * non-blocking stream synchronises it first. */
g4s_status g4s_csr_create(g4s_csr_t *out, int32_t rows, int32_t cols, int64_t nnz,
                          const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags);
/* New values for the same pattern — what a time-stepping caller does to its operator (CitcomS rebuilds the stiffness matrix before every Stokes solve and
 * inside the viscosity iteration: citcoms/lib/Drive_solvers.c:88,134 → construct_stiffness_B_matrix, Construct_arrays.c:740). values: nnz entries in the
 * order of the arrays the handle was created from (flags: G4S_HOST_POINTERS or G4S_DEVICE_POINTERS), or NULL when a BORROWED device array has been rewritten
 * in place. A handle that owns its arrays copies them in; a handle that borrows device arrays borrows the new array from here on (device pointer required).
 * The plan's own copy is refreshed on `stream` (asynchronous like g4s_spmv, ordered with the products on that stream): one gather pass on the blocked path
 * (created with G4S_SPMV_UPDATABLE; without the flag the regrouping is repeated — the cost of a create), a refill of the diagonals / the block-major copy,
 * nothing on the row-streaming path. Results afterwards are those of a handle freshly created from the new values, bit for bit. */
g4s_status g4s_csr_update_values(g4s_csr_t A, const double *values, unsigned flags, void *stream);
g4s_status g4s_csr_destroy(g4s_csr_t A);
g4s_status g4s_csr_get_info(g4s_csr_t A, g4s_csr_info *info);
/* Device pointers of the handle's arrays (borrowed). */
g4s_status g4s_csr_device_arrays(g4s_csr_t A, const int32_t **rowptr, const int32_t **colids, const double **values);

/* Asynchronous SpMV on `stream` (a hipStream_t; NULL = the default stream). x_dev (cols doubles) and
 * y_dev (rows doubles) are device pointers and must not alias. beta == 0 never reads y (BLAS convention).
 * The call only enqueues kernels on `stream` — no allocation, no synchronisation, no host read — so a sequence of products may be recorded
 * in a hipGraph (stream capture) and replayed (tests/test_graph_capture_gpu.py, all three paths). */
g4s_status g4s_spmv(g4s_csr_t A, const double *x_dev, double *y_dev, double alpha, double beta, void *stream);

/* One-shot form with the call shape of mv/mv.c:6-27 (caller-owned in/out arrays, synchronous).
 * flags: G4S_HOST_POINTERS or G4S_DEVICE_POINTERS applies to all five arrays. */
g4s_status g4s_spmv_csr_i32_f64(int32_t rows, int32_t cols, const int32_t *rowptr, const int32_t *colids,
                                const double *values, const double *x, double *y,
                                double alpha, double beta, unsigned flags);

/* ---- B2 on several GPUs: the 1-D row-partitioned product (SURVEY.md §8b "g4s_spmv_dist_*", §8e). One process per GPU. Rank r owns rows
 * [row_offsets[r], row_offsets[r+1]) of a square operator and the same slab of x and y; the local rows are given as CSR with GLOBAL column
 * ids. The reference has no multi-device code on this path; the pattern replaced is CitcomS's per-mat-vec neighbour exchange
 * (citcoms/lib/Regional_parallel_related.c:744-789) with the equal-work row split of mm/inc/BIN.h:101-122. Per product only the x entries
 * a rank's rows actually reference travel (halo planes for stencils), each peer pair over its own xGMI link (ncclSend/ncclRecv in one
 * group), while the own-column part of the product runs. */
#define G4S_DIST_LOOPBACK 32u   /* single-rank rehearsal: half of the own slab is treated as remote and travels rank 0 → rank 0 through RCCL, cut into
                                 * G4S_DIST_LOOPBACK_PEERS (default 7) segments with their own ncclSend / ncclRecv pair each — the message pattern
                                 * one rank of an 8-GPU node has, on one GPU */
#define G4S_DIST_ALLGATHER 64u  /* exchange = ONE in-place ncclAllGather of the whole vector (every slab padded to the longest; north_star's
                                 * "RCCL all-gather of the dense vector") instead of packed point-to-point messages: more bytes, no index
                                 * lists, nothing to wire. Also selected by G4S_DIST_EXCHANGE=allgather in the environment. */

/* Equal-work contiguous row partition — BIN::set_rows_offset, mm/inc/BIN.h:101-122: prefix-sum the per-row work, average share
 * avg = ceil(total / parts), boundary t = lower_bound(prefix, avg·t), last boundary = rows. The reference splits rows over threads (work =
 * flop per row); the same rule gives g4s_spmv_dist_create its row_offsets. work_i = row_work[i] when row_work is given (host array, rows
 * entries — any cost model the caller has), else (rowptr[i+1] − rowptr[i]) + row_weight (bench.py: row_weight = 1). row_offsets: parts+1
 * entries, host memory. flags: G4S_HOST_POINTERS / G4S_DEVICE_POINTERS for rowptr. Host-side set-up logic: runs without a GPU. */
g4s_status g4s_row_partition(int32_t rows, const int32_t *rowptr, const int64_t *row_work, int64_t row_weight, int32_t parts,
                             int64_t *row_offsets, unsigned flags);

/* One rank's rows cut into the own-column part (columns renumbered to the own slab) and the remote-column part (columns renumbered into
 * the remote x: packed mode — the referenced columns only, ascending, segment [recv_cut[k], recv_cut[k+1]) from owner k, `want` = their
 * indices local to the owner's slab; all-gather mode — k·pad + (c − row_offsets[k])). What g4s_spmv_dist_create does before it uploads;
 * exported for hosts that bring their own transport or device set-up. Host arrays in, g4s_malloc'ed host arrays out (g4s_dist_split_free);
 * needs no GPU. flags: G4S_DIST_ALLGATHER, G4S_DIST_LOOPBACK. merged = 1: fewer than a quarter of the entries sit in own columns and ALL
 * columns went to the remote part (one product per step). */
typedef struct g4s_dist_split {
    int32_t local_rows, n_ref, merged, allgather;
    int64_t nnz_own, nnz_rem, pad;
    int32_t *own_rowptr, *own_colids; double *own_values;
    int32_t *rem_rowptr, *rem_colids; double *rem_values;
    int32_t *want;            /* packed mode: n_ref entries */
    int64_t *recv_cut;        /* segments + 1 entries (segments = world; in loopback mode the rehearsed peer count) */
} g4s_dist_split;
g4s_status g4s_dist_split_rows(int32_t rank, int32_t world, const int64_t *row_offsets, int64_t n_cols,
                               const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags, g4s_dist_split *out);
void g4s_dist_split_free(g4s_dist_split *s);

typedef struct g4s_spmv_dist_s *g4s_spmv_dist_t;
typedef struct g4s_spmv_dist_info {
    int32_t rank, world, local_rows, n_ref;     /* n_ref: distinct remote columns this rank's rows reference (length of the compact remote x) */
    int64_t nnz_own, nnz_rem;                   /* nonzeros in own / remote columns */
    int64_t send_bytes, recv_bytes;             /* per product */
    int32_t own_path, rem_path;                 /* g4s_csr_info.spmv_path of the two parts */
    int32_t connected, reserved;                /* connected: every peer's give list is known; reserved: bit 0 = merged form (own columns are few and live in
                                                 * the remote x with the remote ones: one product per step instead of two), bit 1 = all-gather exchange,
                                                 * bit 2 = poisoned by a failed set-up exchange (see g4s_spmv_dist_apply) */
} g4s_spmv_dist_info;
/* flags: G4S_HOST_POINTERS / G4S_DEVICE_POINTERS for the three matrix arrays, the G4S_SPMV_* path flags, G4S_DIST_LOOPBACK, G4S_DIST_ALLGATHER.
 * row_offsets: world+1 entries, host memory; row_offsets[world] == n_cols. Collective only in the sense that every rank creates its own. */
g4s_status g4s_spmv_dist_create(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *row_offsets, int64_t n_cols,
                                const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags);
/* Rectangular operator: rows partitioned by row_offsets (y), columns by col_offsets (x) — the discrete divergence / gradient of the Stokes
 * iteration (elements × equations and back, citcoms/lib/Element_calculations.c:701-779) next to the square stiffness operator. */
g4s_status g4s_spmv_dist_create_rect(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *row_offsets, const int64_t *col_offsets,
                                     const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags);
g4s_status g4s_spmv_dist_destroy(g4s_spmv_dist_t h);
g4s_status g4s_spmv_dist_get_info(g4s_spmv_dist_t h, g4s_spmv_dist_info *info);
/* Wiring, RCCL: comm is an ncclComm_t over the same ranks (the caller's own, or g4s_comm_create below); the want / give index lists
 * are exchanged once with ncclSend / ncclRecv. Collective over the communicator. */
g4s_status g4s_spmv_dist_connect_rccl(g4s_spmv_dist_t h, void *comm);
/* Wiring, any other transport (MPI, gloo): idx_dev[0..count) are the entries this rank wants from `peer` (indices local to the peer's
 * slab, device memory); the caller carries every list to its owner and hands it over with _set_give (flags: host or device pointer). */
g4s_status g4s_spmv_dist_want(g4s_spmv_dist_t h, int32_t peer, int64_t *count, const int32_t **idx_dev);
g4s_status g4s_spmv_dist_set_give(g4s_spmv_dist_t h, int32_t peer, int64_t count, const int32_t *idx, unsigned flags);
/* y_local = (A·x)[own rows]. x_local_dev / y_local_dev: this rank's slabs, device memory (y may be NULL on a rank that owns no rows, x on a
 * rank that owns no x entries — a rectangular operator; such a rank still posts its part of the exchange). Asynchronous on `stream`; RCCL
 * traffic runs on a stream of the handle. Needs g4s_spmv_dist_connect_rccl when world > 1.
 * Failure of the set-up exchange: g4s_spmv_dist_connect_rccl waits with a deadline (G4S_DIST_TIMEOUT_S, default 120 s) and watches RCCL's
 * asynchronous error state. When either fires — a peer never entered the exchange — the communicator is ABORTED (ncclCommAbort) and the handle
 * is poisoned: later calls on it fail, g4s_spmv_dist_destroy drops only its host side (the device memory of the stuck operation is left to the
 * process's exit, nothing is synchronised). The caller reports the error and exits non-zero; a supervisor starts a fresh process. */
g4s_status g4s_spmv_dist_apply(g4s_spmv_dist_t h, const double *x_local_dev, double *y_local_dev, void *stream);
/* The same in two halves for callers with their own transport: _begin packs the send buffer and starts y = A_own·x_local; the caller moves
 * send_dev[send_cut[k]..send_cut[k+1]) to peer k and receives peer k's entries into recv_dev[recv_cut[k]..recv_cut[k+1]) (both ordered after
 * _begin on `stream`); _finish adds the remote-column part. */
g4s_status g4s_spmv_dist_begin(g4s_spmv_dist_t h, const double *x_local_dev, double *y_local_dev, void *stream);
/* All-gather mode: send_dev = this rank's slot (pad entries) and goes to EVERY peer (send_cut = {0, …, 0, pad}); recv_dev = the gathered vector,
 * slot k = [k·pad, (k+1)·pad) comes from rank k. */
g4s_status g4s_spmv_dist_buffers(g4s_spmv_dist_t h, double **send_dev, const int64_t **send_cut, double **recv_dev, const int64_t **recv_cut);
g4s_status g4s_spmv_dist_finish(g4s_spmv_dist_t h, double *y_local_dev, void *stream);
/* Column-partition variant — a CORRECTNESS variant (north_star's "all-reduce of partial products"): rank g holds the columns
 * [col_offsets[g], col_offsets[g+1]) of A (all n_rows rows, global column ids inside the slab) and that slab of x; _begin forms the partial y of ALL rows
 * (y_local_dev: n_rows entries), _finish / _apply sum it over the ranks with ncclAllReduce on the communicator of g4s_spmv_dist_connect_rccl — every rank ends
 * with the whole y. With another transport the caller all-reduces y itself between _begin and _finish (_buffers has nothing to hand out). Twice the traffic of
 * the all-gather exchange and every rank reduces all of y: the row partition is the product's form. get_info.reserved bit 3. */
g4s_status g4s_spmv_dist_create_columns(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *col_offsets, int32_t n_rows,
                                        const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags);
/* New values for this rank's rows, same pattern (see g4s_csr_update_values): values_local_dev in the order of the arrays given to _create, device memory;
 * the handle must have been created with G4S_SPMV_UPDATABLE. Asynchronous on `stream`; no communication. */
g4s_status g4s_spmv_dist_update_values(g4s_spmv_dist_t h, const double *values_local_dev, unsigned flags, void *stream);
/* RCCL communicator for hosts that have none (rank 0 makes the 128-byte id, every rank gets it by its own means and calls _create), and
 * the sum all-reduce the dot products of a Krylov solver need (citcoms/lib/Global_operations.c:534-562). RCCL is dlopen'ed at first use. */
g4s_status g4s_comm_unique_id(void *id128);
g4s_status g4s_comm_create(void **comm, int32_t world, int32_t rank, const void *id128);
g4s_status g4s_comm_destroy(void *comm);
g4s_status g4s_comm_allreduce_sum_f64(void *comm, double *buf_dev, int64_t count, void *stream);

/* ------------------------------------------------------------------ B1: CSR SpGEMM  C = A·B */

/* Stage timer with the reference's seven fields, milliseconds (mm/inc/Timings.h:4-23). */
typedef struct g4s_timings {
    double create, spmm, convert, order, export_csr, destroy, total;
} g4s_timings;

/* flop = Σ_i Σ_{j∈A(i,:)} nnz(B(acol_j,:))  (mm/inc/mkl_mult.h:8-38, hash_mult.h:46-62); device pointers
 * when flags has G4S_DEVICE_POINTERS. row_flop (may be NULL) receives the per-row count as int64. */
g4s_status g4s_spgemm_flop(int32_t M, const int32_t *arpt, const int32_t *acol, const int32_t *brpt,
                           int64_t *flop, int64_t *row_flop, unsigned flags);

/* Input contract of the three SpGEMM entry points: zero-based CSR; the rows of A and of B may be in any order, as for HashSpGEMM (its hash traversal never
 * looks at the order, mm/inc/hash_mult.h:579-600, and HashSpGEMM<..., false> emits unsorted rows itself, :530-551) and mkl_sparse_spmm (mm/inc/mkl_mult.h:58);
 * repeated columns inside a row are allowed and are added up. The kernels cut B's rows at column boundaries, so a B whose rows are NOT sorted by column (checked
 * on the opening pass of every call) is sorted into a private copy first — the caller's arrays are never modified; sorted input (what CSR::construct produces,
 * mm/inc/CSR.h:640-651) is the fast path. Column ids are range-checked (G4S_ERR_INVALID).
 * Raw-pointer SpGEMM with the call shape of mkl(...) (mm/inc/mkl_mult.h:40-43): inputs borrowed,
 * outputs allocated by the callee — with g4s_malloc for host pointers (free with g4s_free), with
 * g4s_dev_alloc for G4S_DEVICE_POINTERS (free with g4s_dev_free). A is M×K, B is K×N, C is M×N.
 * cnnz is int64; G4S_ERR_OVERFLOW is returned (and nothing allocated) if it exceeds INT32_MAX,
 * because crpt keeps the reference's int32 type. timings may be NULL. */
g4s_status g4s_spgemm_csr_i32_f64(const int32_t *arpt, const int32_t *acol, const double *aval,
                                  const int32_t *brpt, const int32_t *bcol, const double *bval,
                                  int32_t **crpt, int32_t **ccol, double **cval,
                                  int32_t M, int32_t K, int32_t N, int64_t *cnnz,
                                  g4s_timings *timings, unsigned flags);

/* Two-phase form (hash_symbolic / hash_numeric, mm/inc/hash_mult.h:496-508,559-608) on device pointers:
 * symbolic writes crpt_dev[M+1] (int32) and *cnnz; numeric fills ccol_dev/cval_dev (cnnz entries each).
 * The symbolic call keeps what it learned about the product (the sorted columns of the long rows, the column map of B, B's window splits) for the numeric
 * call that follows it with THE SAME arrays (same pointers, unchanged contents — crpt describes this product and no other): that call then does not traverse
 * the structure again (the pair costs what the one-call form costs), nor do further numeric calls on the same arrays (new VALUES of A or B, same pattern: the
 * time-stepping case). One product at a time per process: the next symbolic or one-call product, g4s_trim
 * and g4s_shutdown release whatever is still held. The state is keyed by the pointers AND by a hash of the five index arrays (arpt, acol, brpt, bcol, crpt)
 * taken at the end of the symbolic call and checked at the start of every numeric call that would use it: a numeric call with other arrays, with the same
 * buffers refilled by another pattern, with its own crpt, or without a symbolic call before it derives everything it needs itself (no carried state is used). */
g4s_status g4s_spgemm_symbolic(int32_t M, int32_t K, int32_t N,
                               const int32_t *arpt_dev, const int32_t *acol_dev,
                               const int32_t *brpt_dev, const int32_t *bcol_dev,
                               int32_t *crpt_dev, int64_t *cnnz, void *stream);
g4s_status g4s_spgemm_numeric(int32_t M, int32_t K, int32_t N,
                              const int32_t *arpt_dev, const int32_t *acol_dev, const double *aval_dev,
                              const int32_t *brpt_dev, const int32_t *bcol_dev, const double *bval_dev,
                              const int32_t *crpt_dev, int32_t *ccol_dev, double *cval_dev,
                              unsigned flags, void *stream);

/* ------------------------------------------------------------------ B3: graph gather/apply */

typedef void (*fun_gather)(int, int, const double **, const double *, double *); /* citcoms/lib/global_defs.h:48 */
typedef void (*fun_apply)(int, const double **, const double *, double *);       /* citcoms/lib/global_defs.h:49 */

/* Known gather/apply patterns the device can execute (host callbacks cannot run on the GPU).            */
#define G4S_PATTERN_ELEMENT_BLOCK_MATVEC   1 /* CitcomS e_assemble_del2_u gather, Element_calculations.c:453-471 */
#define G4S_PATTERN_DENSE_ROW_TIMES_MATRIX 2 /* DeePMD OptMatmul gather, opt_matmul.cc:52-58                     */
#define G4S_PATTERN_SYM_QUADRATIC_FORM     3 /* Cantera gather1/apply1, gather2/apply2, RedlichKwongMFTP.cpp:927-970 */

typedef struct g4s_pattern_desc {
    int32_t kind;
    /* ELEMENT_BLOCK_MATVEC: result[eq(e,a,i)] += Σ_b Σ_d K_e[(dof·a+i)·(npe·dof) + dof·b+d] · u[eq(e,b,d)],
     * eq(e,a,i) = id[ ien[e·npe + a]·dof + i ]   (0-based restatement of IEN/ID, Element_calculations.c:460-468). */
    int32_t num_elems;        /* nel: number of elements (= numNodes of the spmm_dense calls that follow)                */
    int32_t nodes_per_elem;   /* 8  (enodes[3])                                                          */
    int32_t dof;              /* 3  (mesh.nsd)                                                           */
    const int32_t *ien;       /* [numNodes · nodes_per_elem] element → node, 0-based, host memory         */
    const int32_t *id;        /* [nno · dof] node,dof → equation, 0-based, host memory                    */
    int32_t nno;              /* number of nodes                                                         */
    int32_t neq;              /* number of equations (length of vertexStates / result)                   */
    int32_t edge_weight_base; /* 1 if edgeWeight[0] is unused and element e lives at edgeWeight[e+1] (CitcomS, Drive_solvers.c:52-55), else 0 */
    int32_t static_weights;   /* 1: the element matrices behind an edgeWeight pointer do not change between calls (true inside one
                                 CitcomS CG solve); they are uploaded once per distinct edgeWeight pointer. Re-register to invalidate. */
    /* DENSE_ROW_TIMES_MATRIX: result[e·degree + a] = Σ_k edgeWeight[e][k] · states[k·degree + a]        */
    int32_t inner;            /* N (opt_matmul.cc:33, global Nsize)                                      */
    /* SYM_QUADRATIC_FORM: see RedlichKwongMFTP.cpp:927-970                                              */
    int32_t numbers;          /* coefficient stride (1 = gather1/apply1 form, >1 = gather2/apply2 form)   */
} g4s_pattern_desc;

/* Register a (gather, apply) pair as an instance of a known pattern; later spmm_dense calls with that
 * pair run on the device. desc (and the ien/id arrays) are copied. */
g4s_status g4s_register_pattern(fun_gather gather, fun_apply apply, const g4s_pattern_desc *desc);
g4s_status g4s_unregister_pattern(fun_gather gather, fun_apply apply);

/* What an UNREGISTERED (gather, apply) pair gets. Callbacks are host code the device cannot execute, and the interface
 * promises every pair "gather degree times per vertex, then apply" (deepmd/source/op/graph.h:21-32), so by default the
 * reference's driver loop runs on the host, one thread, vertices in ascending order (the reference itself hard-codes
 * 8 OpenMP threads, graph.h:23, racing on gathers that scatter — e.g. CitcomS' Au[aa] +=, Element_calculations.c:466).
 *   SERIAL    (default) one host thread;
 *   PARALLEL  the caller declares its gathers race-free: threadNum threads, vertices handed out one at a time
 *             (schedule(dynamic,1), graph.h:24);
 *   REFUSE    G4S_ERR_UNSUPPORTED (spmm_dense: abort) — for deployments that must never fall off the device. */
#define G4S_HOST_CALLBACKS_SERIAL   0
#define G4S_HOST_CALLBACKS_PARALLEL 1
#define G4S_HOST_CALLBACKS_REFUSE   2
g4s_status g4s_set_host_callback_policy(int32_t policy);
/* The same for the calling thread only (-1: back to the process-wide policy); *previous (may be NULL) receives what was set before, so that scopes nest
 * (g4s::ScopedRaceFree, include/g4s/graph.hpp: a call site declares ITS gathers race-free without changing what other threads' callbacks get). */
g4s_status g4s_set_host_callback_policy_thread(int32_t policy, int32_t *previous);

/* The reference symbol, exactly (citcoms/lib/global_defs.h:854-857; bound at citcoms/bin/Citcom.c:93).
 * Registered pairs run as HIP kernels; any other pair runs the reference's host loop (policy above). spmm_dense returns
 * void as the reference's does, so a failure aborts with a message — g4s_spmm_dense is the status-returning form. */
void spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight, const double *vertexStates,
                double *temp, double *result, fun_gather gather, fun_apply apply, double *time, int threadNum);
g4s_status g4s_spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight, const double *vertexStates,
                          double *temp, double *result, fun_gather gather, fun_apply apply, double *time, int threadNum);

/* Device-resident forms of the three patterns (what spmm_dense dispatches to; solvers call these directly
 * to keep vectors on the device between iterations). */
typedef struct g4s_elem_op_s *g4s_elem_op_t;
typedef struct g4s_node_op_s *g4s_node_op_t;
typedef struct g4s_cg_ws_s *g4s_cg_ws_t;
/* elt_k_dev: numElems × (npe·dof)² doubles, contiguous, device memory (borrowed). ien/id host arrays as in the descriptor. */
g4s_status g4s_elem_op_create(g4s_elem_op_t *out, int32_t numElems, int32_t nodes_per_elem, int32_t dof,
                              const int32_t *ien_host, const int32_t *id_host, int32_t nno, int32_t neq,
                              const double *elt_k_dev);
g4s_status g4s_elem_op_destroy(g4s_elem_op_t op);
/* Au_dev[0..neq) = Σ_e scatter(K_e · gather(u_dev))  — overwrites Au (the caller's zeroing at
 * Element_calculations.c:495-496 is folded in). Deterministic (no atomics). */
g4s_status g4s_elem_op_apply(g4s_elem_op_t op, const double *u_dev, double *Au_dev, void *stream);

/* BI_dev[eq] = 1 / Σ_e K_e[p·n+p] over the (element, local dof p) pairs that map to eq — build_diagonal_of_K
 * (citcoms/lib/Element_calculations.c:580-611) followed by the inversion at Construct_arrays.c:469. Equations with a zero
 * diagonal get 0 (the reference asserts instead). */
g4s_status g4s_elem_op_inverse_diagonal(g4s_elem_op_t op, double *BI_dev, void *stream);

/* Device-resident Jacobi-preconditioned conjugate gradient with the update order of conj_grad
 * (citcoms/lib/General_matrix_functions.c:307-424): d0 = 0, r = F; loop while (residual > acc && count < *cycles) || count == 0;
 * the mat-vec is g4s_elem_op_apply followed by zeroing the boundary rows (assemble_del2_u(..., strip_bcs = 1), the list of
 * citcoms/lib/BC_util.c:89-102). All vectors stay in HBM and the loop test runs on the device: the host enqueues iterations in batches and
 * reads 40 bytes of state per batch.
 * *cycles: in = iteration cap (vlowstep), out = iterations done. zero_resid_dev may be NULL when n_zero == 0.
 * Exactly one of op / A selects the operator (element-by-element or assembled CSR). */
g4s_status g4s_conj_grad(g4s_elem_op_t op, g4s_csr_t A, int32_t neq, const double *BI_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                         const double *F_dev, double *d0_dev, double acc, int32_t *cycles, double *residual, void *stream);

/* ---- CitcomS's node-assembled stiffness operator (the other mat-vec format of assemble_del2_u) — SURVEY.md §8 f1.
 * node_map[nno·max_eqn] = E->Node_map[lev][m] (slot group 0: the node's own three equations, groups 1..13: lower-numbered
 * neighbours, unused slots = neq; citcoms/lib/Construct_arrays.c:254-328), eqn_k1..3[nno·max_eqn] = E->Eqn_k1..3[lev][m]
 * (construct_node_ks :335-470), id[nno·3] = E->ID .doff[1..3]; all host pointers, 0-based nodes, copied. The stored symmetric half
 * is expanded once into per-node 3×3 neighbour blocks so that the mat-vec is a gather. */
g4s_status g4s_node_op_create(g4s_node_op_t *out, int32_t nno, int32_t neq, int32_t max_eqn, const int32_t *node_map,
                              const int32_t *id, const double *eqn_k1, const double *eqn_k2, const double *eqn_k3);
g4s_status g4s_node_op_destroy(g4s_node_op_t op);
/* Au = K·u, then the listed rows zeroed (strip_bcs) — n_assemble_del2_u, citcoms/lib/Element_calculations.c:516-577.
 * u_dev / Au_dev hold neq doubles (the reference's dummy entry [neq] is not needed: unused slots are dropped at create). */
g4s_status g4s_node_op_apply(g4s_node_op_t op, const double *u_dev, double *Au_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                             void *stream);
/* g4s_conj_grad (below) on the node-assembled operator. */
g4s_status g4s_conj_grad_node(g4s_node_op_t op, int32_t neq, const double *BI_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                              const double *F_dev, double *d0_dev, double acc, int32_t *cycles, double *residual, void *stream);

/* ---- The incompressibility (Uzawa) iteration of CitcomS around the velocity solve — SURVEY.md §8 f1.
 * g_dev[e·npe·dof + p] = elt_del[e].g[p][0], the per-element divergence / gradient vector; pressure unknowns are elements. */

/* divU[e] = Σ_a (g[3a]·U[eq1] + g[3a+1]·U[eq2] + g[3a+2]·U[eq3]) — assemble_div_u, citcoms/lib/Element_calculations.c:701-729. */
g4s_status g4s_elem_op_div_u(g4s_elem_op_t op, const double *g_dev, const double *U_dev, double *divU_dev, void *stream);
/* gradP = Σ_e g·P[e] scattered to the equations (here: gathered per node, no atomics), then the boundary rows zeroed —
 * assemble_grad_p, Element_calculations.c:737-779 (strip_bcs_from_residual, BC_util.c:89-102). */
g4s_status g4s_elem_op_grad_p(g4s_elem_op_t op, const double *g_dev, const double *P_dev, double *gradP_dev,
                              const int32_t *zero_resid_dev, int32_t n_zero, void *stream);
/* BPI[e] = 1 / Σ_p g[e][p]·BI[eq(e,p)]·g[e][p] (1 where that is 0) — build_diagonal_of_Ahat / assemble_dAhatp_entry,
 * Element_calculations.c:613-644, 785-830. */
g4s_status g4s_elem_op_pressure_preconditioner(g4s_elem_op_t op, const double *g_dev, const double *BI_dev, double *BPI_dev, void *stream);

typedef struct g4s_stokes_params {
    double imp;                            /* accuracy of the outer iteration (control.accuracy) */
    double inner_accuracy_scale;           /* control.inner_accuracy_scale */
    double v_res;                          /* monitor.fdotf: the inner solves run to imp·inner_accuracy_scale·v_res */
    int32_t v_steps_low;                   /* iteration cap of one velocity solve (control.v_steps_low) */
    int32_t steps_max;                     /* cap of the outer iteration */
    int32_t check_continuity_convergence;  /* keep_iterating: || instead of && (Stokes_flow_Incomp.c:150-162) */
    int32_t check_pressure_convergence;    /* "converging" also needs dpressure < imp */
} g4s_stokes_params;

typedef struct g4s_stokes_result {
    int32_t outer_iterations;              /* *steps_max on return */
    int32_t last_solve_valid;              /* the last velocity solve reached its accuracy */
    int64_t inner_iterations;              /* CG iterations of all velocity solves */
    double incompressibility, v_norm, p_norm, dvelocity, dpressure;   /* the quantities of print_convergence_progress */
} g4s_stokes_result;

/* solve_Ahat_p_fhat_CG, citcoms/lib/Stokes_flow_Incomp.c:188-452, incompressible case (initial_vel_residual :839-881 included),
 * velocity solves by g4s_conj_grad (solve_del2_u's CG branch, General_matrix_functions.c:89-94) on the element-by-element
 * operator of `op`, or — K_csr != NULL — on the assembled stiffness matrix through g4s_spmv (BASELINE config 5: "assembled stiffness
 * matrix driving G4S SpMV inside the CG/Uzawa solver loop"; op then only supplies the mesh maps of div/grad; the boundary rows are
 * zeroed after every product either way, so K_csr is the plain assembly of the element matrices).
 * V_dev[neq] and P_dev[nel] are updated in place; F_dev is not modified. nmass_dev[nno] = NMass, area_dev[nel] = eco[].area,
 * volume = mesh.volume (the weights of global_v_norm2 / global_p_norm2 / global_div_norm2, Global_operations.c:591-656).
 * hist (host, may be NULL): 5 doubles per printed line — v_norm, p_norm, dvelocity, dpressure, incompressibility — line 0 before
 * the loop, at most hist_lines lines. Every vector stays on the device; a few scalars per outer iteration cross PCIe. */
g4s_status g4s_stokes_uzawa_cg(g4s_elem_op_t op, g4s_csr_t K_csr, const double *g_dev, const double *BI_dev, const double *BPI_dev, const double *nmass_dev,
                               const double *area_dev, double volume, const int32_t *zero_resid_dev, int32_t n_zero, const double *F_dev,
                               double *V_dev, double *P_dev, const g4s_stokes_params *params, g4s_stokes_result *result,
                               double *hist, int32_t hist_lines, void *stream);

/* ---- g4s_conj_grad with the loop opened up, for a row-partitioned operator on several GPUs (1-D row partition,
 * all-gather / halo exchange of the direction vector, all-reduce of the dot products). Every rank holds its slab of F, BI, d0; the
 * caller owns the mat-vec (exchange p, local g4s_spmv into Ap) and, between the steps, all-reduces (sum) the partial sums
 * element-wise: partials[0..256) = r·z, [256..512) = p·Ap, [512..768) = r·r. Sequence:
 *   g4s_cg_begin → all-reduce [512..768) and [0..256) → g4s_cg_direction → g4s_cg_state (done?) → g4s_cg_buffers: p → Ap = A·p →
 *   g4s_cg_reduce_pAp → all-reduce [256..512) → g4s_cg_update → all-reduce [512..768), [0..256) → g4s_cg_direction → … → g4s_cg_end.
 * With one rank and no all-reduce this is g4s_conj_grad step by step. zero_resid indices are local to the slab. */
g4s_status g4s_cg_ws_create(g4s_cg_ws_t *out, int32_t n_local);
g4s_status g4s_cg_ws_destroy(g4s_cg_ws_t ws);
g4s_status g4s_cg_begin(g4s_cg_ws_t ws, const double *F_dev, const double *BI_dev, double *d0_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                        void *stream);
g4s_status g4s_cg_direction(g4s_cg_ws_t ws, int32_t steps, double acc, void *stream);   /* loop test + β + p (conj_grad :364-379) */
g4s_status g4s_cg_state(g4s_cg_ws_t ws, int32_t *count, int32_t *done, double *residual, void *stream);   /* synchronises */
g4s_status g4s_cg_buffers(g4s_cg_ws_t ws, double **p_dev, double **Ap_dev, double **partials_dev);        /* valid until the next g4s_cg_update */
g4s_status g4s_cg_reduce_pAp(g4s_cg_ws_t ws, void *stream);                              /* boundary rows of Ap := 0, partial p·Ap */
g4s_status g4s_cg_update(g4s_cg_ws_t ws, const double *BI_dev, double *d0_dev, void *stream);   /* α, d0, r, z (:383-402) */
g4s_status g4s_cg_end(g4s_cg_ws_t ws, double *d0_dev, const int32_t *zero_resid_dev, int32_t n_zero, void *stream);   /* d0 boundary rows := 0 (:409) */

/* The same loop closed in C for a row-partitioned operator: conj_grad (General_matrix_functions.c:307-424) with the product
 * g4s_spmv_dist_apply(A) and the dot products' partial sums all-reduced by g4s_comm_allreduce_sum_f64(comm) — the neighbour exchange of
 * Regional_parallel_related.c:744-789 and the MPI_Allreduce of Global_operations.c:534-562 on RCCL. Every rank calls it with its slab
 * (n_local rows; BI, F, d0 device arrays of that length; zero_resid: LOCAL indices of the boundary equations) and gets the same cycles
 * and residual. A must be connected to comm (g4s_spmv_dist_connect_rccl). */
g4s_status g4s_conj_grad_dist(g4s_spmv_dist_t A, void *comm, int32_t n_local, const double *BI_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                              const double *F_dev, double *d0_dev, double acc, int32_t steps, int32_t *cycles, double *residual, void *stream);

/* The two collectives a partitioned Krylov solver needs, as a pair of callbacks, so that ONE loop in C serves RCCL and any transport the
 * host already has (MPI in CitcomS; gloo in the tests): the sum all-reduce of the dot products (Global_operations.c:534-562) and the
 * exchange of the x entries of one distributed product (Regional_parallel_related.c:744-789).
 *   allreduce_sum_f64(ctx, buf_dev, count, stream): element-wise sum over the ranks, in place, device buffer, ordered on `stream`.
 *   exchange(ctx, h, stream): called between g4s_spmv_dist_begin(h) and g4s_spmv_dist_finish(h): carry h's send buffer to the peers and fill
 *   its receive buffer (g4s_spmv_dist_buffers). NULL: the handle's own RCCL wiring does it (g4s_spmv_dist_apply).
 * g4s_transport_rccl fills the pair for an RCCL communicator the handles are connected to. */
typedef struct g4s_transport {
    void *ctx;
    g4s_status (*allreduce_sum_f64)(void *ctx, double *buf_dev, int64_t count, void *stream);
    g4s_status (*exchange)(void *ctx, g4s_spmv_dist_t h, void *stream);
} g4s_transport;
g4s_status g4s_transport_rccl(void *comm, g4s_transport *out);
/* g4s_conj_grad_dist over a transport. */
g4s_status g4s_conj_grad_dist_tr(g4s_spmv_dist_t A, const g4s_transport *tr, int32_t n_local, const double *BI_dev, const int32_t *zero_resid_dev,
                                 int32_t n_zero, const double *F_dev, double *d0_dev, double acc, int32_t steps, int32_t *cycles, double *residual,
                                 void *stream);

/* The Uzawa / Schur-complement CG iteration of g4s_stokes_uzawa_cg on a PARTITIONED operator (BASELINE configs[4] "1 vs 8 GPUs";
 * solve_Ahat_p_fhat_CG, citcoms/lib/Stokes_flow_Incomp.c:188-452, with the reductions of Global_operations.c:591-656 over the ranks).
 * The velocity unknowns (equations) and the pressure unknowns (elements) each have a 1-D partition; every rank holds its slabs of all
 * vectors and three distributed operators over those partitions:
 *   K   equations × equations   the assembled stiffness matrix (square; the inner solves are g4s_conj_grad_dist_tr on it);
 *   D   elements × equations    assemble_div_u:  divU[e] = Σ_p g[e][p]·U[eq(e,p)]      (Element_calculations.c:744-779), as a CSR matrix;
 *   Dt  equations × elements    assemble_grad_p: its transpose                           (:701-741)            (g4s_spmv_dist_create_rect).
 * On one rank the element-ordered sums of g4s_stokes_uzawa_cg become CSR row sums: same terms, same order within a part, the own-column
 * and remote-column parts added separately — results agree to rounding, iteration counts normally exactly.
 * vmass_dev[neq_local] = NMass of the node that owns the equation (the weight of global_v_norm2), area_dev[nel_local], volume = mesh volume,
 * zero_resid: LOCAL equation indices; BI / BPI: the two preconditioner diagonals, local slabs. V / P updated in place.
 * The host waits once per outer iteration: a velocity solve's first batch of iterations is enqueued with what follows it, and the rest of the iteration is
 * enqueued again if that batch turns out not to have met the solve's test (all ranks read the same all-reduced sums and take the same turn);
 * G4S_STOKES_SYNC=1 waits for every solve instead. */
g4s_status g4s_stokes_uzawa_cg_dist(g4s_spmv_dist_t K, g4s_spmv_dist_t D, g4s_spmv_dist_t Dt, const g4s_transport *tr, int32_t neq_local, int32_t nel_local,
                                    const double *BI_dev, const double *BPI_dev, const double *vmass_dev, const double *area_dev, double volume,
                                    const int32_t *zero_resid_dev, int32_t n_zero, const double *F_dev, double *V_dev, double *P_dev,
                                    const g4s_stokes_params *params, g4s_stokes_result *result, double *hist, int32_t hist_lines, void *stream);

/* result[M×K] = xx[M×N] · w[N×K], row-major fp64, device pointers (opt_matmul.cc:24-62). */
g4s_status g4s_dense_rows_times_matrix(int32_t M, int32_t N, int32_t K, const double *xx_dev, const double *w_dev,
                                       double *result_dev, void *stream);

/* Gradient of the op above — _opt_matmul_grad (deepmd/source/op/_opt_matmul_grad.py:6-12):
 *   dxx[M×N] = grad[M×K] · wᵀ      (tf.matmul(grad, w, False, True))
 *   dw[N×K]  = xxᵀ · grad[M×K]     (tf.matmul(xx, grad, True, False))
 * Row-major fp64 device pointers. dxx_dev or dw_dev may be NULL to skip that product. dw is reduced over the M rows in a fixed
 * slab order (no atomics): the same inputs give the same bits. */
g4s_status g4s_dense_rows_times_matrix_grad(int32_t M, int32_t N, int32_t K, const double *xx_dev, const double *w_dev,
                                            const double *grad_dev, double *dxx_dev, double *dw_dev, void *stream);

/* ---- The reference's dense comparison drivers: what it times next to the sparse kernels, not the hot path.
 * Column-major dim×dim fp64, alpha = 1, beta = 0, as the reference calls MKL. Host pointers, or device pointers with
 * G4S_DEVICE_POINTERS; synchronous.
 *   g4s_dense_mm — mm/src/cblas_dxxmm.c:57-111:  DGEMM  C = A·B          (cblas_dgemm ColMajor NoTrans NoTrans, :96-111)
 *                                                DSYMM  C = sym(A)·B     (cblas_dsymm Left Upper: only A's upper triangle is read, :57-76)
 *                                                DTRMM  B := B·triu(A)   (cblas_dtrmm Right Upper NoTrans NonUnit, in place; C unused, :78-95)
 *   g4s_dense_mv — mv/mv.c:6-27:                 DGEMV  y = A·x (:23-27)   DSYMV  y = sym(A)·x (Upper, :6-10)
 *                                                DTRMV  x := triu(A)ᵀ·x (Upper Trans NonUnit, in place; y unused, :12-15)
 *                                                DSPMV  y = sym(AP)·x, AP the packed upper triangle, dim(dim+1)/2 doubles (:17-21)
 * Level 3 runs on the fp64 MFMA GEMM of g4s_dense_rows_times_matrix; level 2 is one HBM pass over the matrix. */
#define G4S_DENSE_DGEMM 1
#define G4S_DENSE_DSYMM 2
#define G4S_DENSE_DTRMM 3
#define G4S_DENSE_DGEMV 4
#define G4S_DENSE_DSYMV 5
#define G4S_DENSE_DTRMV 6
#define G4S_DENSE_DSPMV 7
g4s_status g4s_dense_mm(int32_t kind, int32_t dim, const double *A, double *B, double *C, unsigned flags);
g4s_status g4s_dense_mv(int32_t kind, int32_t dim, const double *A, double *x, double *y, unsigned flags);

/* result[0] += Σ_i Σ_{j<i} x_i x_j (a[num·(i+m·j)] + a[num·(j+m·i)]) + Σ_i x_i² a[num·(i+m·i)];
 * numbers == 1: result[1] += Σ_i x_i·b_i (apply1); numbers > 1: result[1] likewise on a[…+1] (gather2/apply2).
 * Host pointers (the operands are ~100×100); result is a host double[2] that is accumulated into. */
g4s_status g4s_sym_quadratic_form(int32_t m, int32_t numbers, const double *a, const double *x, const double *b,
                                  double *result);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* G4S_H */
