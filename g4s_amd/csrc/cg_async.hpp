// cg_async.hpp — a conjugate-gradient solve whose first batch of iterations is only ENQUEUED (cg.hip: CgRun), for callers that go on enqueuing
// dependent work speculatively and read the solve's state with their own scalars in one host synchronisation (stokes.hip).
#pragma once
#include "common.hpp"

namespace g4s {
struct CgAsync;
// Enqueues: set-up, the first batch of iterations (one more than this thread's previous solve needed), the loop test, the strip of d0's boundary rows.
int cg_async_start(CgAsync **out, g4s_elem_op_t op, g4s_csr_t A, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                   const double *F, double *d0, double acc, int32_t steps, hipStream_t s, const unsigned char *bc_mask = nullptr);
// the boundary-equation byte mask a solve needs (neq bytes): a caller with many solves on one set of boundary rows builds it once and passes it in
int cg_build_mask(int32_t neq, const int32_t *zero_resid, int32_t n_zero, unsigned char *mask, hipStream_t s);
int cg_async_read(CgAsync *c);                      // enqueues the copy of the solve's state into the object (the caller synchronises)
// After that synchronisation. *speculation_held = the first batch met the loop test, so everything enqueued behind it used the final d0. If it did
// not, the solve is run to its end here (synchronising) and d0 stripped again; the caller must redo what it had enqueued behind the solve.
int cg_async_settle(CgAsync *c, bool *speculation_held, int32_t *cycles, double *residual);
void cg_async_free(CgAsync *c);

// The same for a row-partitioned operator (g4s_conj_grad_dist_tr's loop): the product and the all-reduces of the dot products go through the
// transport; `ws` (g4s_cg_ws_create) is the caller's, reused from solve to solve. Every rank reads the same all-reduced sums and takes the same turn.
struct DistCgAsync;
int dist_cg_async_start(DistCgAsync **out, g4s_cg_ws_t ws, g4s_spmv_dist_t A, const g4s_transport *tr, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                        const double *F, double *d0, double acc, int32_t steps, void *stream);
void cg_ws_hold_mask(g4s_cg_ws_t ws, bool hold);   // the caller promises that the zero_resid list it passes keeps its contents while it holds: the mask is built once
int dist_cg_async_read(DistCgAsync *c);
int dist_cg_async_settle(DistCgAsync *c, bool *speculation_held, int32_t *cycles, double *residual);
void dist_cg_async_free(DistCgAsync *c);
} // namespace g4s
