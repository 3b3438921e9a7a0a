// spmv_bcsr.hpp — interface of the block-row SpMV path (spmv_bcsr.hip) used by the CSR handle (spmv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace g4s {
struct BcsrPlan;
// *out stays NULL (status OK) when the matrix is not made of aligned b×b blocks (b = 3, 2, 4 are tried)
int bcsr_try_build(BcsrPlan **out, int rows, int cols, long long nnz, const int32_t *d_rowptr, const int32_t *d_colids, const double *d_values, bool use_nt);
int bcsr_update_values(BcsrPlan *plan, const int32_t *d_colids, const double *d_values, hipStream_t stream);
void bcsr_destroy(BcsrPlan *plan);
long long bcsr_bytes(const BcsrPlan *plan);
int bcsr_block(const BcsrPlan *plan);
int bcsr_spmv(BcsrPlan *plan, const double *x, double *y, double alpha, double beta, hipStream_t stream);
} // namespace g4s
