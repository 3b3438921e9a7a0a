// spmv_pb.hpp — interface of the propagation-blocked SpMV path (spmv_pb.hip) used by the CSR handle (spmv.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace g4s {
struct PbPlan;
int pb_build(PbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values, bool keep_value_map);
bool pb_has_value_map(const PbPlan *plan);
int pb_update_values(PbPlan *plan, const double *d_values, hipStream_t stream);   // needs keep_value_map
void pb_destroy(PbPlan *plan);
long long pb_bytes(const PbPlan *plan);
int pb_spmv(PbPlan *plan, const double *x, double *y, double alpha, double beta, hipStream_t stream);
bool pb_should_use(int rows, int cols, long long nnz, const int *d_colids);
} // namespace g4s
