// spmv_pb.hpp — interface of the propagation-blocked SpMV path (spmv_pb.hip) used by the CSR handle (spmv.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace g4s {
struct PbPlan;
int pb_build(PbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values);
void pb_destroy(PbPlan *plan);
long long pb_bytes(const PbPlan *plan);
int pb_spmv(PbPlan *plan, const double *x, double *y, double alpha, double beta, hipStream_t stream);
int pb_spmv_status(const PbPlan *plan);   // device-synchronising health check of the one-launch form (used by g4s_csr_get_info)
bool pb_should_use(int rows, int cols, long long nnz, const int *d_colids);
// tile-blocked successor (spmv_tb.hip): y tile and x tile both in LDS for dense cells, propagation only for the sparse rest
struct TbPlan;
int tb_build(TbPlan **out, int rows, int cols, long long nnz, const int *d_rowptr, const int *d_colids, const double *d_values);
void tb_destroy(TbPlan *plan);
long long tb_bytes(const TbPlan *plan);
int tb_spmv(TbPlan *plan, const double *x, double *y, double alpha, double beta, hipStream_t stream);
} // namespace g4s
