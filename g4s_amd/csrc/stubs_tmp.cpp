// TEMPORARY: entry points not implemented yet return G4S_ERR_UNSUPPORTED (replaced as each lands).
#include "common.hpp"
#define NOTYET return g4s::set_error(G4S_ERR_UNSUPPORTED, "%s: not implemented yet", __func__)
G4S_API g4s_status g4s_spgemm_flop(int32_t, const int32_t *, const int32_t *, const int32_t *, int64_t *, int64_t *, unsigned) { NOTYET; }
G4S_API g4s_status g4s_spgemm_csr_i32_f64(const int32_t *, const int32_t *, const double *, const int32_t *, const int32_t *, const double *,
                                          int32_t **, int32_t **, double **, int32_t, int32_t, int32_t, int64_t *, g4s_timings *, unsigned) { NOTYET; }
G4S_API g4s_status g4s_spgemm_symbolic(int32_t, int32_t, int32_t, const int32_t *, const int32_t *, const int32_t *, const int32_t *,
                                       int32_t *, int64_t *, void *) { NOTYET; }
G4S_API g4s_status g4s_spgemm_numeric(int32_t, int32_t, int32_t, const int32_t *, const int32_t *, const double *, const int32_t *,
                                      const int32_t *, const double *, const int32_t *, int32_t *, double *, unsigned, void *) { NOTYET; }
G4S_API g4s_status g4s_register_pattern(fun_gather, fun_apply, const g4s_pattern_desc *) { NOTYET; }
G4S_API g4s_status g4s_unregister_pattern(fun_gather, fun_apply) { NOTYET; }
G4S_API void spmm_dense(uint32_t, uint32_t, const double **, const double *, double *, double *, fun_gather, fun_apply, double *, int)
{ fprintf(stderr, "g4s: spmm_dense not implemented yet\n"); abort(); }
G4S_API g4s_status g4s_spmm_dense(uint32_t, uint32_t, const double **, const double *, double *, double *, fun_gather, fun_apply, double *, int) { NOTYET; }
G4S_API g4s_status g4s_elem_op_create(g4s_elem_op_t *, int32_t, int32_t, int32_t, const int32_t *, const int32_t *, int32_t, int32_t, const double *) { NOTYET; }
G4S_API g4s_status g4s_elem_op_destroy(g4s_elem_op_t) { NOTYET; }
G4S_API g4s_status g4s_elem_op_apply(g4s_elem_op_t, const double *, double *, void *) { NOTYET; }
G4S_API g4s_status g4s_dense_rows_times_matrix(int32_t, int32_t, int32_t, const double *, const double *, double *, void *) { NOTYET; }
G4S_API g4s_status g4s_sym_quadratic_form(int32_t, int32_t, const double *, const double *, const double *, double *) { NOTYET; }
