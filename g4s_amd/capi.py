"""ctypes binding of libg4s_hip.so (the C-ABI declared in include/g4s.h and include/g4s_synth.h).

The library is the product; this module only loads it and declares argument types. It fails loudly when the
shared object is missing — there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("G4S_LIB") or os.path.join(_HERE, "lib", "libg4s_hip.so")   # G4S_LIB: the host-sanitized build (tools/run_sanitized_cpu_tests.sh)

OK, ERR_INVALID, ERR_NOMEM, ERR_HIP, ERR_OVERFLOW, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
HOST_POINTERS, DEVICE_POINTERS, SORT_OUTPUT, SPMV_NO_NT, SPMV_BLOCKED, SPMV_STREAM, DIST_LOOPBACK, DIST_ALLGATHER, SPMV_UPDATABLE = 0, 1, 2, 4, 8, 16, 32, 64, 128
PATTERN_ELEMENT_BLOCK_MATVEC, PATTERN_DENSE_ROW_TIMES_MATRIX, PATTERN_SYM_QUADRATIC_FORM = 1, 2, 3
DENSE_DGEMM, DENSE_DSYMM, DENSE_DTRMM, DENSE_DGEMV, DENSE_DSYMV, DENSE_DTRMV, DENSE_DSPMV = 1, 2, 3, 4, 5, 6, 7

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p

FUN_GATHER = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(f64p), f64p, f64p)   # citcoms/lib/global_defs.h:48
FUN_APPLY = C.CFUNCTYPE(None, C.c_int, C.POINTER(f64p), f64p, f64p)             # citcoms/lib/global_defs.h:49


class CsrInfo(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("nnz", C.c_int64),
                ("stream_blocks", C.c_int32), ("long_rows", C.c_int32), ("long_chunks", C.c_int32),
                ("tile_nnz", C.c_int32), ("tile_rows", C.c_int32), ("long_chunk_nnz", C.c_int32),
                ("algorithmic_bytes", C.c_int64), ("plan_bytes", C.c_int64), ("spmv_path", C.c_int32), ("reserved", C.c_int32)]


class DistInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("local_rows", C.c_int32), ("n_ref", C.c_int32), ("nnz_own", C.c_int64), ("nnz_rem", C.c_int64),
                ("send_bytes", C.c_int64), ("recv_bytes", C.c_int64), ("own_path", C.c_int32), ("rem_path", C.c_int32), ("connected", C.c_int32),
                ("reserved", C.c_int32)]


class DistSplit(C.Structure):
    """g4s_dist_split: one rank's rows cut into own-column and remote-column parts (host arrays from g4s_malloc)."""
    _fields_ = [("local_rows", C.c_int32), ("n_ref", C.c_int32), ("merged", C.c_int32), ("allgather", C.c_int32),
                ("nnz_own", C.c_int64), ("nnz_rem", C.c_int64), ("pad", C.c_int64),
                ("own_rowptr", C.POINTER(C.c_int32)), ("own_colids", C.POINTER(C.c_int32)), ("own_values", C.POINTER(C.c_double)),
                ("rem_rowptr", C.POINTER(C.c_int32)), ("rem_colids", C.POINTER(C.c_int32)), ("rem_values", C.POINTER(C.c_double)),
                ("want", C.POINTER(C.c_int32)), ("recv_cut", C.POINTER(C.c_int64))]


ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)      # g4s_transport.allreduce_sum_f64(ctx, buf_dev, count, stream)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)                  # g4s_transport.exchange(ctx, h, stream)


class Transport(C.Structure):
    """g4s_transport: the two collectives of a partitioned Krylov solver as callbacks."""
    _fields_ = [("ctx", C.c_void_p), ("allreduce_sum_f64", ALLREDUCE_CB), ("exchange", EXCHANGE_CB)]


class Timings(C.Structure):
    """mm/inc/Timings.h:4-23 — seven stage times in milliseconds."""
    _fields_ = [(n, C.c_double) for n in ("create", "spmm", "convert", "order", "export_csr", "destroy", "total")]


class PatternDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("num_elems", C.c_int32), ("nodes_per_elem", C.c_int32), ("dof", C.c_int32),
                ("ien", vp), ("id", vp), ("nno", C.c_int32), ("neq", C.c_int32), ("edge_weight_base", C.c_int32),
                ("static_weights", C.c_int32), ("inner", C.c_int32), ("numbers", C.c_int32)]


class StokesParams(C.Structure):
    """g4s_stokes_params — the controls solve_Ahat_p_fhat_CG reads (citcoms/lib/Stokes_flow_Incomp.c:188-452)."""
    _fields_ = [("imp", C.c_double), ("inner_accuracy_scale", C.c_double), ("v_res", C.c_double), ("v_steps_low", C.c_int32),
                ("steps_max", C.c_int32), ("check_continuity_convergence", C.c_int32), ("check_pressure_convergence", C.c_int32)]


class StokesResult(C.Structure):
    _fields_ = [("outer_iterations", C.c_int32), ("last_solve_valid", C.c_int32), ("inner_iterations", C.c_int64),
                ("incompressibility", C.c_double), ("v_norm", C.c_double), ("p_norm", C.c_double), ("dvelocity", C.c_double),
                ("dpressure", C.c_double)]


# name -> (restype, argtypes); every symbol declared in include/*.h appears here (tests check the export list).
SIGNATURES = {
    "g4s_version": (C.c_char_p, []),
    "g4s_build_info": (C.c_char_p, []),
    "g4s_warm_up": (C.c_int, []),
    "g4s_last_error": (C.c_char_p, []),
    "g4s_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "g4s_set_device": (C.c_int, [C.c_int]),
    "g4s_device_synchronize": (C.c_int, []),
    "g4s_shutdown": (C.c_int, []),
    "g4s_trim": (C.c_int, []),
    "g4s_malloc": (vp, [C.c_size_t]),
    "g4s_free": (None, [vp]),
    "g4s_dev_alloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
    "g4s_dev_free": (C.c_int, [vp]),
    "g4s_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
    "g4s_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
    "g4s_csr_create": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, C.c_uint]),
    "g4s_csr_destroy": (C.c_int, [vp]),
    "g4s_csr_update_values": (C.c_int, [vp, vp, C.c_uint, vp]),
    "g4s_spmv_dist_update_values": (C.c_int, [vp, vp, C.c_uint, vp]),
    "g4s_csr_get_info": (C.c_int, [vp, C.POINTER(CsrInfo)]),
    "g4s_csr_device_arrays": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "g4s_spmv": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, vp]),
    "g4s_spmv_csr_i32_f64": (C.c_int, [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_uint]),
    "g4s_prim_exclusive_scan_i32": (C.c_int, [vp, vp, C.c_int64, vp]),
    "g4s_prim_exclusive_scan_i64": (C.c_int, [vp, vp, C.c_int64, vp]),
    "g4s_prim_sort_pairs_desc_i32": (C.c_int, [vp, vp, vp, vp, C.c_int32, C.c_int32, vp]),
    "g4s_row_partition": (C.c_int, [C.c_int32, vp, vp, C.c_int64, C.c_int32, i64p, C.c_uint]),
    "g4s_dist_split_rows": (C.c_int, [C.c_int32, C.c_int32, i64p, C.c_int64, vp, vp, vp, C.c_uint, C.POINTER(DistSplit)]),
    "g4s_dist_split_free": (None, [C.POINTER(DistSplit)]),
    "g4s_spmv_dist_create": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, i64p, C.c_int64, vp, vp, vp, C.c_uint]),
    "g4s_spmv_dist_create_rect": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, i64p, i64p, vp, vp, vp, C.c_uint]),
    "g4s_spmv_dist_create_columns": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, i64p, C.c_int32, vp, vp, vp, C.c_uint]),
    "g4s_spmv_dist_destroy": (C.c_int, [vp]),
    "g4s_spmv_dist_get_info": (C.c_int, [vp, C.POINTER(DistInfo)]),
    "g4s_spmv_dist_connect_rccl": (C.c_int, [vp, vp]),
    "g4s_spmv_dist_want": (C.c_int, [vp, C.c_int32, i64p, C.POINTER(vp)]),
    "g4s_spmv_dist_set_give": (C.c_int, [vp, C.c_int32, C.c_int64, vp, C.c_uint]),
    "g4s_spmv_dist_apply": (C.c_int, [vp, vp, vp, vp]),
    "g4s_spmv_dist_begin": (C.c_int, [vp, vp, vp, vp]),
    "g4s_spmv_dist_buffers": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64p), C.POINTER(vp), C.POINTER(i64p)]),
    "g4s_spmv_dist_finish": (C.c_int, [vp, vp, vp]),
    "g4s_transport_rccl": (C.c_int, [vp, C.POINTER(Transport)]),
    "g4s_conj_grad_dist_tr": (C.c_int, [vp, C.POINTER(Transport), C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_double, C.c_int32, C.POINTER(C.c_int32), f64p, vp]),
    "g4s_stokes_uzawa_cg_dist": (C.c_int, [vp, vp, vp, C.POINTER(Transport), C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_double, vp, C.c_int32, vp, vp, vp,
                                           C.POINTER(StokesParams), C.POINTER(StokesResult), vp, C.c_int32, vp]),
    "g4s_comm_unique_id": (C.c_int, [vp]),
    "g4s_comm_create": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, vp]),
    "g4s_comm_destroy": (C.c_int, [vp]),
    "g4s_comm_allreduce_sum_f64": (C.c_int, [vp, vp, C.c_int64, vp]),
    "g4s_conj_grad_dist": (C.c_int, [vp, vp, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_double, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), vp]),
    "g4s_spgemm_flop": (C.c_int, [C.c_int32, vp, vp, vp, i64p, vp, C.c_uint]),
    "g4s_spgemm_csr_i32_f64": (C.c_int, [vp, vp, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                          C.c_int32, C.c_int32, C.c_int32, i64p, C.POINTER(Timings), C.c_uint]),
    "g4s_spgemm_symbolic": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, i64p, vp]),
    "g4s_spgemm_numeric": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint, vp]),
    "g4s_register_pattern": (C.c_int, [FUN_GATHER, FUN_APPLY, C.POINTER(PatternDesc)]),
    "g4s_unregister_pattern": (C.c_int, [FUN_GATHER, FUN_APPLY]),
    "g4s_set_host_callback_policy": (C.c_int, [C.c_int32]),
    "g4s_set_host_callback_policy_thread": (C.c_int, [C.c_int32, C.POINTER(C.c_int32)]),
    "spmm_dense": (None, [C.c_uint32, C.c_uint32, vp, vp, vp, vp, FUN_GATHER, FUN_APPLY, f64p, C.c_int]),
    "g4s_spmm_dense": (C.c_int, [C.c_uint32, C.c_uint32, vp, vp, vp, vp, FUN_GATHER, FUN_APPLY, f64p, C.c_int]),
    "g4s_elem_op_create": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp]),
    "g4s_elem_op_destroy": (C.c_int, [vp]),
    "g4s_elem_op_apply": (C.c_int, [vp, vp, vp, vp]),
    "g4s_elem_op_inverse_diagonal": (C.c_int, [vp, vp, vp]),
    "g4s_conj_grad": (C.c_int, [vp, vp, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_double, C.POINTER(C.c_int32), f64p, vp]),
    "g4s_cg_ws_create": (C.c_int, [C.POINTER(vp), C.c_int32]),
    "g4s_cg_ws_destroy": (C.c_int, [vp]),
    "g4s_cg_begin": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, vp]),
    "g4s_cg_direction": (C.c_int, [vp, C.c_int32, C.c_double, vp]),
    "g4s_cg_state": (C.c_int, [vp, i32p, i32p, f64p, vp]),
    "g4s_cg_buffers": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "g4s_cg_reduce_pAp": (C.c_int, [vp, vp]),
    "g4s_cg_update": (C.c_int, [vp, vp, vp, vp]),
    "g4s_cg_end": (C.c_int, [vp, vp, vp, C.c_int32, vp]),
    "g4s_node_op_create": (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]),
    "g4s_node_op_destroy": (C.c_int, [vp]),
    "g4s_node_op_apply": (C.c_int, [vp, vp, vp, vp, C.c_int32, vp]),
    "g4s_conj_grad_node": (C.c_int, [vp, C.c_int32, vp, vp, C.c_int32, vp, vp, C.c_double, C.POINTER(C.c_int32), f64p, vp]),
    "g4s_elem_op_div_u": (C.c_int, [vp, vp, vp, vp, vp]),
    "g4s_elem_op_grad_p": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, vp]),
    "g4s_elem_op_pressure_preconditioner": (C.c_int, [vp, vp, vp, vp, vp]),
    "g4s_stokes_uzawa_cg": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_double, vp, C.c_int32, vp, vp, vp, C.POINTER(StokesParams),
                                      C.POINTER(StokesResult), vp, C.c_int32, vp]),
    "g4s_dense_rows_times_matrix": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]),
    "g4s_dense_rows_times_matrix_grad": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp]),
    "g4s_sym_quadratic_form": (C.c_int, [C.c_int32, C.c_int32, vp, vp, vp, vp]),
    "g4s_dense_mm": (C.c_int, [C.c_int32, C.c_int32, vp, vp, vp, C.c_uint]),
    "g4s_dense_mv": (C.c_int, [C.c_int32, C.c_int32, vp, vp, vp, C.c_uint]),
    # include/g4s_synth.h
    "g4s_synth_rmat_keys": (C.c_int, [C.c_uint64, C.c_int32, C.c_int64, C.c_int64, C.c_int64, vp, vp]),
    "g4s_synth_csr_from_keys": (C.c_int, [C.c_uint64, C.c_int64, C.c_int32, vp, C.c_int64, vp, vp, vp, vp]),
    "g4s_synth_vector": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, vp, vp]),
    "g4s_synth_laplacian_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, vp, vp, vp, vp,
                                           C.c_int32, vp]),
    "g4s_synth_banded": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, vp, vp, vp, vp]),
}


class G4SError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"g4s status {status}: {message}")
        self.status = status


_lib = None


def load():
    """Load libg4s_hip.so once. torch is imported first so that the HIP runtime both sides use is the one
    already mapped (same SONAME, libamdhip64.so.7); two runtimes in one process would not share device pointers."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"or `make -C g4s_amd/csrc`. g4s_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (maps torch's libamdhip64.so.7 before ours resolves its NEEDED entry)
    except Exception:  # pragma: no cover - torch is plumbing; a plain C++ host does not need it
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != OK:
        raise G4SError(status, load().g4s_last_error().decode())
    return status

HOST_CALLBACKS_SERIAL, HOST_CALLBACKS_PARALLEL, HOST_CALLBACKS_REFUSE = 0, 1, 2   # g4s_set_host_callback_policy
