"""Intel oneMKL called directly through ctypes — TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/: only tests/,
tests/golden/make_golden.py and bench.py's cpu_baseline leg may import this module; nothing under g4s_amd/ may).

Why this exists. The reference's shipped SpGEMM binary computes C = A·B with oneMKL, a third-party dependency that is not under
/root/reference (mm/Makefile:3-4 names 2022.2.0 / 2023.0.0): `mkl(...)`, mm/inc/mkl_mult.h:40-111, is the call sequence
    mkl_sparse_d_create_csr ×2 (:50,52) → mkl_sparse_spmm (:58) → mkl_sparse_convert_csr (:64) → mkl_sparse_order (:70)
    → mkl_sparse_d_export_csr (:79) → copy out (:85-98) → mkl_sparse_destroy ×3 (:102-106).
mm/inc/mkl_mult.h itself cannot be compiled here: it includes <mkl.h> (absent — the image has MKL's runtime libraries under
/opt/conda/lib but no headers) and utility.h → TBB's <scalable_allocator.h> (absent); writing stand-in headers is not allowed. What IS
here is the library the reference's arithmetic lives in: oneMKL 2021.4 (libmkl_rt.so, conda package mkl-2021.4.0). This module
repeats the reference's call sequence against it, entry point by entry point, with the documented C prototypes declared as ctypes
argtypes, and is used to (a) generate the golden vectors under tests/golden/mkl_*.npz (make_golden.py) that pin the oracle's SpGEMM
and the GPU path to what the reference's own dependency computes, and (b) time the reference's call sequence as a CPU baseline
where the runtime exists. Dense comparison drivers of the reference (mv/mv.c:6-27, mm/src/cblas_dxxmm.c:57-111) call cblas_dsymv /
dtrmv / dspmv / dgemv and cblas_dsymm / dtrmm / dgemm; the same calls are exposed here for the a14 comparison fixtures.

The MKL build differs from the one the reference's Makefile names (2021.4 vs 2022.2): recorded in every fixture as `mkl_version`.
"""
import ctypes as C
import os

import numpy as np

MKL_DIRS = [os.environ.get("G4S_MKL_DIR", ""), "/opt/conda/lib"]

SPARSE_INDEX_BASE_ZERO = 0
SPARSE_OPERATION_NON_TRANSPOSE = 10
SPARSE_STATUS_SUCCESS = 0
CblasRowMajor, CblasColMajor = 101, 102
CblasNoTrans, CblasTrans = 111, 112
CblasUpper, CblasLower = 121, 122
CblasNonUnit, CblasUnit = 131, 132
CblasLeft, CblasRight = 141, 142

_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
vp = C.c_void_p
_mkl = None


def available():
    return any(d and os.path.exists(os.path.join(d, "libmkl_rt.so")) for d in MKL_DIRS)


def load(threading="sequential"):
    """libmkl_rt.so with the LP64 interface (MKL_INT = int32, as mm/inc/define.h:14 assumes). threading: 'sequential' | 'gnu' | 'intel'."""
    global _mkl
    if _mkl is not None:
        return _mkl
    d = next((d for d in MKL_DIRS if d and os.path.exists(os.path.join(d, "libmkl_rt.so"))), None)
    if d is None:
        raise ImportError("libmkl_rt.so not found (looked in $G4S_MKL_DIR and /opt/conda/lib)")
    os.environ.setdefault("MKL_INTERFACE_LAYER", "LP64")
    os.environ.setdefault("MKL_THREADING_LAYER", {"sequential": "SEQUENTIAL", "gnu": "GNU", "intel": "INTEL"}[threading])
    if threading == "intel":
        C.CDLL(os.path.join(d, "libiomp5.so"), mode=C.RTLD_GLOBAL)
    lib = C.CDLL(os.path.join(d, "libmkl_rt.so"), mode=C.RTLD_GLOBAL)
    lib.mkl_sparse_d_create_csr.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.mkl_sparse_spmm.argtypes = [C.c_int, vp, vp, C.POINTER(vp)]
    lib.mkl_sparse_convert_csr.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.mkl_sparse_order.argtypes = [vp]
    lib.mkl_sparse_d_export_csr.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp),
                                            C.POINTER(vp), C.POINTER(vp)]
    lib.mkl_sparse_destroy.argtypes = [vp]
    lib.MKL_Set_Num_Threads.argtypes = [C.c_int]   # the lower-case symbol is the Fortran binding (argument by reference)
    lib.mkl_get_version_string.argtypes = [C.c_char_p, C.c_int]
    d_ = C.c_double
    lib.cblas_dgemv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, d_, _f64, C.c_int, _f64, C.c_int, d_, _f64, C.c_int]
    lib.cblas_dsymv.argtypes = [C.c_int, C.c_int, C.c_int, d_, _f64, C.c_int, _f64, C.c_int, d_, _f64, C.c_int]
    lib.cblas_dtrmv.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f64, C.c_int, _f64, C.c_int]
    lib.cblas_dspmv.argtypes = [C.c_int, C.c_int, C.c_int, d_, _f64, _f64, C.c_int, d_, _f64, C.c_int]
    lib.cblas_dgemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, d_, _f64, C.c_int, _f64, C.c_int, d_, _f64, C.c_int]
    lib.cblas_dsymm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, d_, _f64, C.c_int, _f64, C.c_int, d_, _f64, C.c_int]
    lib.cblas_dtrmm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, d_, _f64, C.c_int, _f64, C.c_int]
    _mkl = lib
    return lib


def version():
    buf = C.create_string_buffer(256)
    load().mkl_get_version_string(buf, 256)
    return buf.value.decode().strip()


def _ok(status, what):
    if status != SPARSE_STATUS_SUCCESS:
        raise RuntimeError(f"{what} returned sparse_status_t {status}")


def mkl_spgemm(A, B, M, K, N, timings=None, threads=None):
    """C = A·B with the reference's call sequence (mm/inc/mkl_mult.h:40-111). A = (arpt, acol, aval) M×K, B K×N, int32 / fp64,
    zero-based. Returns (crpt[M+1], ccol, cval) as fresh numpy arrays (the reference copies out of MKL's memory, :85-98).
    `timings`: dict that receives the seven stage times of mm/inc/Timings.h in milliseconds."""
    import time
    lib = load()
    if threads is not None:
        lib.MKL_Set_Num_Threads(int(threads))
    arpt, acol, aval = (np.ascontiguousarray(A[0], np.int32), np.ascontiguousarray(A[1], np.int32), np.ascontiguousarray(A[2], np.float64))
    brpt, bcol, bval = (np.ascontiguousarray(B[0], np.int32), np.ascontiguousarray(B[1], np.int32), np.ascontiguousarray(B[2], np.float64))
    hA, hB, hC = vp(), vp(), vp()
    clock = time.perf_counter
    t = {}
    t1 = t0 = clock()
    _ok(lib.mkl_sparse_d_create_csr(C.byref(hA), SPARSE_INDEX_BASE_ZERO, M, K, arpt.ctypes.data, arpt.ctypes.data + 4, acol.ctypes.data,
                                    aval.ctypes.data), "mkl_sparse_d_create_csr(A)")                      # :50
    _ok(lib.mkl_sparse_d_create_csr(C.byref(hB), SPARSE_INDEX_BASE_ZERO, K, N, brpt.ctypes.data, brpt.ctypes.data + 4, bcol.ctypes.data,
                                    bval.ctypes.data), "mkl_sparse_d_create_csr(B)")                      # :52
    t["create"] = clock() - t0
    t0 = clock()
    _ok(lib.mkl_sparse_spmm(SPARSE_OPERATION_NON_TRANSPOSE, hA, hB, C.byref(hC)), "mkl_sparse_spmm")      # :58
    t["spmm"] = clock() - t0
    t0 = clock()
    _ok(lib.mkl_sparse_convert_csr(hC, SPARSE_OPERATION_NON_TRANSPOSE, C.byref(hC)), "mkl_sparse_convert_csr")   # :64
    t["convert"] = clock() - t0
    t0 = clock()
    _ok(lib.mkl_sparse_order(hC), "mkl_sparse_order")                                                     # :70
    t["order"] = clock() - t0
    t0 = clock()
    base, rows, cols = C.c_int(), C.c_int(), C.c_int()
    pB, pE, ci, va = vp(), vp(), vp(), vp()
    _ok(lib.mkl_sparse_d_export_csr(hC, C.byref(base), C.byref(rows), C.byref(cols), C.byref(pB), C.byref(pE), C.byref(ci), C.byref(va)),
        "mkl_sparse_d_export_csr")                                                                        # :79
    pe = np.ctypeslib.as_array(C.cast(pE, C.POINTER(C.c_int32)), (M,)) if M else np.zeros(0, np.int32)
    cnnz = int(pe[M - 1]) if M else 0                                                                     # :81
    t["export_csr"] = clock() - t0
    t["total"] = clock() - t1
    crpt = np.empty(M + 1, np.int32)
    if M:
        crpt[:M] = np.ctypeslib.as_array(C.cast(pB, C.POINTER(C.c_int32)), (M,))                          # :93
    crpt[M] = cnnz                                                                                        # :94
    ccol = np.array(np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), (cnnz,))) if cnnz else np.zeros(0, np.int32)
    cval = np.array(np.ctypeslib.as_array(C.cast(va, C.POINTER(C.c_double)), (cnnz,))) if cnnz else np.zeros(0)
    t0 = clock()
    for h in (hC, hB, hA):                                                                                # :102-106
        _ok(lib.mkl_sparse_destroy(h), "mkl_sparse_destroy")
    t["destroy"] = clock() - t0
    t["total"] += t["destroy"]
    if timings is not None:
        timings.update({k: v * 1e3 for k, v in t.items()})
    return crpt, ccol, cval


# ---- the dense comparison drivers (column-major, dim×dim, as the reference calls them)
def dgemv(A, x):          # mv/mv.c:23-27
    dim = len(x); y = np.zeros(dim)
    load().cblas_dgemv(CblasColMajor, CblasNoTrans, dim, dim, 1.0, np.ascontiguousarray(A, np.float64), dim, np.ascontiguousarray(x, np.float64), 1, 0.0, y, 1)
    return y


def dsymv(A, x):          # mv/mv.c:6-10
    dim = len(x); y = np.zeros(dim)
    load().cblas_dsymv(CblasColMajor, CblasUpper, dim, 1.0, np.ascontiguousarray(A, np.float64), dim, np.ascontiguousarray(x, np.float64), 1, 0.0, y, 1)
    return y


def dtrmv(A, x):          # mv/mv.c:12-15 (x := Aᵀ·x, upper triangle, in place)
    dim = len(x); y = np.array(x, np.float64)
    load().cblas_dtrmv(CblasColMajor, CblasUpper, CblasTrans, CblasNonUnit, dim, np.ascontiguousarray(A, np.float64), dim, y, 1)
    return y


def dspmv(AP, x):         # mv/mv.c:17-21 (packed upper triangle)
    dim = len(x); y = np.zeros(dim)
    load().cblas_dspmv(CblasColMajor, CblasUpper, dim, 1.0, np.ascontiguousarray(AP, np.float64), np.ascontiguousarray(x, np.float64), 1, 0.0, y, 1)
    return y


def dgemm(A, B, dim):     # mm/src/cblas_dxxmm.c:96-111
    Cm = np.zeros(dim * dim)
    load().cblas_dgemm(CblasColMajor, CblasNoTrans, CblasNoTrans, dim, dim, dim, 1.0, np.ascontiguousarray(A, np.float64), dim, np.ascontiguousarray(B, np.float64), dim, 0.0, Cm, dim)
    return Cm


def dsymm(A, B, dim):     # mm/src/cblas_dxxmm.c:57-76
    Cm = np.zeros(dim * dim)
    load().cblas_dsymm(CblasColMajor, CblasLeft, CblasUpper, dim, dim, 1.0, np.ascontiguousarray(A, np.float64), dim, np.ascontiguousarray(B, np.float64), dim, 0.0, Cm, dim)
    return Cm


def dtrmm(A, B, dim):     # mm/src/cblas_dxxmm.c:78-95 (B := B·A, upper triangle of A, in place)
    Bm = np.array(B, np.float64)
    load().cblas_dtrmm(CblasColMajor, CblasRight, CblasUpper, CblasNoTrans, CblasNonUnit, dim, dim, 1.0, np.ascontiguousarray(A, np.float64), dim, Bm, dim)
    return Bm
