"""ctypes view of oracle/liboracle.so and oracle/_ref/libref_graph.so — TEST INFRASTRUCTURE ONLY.

Nothing under g4s_amd/ may import this module (tests/test_no_oracle_in_product.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.environ.get("G4S_ORACLE_SO") or os.path.join(ROOT, "oracle", "liboracle.so")   # G4S_ORACLE_SO: the ASan/UBSan build (make -C oracle asan)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_graph.so")

FUN_GATHER = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_double), C.POINTER(C.c_double))
FUN_APPLY = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_double), C.POINTER(C.c_double))

_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i8 = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
vp = C.c_void_p


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.oracle_spmv_csr.argtypes = [C.c_int32, _i32, _i32, _f64, _f64, _f64, C.c_double, C.c_double]
        L.oracle_spmv_csr_ld.argtypes = [C.c_int32, _i32, _i32, _f64, _f64, _f64, _f64]
        L.oracle_spmv_csr_mt.argtypes = [C.c_int32, _i32, _i32, _f64, _f64, _f64, C.c_double, C.c_double, C.c_int]
        L.oracle_spmv_csr_mt.restype = C.c_int
        L.oracle_spgemm_flop.argtypes = [C.c_int32, _i32, _i32, _i32, vp]
        L.oracle_spgemm_flop.restype = C.c_int64
        L.oracle_bin_id.argtypes = [C.c_int32, C.c_int32, _i64, _i8]
        L.oracle_rows_offset.argtypes = [C.c_int32, _i64, C.c_int32, _i32]
        L.oracle_spgemm_symbolic.argtypes = [C.c_int32, C.c_int32, _i32, _i32, _i32, _i32, _i32]
        L.oracle_spgemm_symbolic.restype = C.c_int64
        L.oracle_spgemm_numeric.argtypes = [C.c_int32, C.c_int32, _i32, _i32, _f64, _i32, _i32, _f64, _i32, _i32, _f64, C.c_int]
        L.oracle_spgemm_outer.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, _i32, _i32, _f64, C.c_int32, _i32, vp, vp]
        L.oracle_spgemm_outer.restype = C.c_int64
        L.oracle_spgemm_heap.argtypes = [C.c_int32, C.c_int32, _i32, _i32, _f64, _i32, _i32, _f64, _i32, vp, vp]
        L.oracle_spgemm_heap.restype = C.c_int64
        L.oracle_mtx_read.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), vp, vp, vp]
        L.oracle_mtx_read.restype = C.c_int
        L.oracle_spmm_dense.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, vp, FUN_GATHER, FUN_APPLY, vp, C.c_int]
        L.oracle_element_matvec.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, vp, C.c_int32, _f64, _f64, C.c_int32]
        L.oracle_dense_rows_times_matrix.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, _f64, _f64]
        L.oracle_construct_node_ks.restype = C.c_int
        L.oracle_construct_node_ks.argtypes = [C.c_int32, C.c_int32, _i32, _i32, C.c_int32, C.c_int32, C.c_int32, _i32, _f64, _f64, _f64, _f64, _f64]
        L.oracle_n_assemble_del2_u.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, _f64, _f64, _f64, _f64, vp, C.c_int32]
        L.oracle_assemble_div_u.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, _f64, _f64]
        L.oracle_assemble_grad_p.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, C.c_int32, vp, C.c_int32, _f64, _f64]
        L.oracle_build_diagonal_of_Ahat.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, _f64, _f64]
        L.oracle_solve_Ahat_p_fhat_CG.restype = C.c_double
        L.oracle_solve_Ahat_p_fhat_CG.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, C.c_int32, C.c_int32, _f64, _f64, _f64, _f64, _f64, _f64,
                                                  C.c_double, vp, C.c_int32, _f64, _f64, _f64, C.c_double, C.c_double, C.c_double, C.c_int32,
                                                  C.c_int32, C.c_int32, C.POINTER(C.c_int32), vp, C.POINTER(C.c_int64)]
        L.oracle_dense_rows_times_matrix_grad.argtypes = [C.c_int32, C.c_int32, C.c_int32, _f64, _f64, _f64, _f64, _f64]
        L.oracle_sym_quadratic_form.argtypes = [C.c_int32, C.c_int32, _f64, _f64, vp, _f64]
        L.oracle_element_inverse_diagonal.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, _f64, C.c_int32]
        L.oracle_conj_grad_elem.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i32, _i32, _f64, C.c_int32, _f64, vp, C.c_int32, _f64, _f64,
                                            C.c_double, C.POINTER(C.c_int32), vp]
        L.oracle_conj_grad_elem.restype = C.c_double
        L.oracle_mix64.argtypes = [C.c_uint64]
        L.oracle_mix64.restype = C.c_uint64
        L.oracle_entry_value.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64]
        L.oracle_entry_value.restype = C.c_double
        L.oracle_vector_value.argtypes = [C.c_uint64, C.c_int64]
        L.oracle_vector_value.restype = C.c_double
        L.oracle_rmat_edges.argtypes = [C.c_uint64, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _i64]
        L.oracle_laplacian5.argtypes = [C.c_int32, C.c_int32, vp, vp, vp]
        L.oracle_laplacian5.restype = C.c_int64
        L.oracle_laplacian7_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, vp, vp, vp]
        L.oracle_laplacian7_rows.restype = C.c_int64
        L.oracle_banded.argtypes = [C.c_int32, C.c_int32, C.c_uint64, vp, vp, vp]
        L.oracle_banded.restype = C.c_int64

    # ---- SpMV
    def spmv(self, rowptr, colids, values, x, y=None, alpha=1.0, beta=0.0):
        rows = len(rowptr) - 1
        y = np.zeros(rows) if y is None else np.array(y, dtype=np.float64)
        self.lib.oracle_spmv_csr(rows, rowptr, colids, values, np.ascontiguousarray(x, np.float64), y, alpha, beta)
        return y

    def spmv_ld(self, rowptr, colids, values, x):
        rows = len(rowptr) - 1
        y, a = np.zeros(rows), np.zeros(rows)
        self.lib.oracle_spmv_csr_ld(rows, rowptr, colids, values, np.ascontiguousarray(x, np.float64), y, a)
        return y, a

    def spmv_mt(self, rowptr, colids, values, x, y, threads):
        return self.lib.oracle_spmv_csr_mt(len(rowptr) - 1, rowptr, colids, values, x, y, 1.0, 0.0, threads)

    def spmv_mt_y(self, rowptr, colids, values, x, threads=8):
        """y = A·x with the rows spread over host threads; every row is still summed left to right by one thread, so the result is
        the single-thread oracle's bit for bit (used where the matrix has 1e8 nonzeros)."""
        y = np.zeros(len(rowptr) - 1)
        self.spmv_mt(np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colids, np.int32), np.ascontiguousarray(values, np.float64),
                     np.ascontiguousarray(x, np.float64), y, threads)
        return y

    # ---- SpGEMM
    def flop(self, arpt, acol, brpt, per_row=False):
        M = len(arpt) - 1
        rf = np.zeros(max(M, 1), np.int64)
        tot = self.lib.oracle_spgemm_flop(M, arpt, acol, brpt, rf.ctypes.data)
        return (tot, rf[:M]) if per_row else tot

    def bin_id(self, cols, row_flop):
        b = np.zeros(max(len(row_flop), 1), np.int8)
        self.lib.oracle_bin_id(len(row_flop), cols, np.ascontiguousarray(row_flop, np.int64), b)
        return b[:len(row_flop)]

    def rows_offset(self, row_work, parts):
        off = np.zeros(parts + 1, np.int32)
        self.lib.oracle_rows_offset(len(row_work), np.ascontiguousarray(row_work, np.int64), parts, off)
        return off

    def spgemm(self, A, B, N, sort_output=True):
        """A, B = (rowptr, colids, values); returns (crpt, ccol, cval)."""
        arpt, acol, aval = A
        brpt, bcol, bval = B
        M = len(arpt) - 1
        crpt = np.zeros(M + 1, np.int32)
        nnz = self.lib.oracle_spgemm_symbolic(M, N, arpt, acol, brpt, bcol, crpt)
        assert nnz >= 0, "nnz(C) overflows int32"
        ccol = np.zeros(max(nnz, 1), np.int32)
        cval = np.zeros(max(nnz, 1), np.float64)
        self.lib.oracle_spgemm_numeric(M, N, arpt, acol, aval, brpt, bcol, bval, crpt, ccol, cval, 1 if sort_output else 0)
        return crpt, ccol[:nnz], cval[:nnz]

    # ---- MatrixMarket
    def spgemm_outer(self, A, B, K, N, nblockers=4):
        """OuterSpGEMM restated (mm/inc/outer_mult.h:271-542) — a cross-check of spgemm(), never a GPU path."""
        M = len(A[0]) - 1
        crpt = np.zeros(M + 1, np.int32)
        nnz = self.lib.oracle_spgemm_outer(M, K, N, A[0], A[1], A[2], B[0], B[1], B[2], nblockers, crpt, None, None)
        ccol, cval = np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.oracle_spgemm_outer(M, K, N, A[0], A[1], A[2], B[0], B[1], B[2], nblockers, crpt, ccol.ctypes.data, cval.ctypes.data)
        return crpt, ccol, cval

    def spgemm_heap(self, A, B, N):
        """HeapSpGEMM restated row-wise (mm/inc/heap_mult.h:47-223) — a cross-check of spgemm(), never a GPU path."""
        M = len(A[0]) - 1
        crpt = np.zeros(M + 1, np.int32)
        nnz = self.lib.oracle_spgemm_heap(M, N, A[0], A[1], A[2], B[0], B[1], B[2], crpt, None, None)
        ccol, cval = np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.oracle_spgemm_heap(M, N, A[0], A[1], A[2], B[0], B[1], B[2], crpt, ccol.ctypes.data, cval.ctypes.data)
        return crpt, ccol, cval

    def mtx_read(self, path):
        r, c, n = C.c_int32(), C.c_int32(), C.c_int64()
        st = self.lib.oracle_mtx_read(path.encode(), C.byref(r), C.byref(c), C.byref(n), None, None, None)
        if st != 0:
            raise ValueError(f"oracle_mtx_read failed: {st}")
        rowptr = np.zeros(r.value + 1, np.int32)
        colids = np.zeros(max(n.value, 1), np.int32)
        values = np.zeros(max(n.value, 1), np.float64)
        st = self.lib.oracle_mtx_read(path.encode(), C.byref(r), C.byref(c), C.byref(n), rowptr.ctypes.data, colids.ctypes.data,
                                      values.ctypes.data)
        assert st == 0
        return r.value, c.value, rowptr, colids[:n.value], values[:n.value]

    # ---- graph patterns
    def element_matvec(self, ien, idmap, elt_k, u, neq, npe=8, dof=3, base=0):
        """elt_k: (nel, n*n) array; base=1 prepends an unused row pointer as CitcomS does."""
        nel = elt_k.shape[0]
        rows = (C.POINTER(C.c_double) * (nel + base))()
        for e in range(nel):
            rows[e + base] = elt_k[e].ctypes.data_as(C.POINTER(C.c_double))
        Au = np.zeros(neq)
        self.lib.oracle_element_matvec(nel, npe, dof, np.ascontiguousarray(ien, np.int32).ravel(),
                                       np.ascontiguousarray(idmap, np.int32).ravel(), C.cast(rows, vp), base,
                                       np.ascontiguousarray(u, np.float64), Au, neq)
        return Au

    def element_inverse_diagonal(self, ien, idmap, elt_k, neq, npe=8, dof=3):
        BI = np.zeros(neq)
        self.lib.oracle_element_inverse_diagonal(elt_k.shape[0], npe, dof, np.ascontiguousarray(ien, np.int32).ravel(),
                                                 np.ascontiguousarray(idmap, np.int32).ravel(), np.ascontiguousarray(elt_k).ravel(), BI, neq)
        return BI

    def conj_grad_elem(self, ien, idmap, elt_k, neq, BI, zero_resid, F, acc, steps, npe=8, dof=3):
        d0 = np.zeros(neq)
        cyc = C.c_int32(steps)
        hist = np.zeros(max(steps, 1))
        zr = np.ascontiguousarray(zero_resid, np.int32)
        res = self.lib.oracle_conj_grad_elem(elt_k.shape[0], npe, dof, np.ascontiguousarray(ien, np.int32).ravel(),
                                             np.ascontiguousarray(idmap, np.int32).ravel(), np.ascontiguousarray(elt_k).ravel(), neq,
                                             np.ascontiguousarray(BI), zr.ctypes.data if len(zr) else None, len(zr), np.ascontiguousarray(F), d0,
                                             acc, C.byref(cyc), hist.ctypes.data)
        return d0, cyc.value, res, hist[:cyc.value]

    # ---- node-assembled operator (Construct_arrays.c, n_assemble_del2_u)
    def construct_node_ks(self, ien, idmap, nno, neq, node_map, elt_k, bcw, npe=8):
        ie, idm = np.ascontiguousarray(ien, np.int32).ravel(), np.ascontiguousarray(idmap, np.int32).ravel()
        max_eqn = node_map.shape[1]
        k1, k2, k3 = (np.zeros(nno * max_eqn) for _ in range(3))
        rc = self.lib.oracle_construct_node_ks(len(ien), npe, ie, idm, nno, neq, max_eqn, np.ascontiguousarray(node_map, np.int32).ravel(),
                                               np.ascontiguousarray(elt_k).ravel(), np.ascontiguousarray(bcw, np.float64).ravel(), k1, k2, k3)
        assert rc == 0, "a Node_map slot is missing"
        return k1, k2, k3

    def n_assemble_del2_u(self, nno, neq, node_map, idmap, k1, k2, k3, u, zero_resid):
        zr = np.ascontiguousarray(zero_resid, np.int32)
        uu, Au = np.concatenate([np.asarray(u, np.float64), [0.0]]), np.zeros(neq + 1)
        self.lib.oracle_n_assemble_del2_u(nno, neq, node_map.shape[1], np.ascontiguousarray(node_map, np.int32).ravel(),
                                          np.ascontiguousarray(idmap, np.int32).ravel(), k1, k2, k3, uu, Au, zr.ctypes.data if len(zr) else None, len(zr))
        return Au[:neq]

    # ---- Uzawa iteration (Stokes_flow_Incomp.c) and its operators
    @staticmethod
    def _mesh(ien, idmap):
        return np.ascontiguousarray(ien, np.int32).ravel(), np.ascontiguousarray(idmap, np.int32).ravel()

    def assemble_div_u(self, ien, idmap, g, U, npe=8, dof=3):
        ie, idm = self._mesh(ien, idmap)
        out = np.zeros(len(ien))
        self.lib.oracle_assemble_div_u(len(ien), npe, dof, ie, idm, np.ascontiguousarray(g).ravel(), np.ascontiguousarray(U), out)
        return out

    def assemble_grad_p(self, ien, idmap, g, neq, zero_resid, P, npe=8, dof=3):
        ie, idm = self._mesh(ien, idmap)
        zr = np.ascontiguousarray(zero_resid, np.int32)
        out = np.zeros(neq)
        self.lib.oracle_assemble_grad_p(len(ien), npe, dof, ie, idm, np.ascontiguousarray(g).ravel(), neq, zr.ctypes.data if len(zr) else None, len(zr),
                                        np.ascontiguousarray(P), out)
        return out

    def build_diagonal_of_Ahat(self, ien, idmap, g, BI, npe=8, dof=3):
        ie, idm = self._mesh(ien, idmap)
        out = np.zeros(len(ien))
        self.lib.oracle_build_diagonal_of_Ahat(len(ien), npe, dof, ie, idm, np.ascontiguousarray(g).ravel(), np.ascontiguousarray(BI), out)
        return out

    def solve_Ahat_p_fhat_CG(self, ien, idmap, nno, neq, elt_k, g, BI, BPI, nmass, area, volume, zero_resid, F, V, P, imp, inner_scale, v_res,
                             v_steps_low, steps_max, check_continuity=0, check_pressure=0, npe=8, dof=3):
        """Returns (V, P, outer count, incompressibility, hist[count+1, 5], inner iterations); V and P are copies."""
        ie, idm = self._mesh(ien, idmap)
        zr = np.ascontiguousarray(zero_resid, np.int32)
        V, P = np.array(V, dtype=np.float64), np.array(P, dtype=np.float64)
        steps = C.c_int32(steps_max)
        hist = np.zeros((steps_max + 1, 5))
        inner = C.c_int64(0)
        inc = self.lib.oracle_solve_Ahat_p_fhat_CG(len(ien), npe, dof, ie, idm, nno, neq, np.ascontiguousarray(elt_k).ravel(), np.ascontiguousarray(g).ravel(),
                                                   np.ascontiguousarray(BI), np.ascontiguousarray(BPI), np.ascontiguousarray(nmass),
                                                   np.ascontiguousarray(area), volume, zr.ctypes.data if len(zr) else None, len(zr),
                                                   np.ascontiguousarray(F), V, P, imp, inner_scale, v_res, v_steps_low, check_continuity, check_pressure,
                                                   C.byref(steps), hist.ctypes.data, C.byref(inner))
        return V, P, steps.value, inc, hist[:steps.value + 1], inner.value

    def dense_rows_times_matrix(self, xx, w):
        M, N = xx.shape
        K = w.shape[1]
        rows = (C.POINTER(C.c_double) * max(M, 1))()
        for e in range(M):
            rows[e] = xx[e].ctypes.data_as(C.POINTER(C.c_double))
        res = np.zeros((M, K))
        self.lib.oracle_dense_rows_times_matrix(M, N, K, C.cast(rows, vp), np.ascontiguousarray(w), res)
        return res

    def dense_rows_times_matrix_grad(self, xx, w, grad):
        M, N = xx.shape
        K = w.shape[1]
        dxx, dw = np.zeros((M, N)), np.zeros((N, K))
        self.lib.oracle_dense_rows_times_matrix_grad(M, N, K, np.ascontiguousarray(xx), np.ascontiguousarray(w), np.ascontiguousarray(grad), dxx, dw)
        return dxx, dw

    def sym_quadratic_form(self, m, numbers, a, x, b=None):
        res = np.zeros(2)
        bp = None if b is None else np.ascontiguousarray(b, np.float64).ctypes.data
        self.lib.oracle_sym_quadratic_form(m, numbers, np.ascontiguousarray(a, np.float64), np.ascontiguousarray(x, np.float64), bp, res)
        return res

    # ---- generators
    def rmat_keys(self, seed, scale, n, e0, count):
        keys = np.zeros(count, np.int64)
        self.lib.oracle_rmat_edges(seed, scale, n, e0, count, keys)
        return keys

    def rmat_csr(self, seed, scale, n, edges):
        keys = np.unique(self.rmat_keys(seed, scale, n, 0, edges))
        rows, cols = keys // n, keys % n
        rowptr = np.zeros(n + 1, np.int64)
        np.add.at(rowptr, rows + 1, 1)
        rowptr = np.cumsum(rowptr).astype(np.int32)
        values = np.array([self.lib.oracle_entry_value(seed, int(i), int(j), n) for i, j in zip(rows, cols)], dtype=np.float64)
        return rowptr, cols.astype(np.int32), values

    def vector(self, seed, count, i0=0):
        return np.array([self.lib.oracle_vector_value(seed, i0 + i) for i in range(count)], dtype=np.float64)

    def laplacian5(self, nx, ny):
        nnz = self.lib.oracle_laplacian5(nx, ny, None, None, None)
        rp, ci, va = np.zeros(nx * ny + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.oracle_laplacian5(nx, ny, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        return rp, ci, va

    def laplacian7(self, nx, ny, nz, r0=0, r1=None):
        r1 = nx * ny * nz if r1 is None else r1
        nnz = self.lib.oracle_laplacian7_rows(nx, ny, nz, r0, r1, None, None, None)
        rp, ci, va = np.zeros(r1 - r0 + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.oracle_laplacian7_rows(nx, ny, nz, r0, r1, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        return rp, ci, va

    def banded(self, n, hb, seed):
        nnz = self.lib.oracle_banded(n, hb, seed, None, None, None)
        rp, ci, va = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        self.lib.oracle_banded(n, hb, seed, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        return rp, ci, va


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _oracle = Oracle(C.CDLL(ORACLE_SO))
    return _oracle


def load_ref():
    """The reference's own GraphProcess (compiled in place by oracle/Makefile). None if it was never built."""
    if not os.path.exists(REF_SO):
        return None
    lib = C.CDLL(REF_SO)
    lib.ref_graph_process_cb.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, FUN_GATHER, FUN_APPLY]
    lib.ref_graph_process_dense.argtypes = [C.c_int, C.c_int, C.c_int, _f64, _f64, _f64]
    return lib
