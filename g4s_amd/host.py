"""Host-side mirror of the reference's interface for the hot path, over device-resident buffers.

Names and argument meaning follow the reference so that tests read like tests of the reference would:
  CSR            — mm/inc/CSR.h:22-100 (rows, cols, nnz, rowptr, colids, values, zerobased)
  HashSpGEMM     — mm/inc/hash_mult.h:1028-1057 (sortOutput flag; multiply/add are the arithmetic semiring only)
  spmv           — the CSR mat-vec the build defines for mv/ (DESIGN.md §SpMV), y = alpha·A·x + beta·y
Everything here calls the C-ABI (libg4s_hip.so); torch tensors only hold device memory. No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("g4s_amd.host needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")


class CSR:
    """Device-resident CSR<int32, fp64> with the reference's field names (mm/inc/CSR.h:92-99)."""

    def __init__(self, rowptr, colids, values, rows, cols, spmv_flags=0):
        _require_gpu()
        assert rowptr.dtype == torch.int32 and colids.dtype == torch.int32 and values.dtype == torch.float64
        assert rowptr.is_cuda and colids.is_cuda and values.is_cuda
        assert rowptr.numel() == rows + 1 and colids.numel() == values.numel()
        self.rows, self.cols, self.nnz = int(rows), int(cols), int(colids.numel())
        self.rowptr, self.colids, self.values = rowptr.contiguous(), colids.contiguous(), values.contiguous()
        self.zerobased = True
        self._spmv_flags = spmv_flags
        self._handle = None

    @classmethod
    def from_host(cls, rowptr, colids, values, rows, cols, device="cuda", **kw):
        _require_gpu()
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
        return cls(t(rowptr, np.int32), t(colids, np.int32), t(values, np.float64), rows, cols, **kw)

    def to_host(self):
        return self.rowptr.cpu().numpy(), self.colids.cpu().numpy(), self.values.cpu().numpy()

    # -- SpMV plan handle (g4s_csr_create with borrowed device arrays)
    @property
    def handle(self):
        if self._handle is None:
            lib = capi.load()
            h = C.c_void_p()
            # g4s_csr_create reads the borrowed arrays on the NULL stream: whatever torch stream filled them must be complete first
            torch.cuda.current_stream().synchronize()
            capi.check(lib.g4s_csr_create(C.byref(h), self.rows, self.cols, self.nnz, _ptr(self.rowptr), _ptr(self.colids),
                                          _ptr(self.values), capi.DEVICE_POINTERS | self._spmv_flags))
            self._handle = h
        return self._handle

    def info(self):
        inf = capi.CsrInfo()
        capi.check(capi.load().g4s_csr_get_info(self.handle, C.byref(inf)))
        return {n: getattr(inf, n) for n, _ in capi.CsrInfo._fields_}

    def spmv(self, x, y=None, alpha=1.0, beta=0.0):
        """y = alpha·A·x + beta·y on the current torch stream (asynchronous)."""
        assert x.dtype == torch.float64 and x.is_cuda and x.numel() == self.cols
        if y is None:
            assert beta == 0.0
            y = torch.empty(self.rows, dtype=torch.float64, device=x.device)
        assert y.dtype == torch.float64 and y.numel() == self.rows
        capi.check(capi.load().g4s_spmv(self.handle, _ptr(x), _ptr(y), float(alpha), float(beta), _stream()))
        return y

    def update_values(self, values):
        """New values for the same pattern (g4s_csr_update_values), in place: `values` (device tensor, nnz doubles) replaces self.values — the handle borrows
        the tensor's memory from here on — and the plan's own copy is refreshed on the current torch stream."""
        assert values.dtype == torch.float64 and values.is_cuda and values.numel() == self.nnz
        self.values = values
        if self._handle is not None:
            capi.check(capi.load().g4s_csr_update_values(self._handle, _ptr(values), capi.DEVICE_POINTERS, _stream()))

    def close(self):
        if self._handle is not None:
            capi.load().g4s_csr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def spmv(A, x, y=None, alpha=1.0, beta=0.0):
    return A.spmv(x, y, alpha, beta)


def get_flop(A, B):
    """Σ_i Σ_{j∈A(i,:)} nnz(B(acol_j,:)) — get_flop, mm/inc/hash_mult.h:46-62; compute_flop, mkl_mult.h:8-38."""
    flop = C.c_int64(0)
    capi.check(capi.load().g4s_spgemm_flop(A.rows, _ptr(A.rowptr), _ptr(A.colids), _ptr(B.rowptr), C.byref(flop), C.c_void_p(0),
                                           capi.DEVICE_POINTERS))
    return flop.value


class _LibraryBuffer:
    """A device array the library allocated for a callee-allocated output (mkl_mult.h:90-92: outputs are allocated by the callee and
    freed by the caller). torch views it without a copy through __cuda_array_interface__ and keeps this object alive; the memory goes
    back through g4s_dev_free when the last view is gone."""

    def __init__(self, ptr, count, typestr):
        self._ptr, self._count, self._typestr = int(ptr or 0), int(count), typestr

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self._count,), "typestr": self._typestr, "data": (self._ptr, False), "version": 2}

    def __del__(self):
        try:
            if self._ptr:
                capi.load().g4s_dev_free(C.c_void_p(self._ptr))
        except Exception:
            pass


class _BorrowedBuffer:
    """A device array owned by a library object (e.g. the vectors of a g4s_cg_ws_t), viewed by torch without a copy."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def view_f64(ptr, count, device="cuda"):
    """torch view of `count` doubles at a device pointer the caller keeps alive."""
    return torch.as_tensor(_BorrowedBuffer(ptr.value if hasattr(ptr, "value") else ptr, count, "<f8"), device=device)


def view_i32(ptr, count, device="cuda"):
    """torch view of `count` int32 at a device pointer the caller keeps alive."""
    return torch.as_tensor(_BorrowedBuffer(ptr.value if hasattr(ptr, "value") else ptr, count, "<i4"), device=device)


def _view(ptr, count, typestr, dtype, device):
    if count == 0:
        if ptr:
            capi.load().g4s_dev_free(ptr)
        return torch.empty(0, dtype=dtype, device=device)
    return torch.as_tensor(_LibraryBuffer(ptr.value if hasattr(ptr, "value") else ptr, count, typestr), device=device)


def HashSpGEMM(a, b, sortOutput=True, two_phase=False):
    """C = A·B — HashSpGEMM, mm/inc/hash_mult.h:1028-1057. One call into the library (symbolic → crpt, numeric → ccol/cval, the sorted
    columns of the large rows carried from the first phase to the second); `two_phase=True` issues g4s_spgemm_symbolic and
    g4s_spgemm_numeric separately into torch-owned arrays. The result's `timings` holds the library's stage times (one-call form)."""
    _require_gpu()
    assert a.cols == b.rows
    lib = capi.load()
    dev = a.rowptr.device
    if not two_phase:
        crpt_p, ccol_p, cval_p = C.c_void_p(), C.c_void_p(), C.c_void_p()
        cnnz, tm = C.c_int64(0), capi.Timings()
        flags = capi.DEVICE_POINTERS | (capi.SORT_OUTPUT if sortOutput else 0)
        capi.check(lib.g4s_spgemm_csr_i32_f64(_ptr(a.rowptr), _ptr(a.colids), _ptr(a.values), _ptr(b.rowptr), _ptr(b.colids), _ptr(b.values),
                                              C.byref(crpt_p), C.byref(ccol_p), C.byref(cval_p), a.rows, a.cols, b.cols, C.byref(cnnz), C.byref(tm), flags))
        c = CSR(_view(crpt_p, a.rows + 1, "<i4", torch.int32, dev), _view(ccol_p, cnnz.value, "<i4", torch.int32, dev),
                _view(cval_p, cnnz.value, "<f8", torch.float64, dev), a.rows, b.cols)
        c.timings = {n: getattr(tm, n) for n, _ in capi.Timings._fields_}
        return c
    crpt = torch.empty(a.rows + 1, dtype=torch.int32, device=dev)
    cnnz = C.c_int64(0)
    capi.check(lib.g4s_spgemm_symbolic(a.rows, a.cols, b.cols, _ptr(a.rowptr), _ptr(a.colids), _ptr(b.rowptr), _ptr(b.colids),
                                       _ptr(crpt), C.byref(cnnz), _stream()))
    ccol = torch.empty(cnnz.value, dtype=torch.int32, device=dev)
    cval = torch.empty(cnnz.value, dtype=torch.float64, device=dev)
    flags = capi.DEVICE_POINTERS | (capi.SORT_OUTPUT if sortOutput else 0)
    capi.check(lib.g4s_spgemm_numeric(a.rows, a.cols, b.cols, _ptr(a.rowptr), _ptr(a.colids), _ptr(a.values),
                                      _ptr(b.rowptr), _ptr(b.colids), _ptr(b.values), _ptr(crpt), _ptr(ccol), _ptr(cval),
                                      flags, _stream()))
    return CSR(crpt, ccol, cval, a.rows, b.cols)


# ------------------------------------------------------------------------------------------------ synthetic inputs
def rmat_csr(n, scale, edges, seed, device="cuda", chunk=1 << 26, **kw):
    """R-MAT (0.57,0.19,0.19,0.05) on n vertices: `edges` draws, duplicates merged, rows sorted by column, values
    U(−1,1) of (seed,row,col) — SURVEY.md §8d C2. Built entirely in HBM (generator kernels + torch sort/unique)."""
    _require_gpu()
    lib = capi.load()
    keys = torch.empty(edges, dtype=torch.int64, device=device)
    for e0 in range(0, edges, chunk):
        cnt = min(chunk, edges - e0)
        capi.check(lib.g4s_synth_rmat_keys(seed, scale, n, e0, cnt, C.c_void_p(keys.data_ptr() + 8 * e0), _stream()))
    keys = torch.unique(keys, sorted=True)
    nnz = keys.numel()
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=device)
    colids = torch.empty(nnz, dtype=torch.int32, device=device)
    values = torch.empty(nnz, dtype=torch.float64, device=device)
    capi.check(lib.g4s_synth_csr_from_keys(seed, n, n, _ptr(keys), nnz, _ptr(rowptr), _ptr(colids), _ptr(values), _stream()))
    torch.cuda.synchronize()
    del keys
    return CSR(rowptr, colids, values, n, n, **kw)


def synth_vector(seed, count, i0=0, device="cuda"):
    _require_gpu()
    x = torch.empty(count, dtype=torch.float64, device=device)
    capi.check(capi.load().g4s_synth_vector(seed, i0, count, _ptr(x), _stream()))
    return x


def laplacian_csr(kind, nx, ny, nz=1, r0=0, r1=None, device="cuda", **kw):
    """Rows [r0,r1) of the 5-point (kind=5, nz=1, diag 4) or 7-point (kind=7, diag 6) Laplacian, global columns."""
    _require_gpu()
    lib = capi.load()
    n = nx * ny * nz
    r1 = n if r1 is None else r1
    m = r1 - r0
    counts = torch.empty(m, dtype=torch.int32, device=device)
    capi.check(lib.g4s_synth_laplacian_rows(kind, nx, ny, nz, r0, r1, _ptr(counts), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), 0,
                                            _stream()))
    rowptr = torch.zeros(m + 1, dtype=torch.int32, device=device)
    rowptr[1:] = torch.cumsum(counts, 0, dtype=torch.int64).to(torch.int32)
    nnz = int(rowptr[-1].item())
    colids = torch.empty(nnz, dtype=torch.int32, device=device)
    values = torch.empty(nnz, dtype=torch.float64, device=device)
    capi.check(lib.g4s_synth_laplacian_rows(kind, nx, ny, nz, r0, r1, C.c_void_p(0), _ptr(rowptr), _ptr(colids), _ptr(values), 1,
                                            _stream()))
    torch.cuda.synchronize()
    return CSR(rowptr, colids, values, m, n, **kw)


def banded_csr(n, hb, seed, device="cuda", **kw):
    _require_gpu()
    full = n * (2 * hb + 1)
    nnz = full - hb * (hb + 1)  # each side loses 1+2+…+hb entries
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=device)
    colids = torch.empty(nnz, dtype=torch.int32, device=device)
    values = torch.empty(nnz, dtype=torch.float64, device=device)
    capi.check(capi.load().g4s_synth_banded(n, hb, seed, _ptr(rowptr), _ptr(colids), _ptr(values), _stream()))
    torch.cuda.synchronize()
    return CSR(rowptr, colids, values, n, n, **kw)
