#!/usr/bin/env python3
"""nnz(C), flop and phase times of C = A·A on R-MAT scale-21 for several edge factors (which one fits int32 crpt?)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402

lib = capi.load()
n = 1 << 21
for ef in [float(a) for a in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    A = host.rmat_csr(n, 21, int(ef * n), 20240522)
    A.values.abs_()
    flop = host.get_flop(A, A)
    crpt = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    cnnz = C.c_int64()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = lib.g4s_spgemm_symbolic(n, n, n, A.rowptr.data_ptr(), A.colids.data_ptr(), A.rowptr.data_ptr(), A.colids.data_ptr(), crpt.data_ptr(),
                                 C.byref(cnnz), None)
    torch.cuda.synchronize()
    t_sym = time.perf_counter() - t0
    line = f"EF={ef}: nnz(A)={A.nnz} flop={flop} nnz(C)={cnnz.value} status={st} symbolic={t_sym * 1e3:.1f} ms"
    if st == 0:
        ccol = torch.empty(cnnz.value, dtype=torch.int32, device="cuda")
        cval = torch.empty(cnnz.value, dtype=torch.float64, device="cuda")
        t0 = time.perf_counter()
        st2 = lib.g4s_spgemm_numeric(n, n, n, A.rowptr.data_ptr(), A.colids.data_ptr(), A.values.data_ptr(), A.rowptr.data_ptr(), A.colids.data_ptr(),
                                     A.values.data_ptr(), crpt.data_ptr(), ccol.data_ptr(), cval.data_ptr(), capi.DEVICE_POINTERS | capi.SORT_OUTPUT, None)
        torch.cuda.synchronize()
        t_num = time.perf_counter() - t0
        line += f" numeric={t_num * 1e3:.1f} ms status={st2} GFLOPS={2 * flop / (t_sym + t_num) / 1e9:.2f}"
        del ccol, cval
    print(line, flush=True)
    del A, crpt
    torch.cuda.empty_cache()
