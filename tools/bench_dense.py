#!/usr/bin/env python3
"""DeePMD OptMatmul shape (opt_matmul.cc: xx[M×N]·w[N×K], embedding-net sizes): fp64 MFMA kernel timing."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import capi
lib = capi.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in [(100000, 25, 50), (100000, 50, 100), (100000, 100, 100), (1000000, 100, 100)]:
    xx = torch.rand(M, N, dtype=torch.float64, device="cuda") - 0.5
    w = torch.rand(N, K, dtype=torch.float64, device="cuda") - 0.5
    r = torch.empty(M, K, dtype=torch.float64, device="cuda")
    for _ in range(5):
        lib.g4s_dense_rows_times_matrix(M, N, K, xx.data_ptr(), w.data_ptr(), r.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.g4s_dense_rows_times_matrix(M, N, K, xx.data_ptr(), w.data_ptr(), r.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    ref = xx @ w
    err = (r - ref).abs().max().item()
    e0.record()
    for _ in range(20):
        torch.matmul(xx, w, out=ref)                               # rocBLAS / hipBLASLt beside it (not in the product: a yardstick for the shape)
    e1.record(); torch.cuda.synchronize()
    lib_ms = e0.elapsed_time(e1) / 20
    byt = 8 * (M * N + N * K + M * K)
    print(json.dumps({"M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(2 * M * N * K / ms / 1e9, 2), "GBps": round(byt / ms / 1e6, 1),
                      "frac_hbm_8TBps": round(byt / ms / 1e6 / 8000, 3), "max_abs_diff_vs_torch": err,
                      "vendor_gemm_ms": round(lib_ms, 4), "vendor_gemm_TFLOPs": round(2 * M * N * K / lib_ms / 1e9, 2)}))

    # gradient (_opt_matmul_grad.py): dxx = grad·wᵀ, dw = xxᵀ·grad
    g = torch.rand(M, K, dtype=torch.float64, device="cuda") - 0.5
    dxx = torch.empty(M, N, dtype=torch.float64, device="cuda")
    dw = torch.empty(N, K, dtype=torch.float64, device="cuda")
    for which, (pa, pb) in {"dxx": (dxx.data_ptr(), None), "dw": (None, dw.data_ptr())}.items():
        for _ in range(3):
            lib.g4s_dense_rows_times_matrix_grad(M, N, K, xx.data_ptr(), w.data_ptr(), g.data_ptr(), pa, pb, st)
        e0.record()
        for _ in range(20):
            lib.g4s_dense_rows_times_matrix_grad(M, N, K, xx.data_ptr(), w.data_ptr(), g.data_ptr(), pa, pb, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if which == "dxx":
            err, byt = (dxx - g @ w.T).abs().max().item(), 8 * (M * K + N * K + M * N)
        else:
            err, byt = (dw - xx.T @ g).abs().max().item(), 8 * (M * N + M * K + N * K)
        print(json.dumps({"grad": which, "M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(2 * M * N * K / ms / 1e9, 2),
                          "GBps": round(byt / ms / 1e6, 1), "max_abs_diff_vs_torch": err}))
