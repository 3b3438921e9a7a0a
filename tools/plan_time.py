"""g4s_csr_create (plan build) time per workload, in ms and in products. Usage: python tools/plan_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from g4s_amd import host

for name in ("rmat", "lap7", "banded", "lap5"):
    A = bench.build_matrix(name, host, False)
    x = host.synth_vector(7, A.cols)
    ts = []
    for _ in range(3):
        B = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        B.handle                                                  # lazy g4s_csr_create
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        y = B.spmv(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): B.spmv(x, y)
        torch.cuda.synchronize(); per = (time.perf_counter() - t0) * 1e3 / 20
        del B
    print(f"{name:7s} nnz {A.nnz:>11d}: create {min(ts):8.1f} ms (best of 3; first {ts[0]:.1f}), product {per:.3f} ms -> {min(ts)/per:7.0f} products, path {host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols).info()['spmv_path']}", flush=True)
    del A, x, y
    torch.cuda.empty_cache()
