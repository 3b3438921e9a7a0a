"""One g4s_csr_create of configs[1] (the blocked path's plan build) in a fresh process — for rocprofv3 --kernel-trace --hip-trace --stats. Usage: python tools/plan_create_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from g4s_amd import host
A = bench.build_matrix("rmat", host, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
A.handle
torch.cuda.synchronize()
print(f"first create {1e3 * (time.perf_counter() - t0):.2f} ms, plan bytes {A.info()['plan_bytes']}")
B = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols)
torch.cuda.synchronize()
t0 = time.perf_counter()
B.handle
torch.cuda.synchronize()
print(f"second create {1e3 * (time.perf_counter() - t0):.2f} ms")
