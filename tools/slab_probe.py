#!/usr/bin/env python3
"""Per-rank cost of the row-partitioned SpMV on one GPU: time the local product of each of N row slabs of the benchmark matrix (the
exchange is not included). Usage: python tools/slab_probe.py [N=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import capi, dist as gdist, host
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
A = bench.build_matrix("rmat", host, False)
x = host.synth_vector(7, A.cols)
offs = gdist.row_partition(A.rowptr, N)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(M, y):
    for _ in range(5): M.spmv(x, y)
    e0.record()
    for _ in range(50): M.spmv(x, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50
full = t(A, torch.empty(A.rows, dtype=torch.float64, device="cuda"))
print(f"full matrix: {full:.4f} ms")
for r in range(N):
    a, b = offs[r], offs[r + 1]
    rp, ci, va = gdist.slice_rows(A.rowptr, A.colids, A.values, a, b)
    S = host.CSR(rp, ci, va, b - a, A.cols)
    ms = t(S, torch.empty(b - a, dtype=torch.float64, device="cuda"))
    # the same slab with its columns renumbered to the referenced ones only (g4s_amd.dist.CompactExchange)
    ref = torch.unique(ci.long())
    lc = torch.bucketize(ci.long(), ref).to(torch.int32)
    Sc = host.CSR(rp, lc, va, b - a, int(ref.numel()), spmv_flags=(capi.SPMV_BLOCKED if os.environ.get('SLAB_FORCE_BLOCKED') else 0))
    xc = x[ref]
    for _ in range(5): Sc.spmv(xc, ybuf := torch.empty(b - a, dtype=torch.float64, device="cuda"))
    e0.record()
    for _ in range(50): Sc.spmv(xc, ybuf)
    e1.record(); torch.cuda.synchronize()
    msc = e0.elapsed_time(e1) / 50
    print(f"slab {r}: rows {b - a:9d} nnz {S.nnz:10d} path {S.info()['spmv_path']} {ms:.4f} ms ({full / N / ms:.2f} of ideal) | compact columns: {ref.numel():8d} of {A.cols} "
          f"referenced, path {Sc.info()['spmv_path']} {msc:.4f} ms ({full / N / msc:.2f} of ideal), exchange {8 * ref.numel() / 1e6:.0f} MB instead of {8 * A.cols / 1e6:.0f} MB")
    del S, Sc
