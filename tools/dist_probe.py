#!/usr/bin/env python3
"""What one GPU can say about the N-rank run of bench.py: for every rank r of an N-way row partition, build the library's distributed handle
(g4s_spmv_dist_create — own-column / remote-column split) on THIS GPU, and time one rank's share of a step with the exchange left out
(g4s_spmv_dist_begin + _finish, empty give lists, x_rem zero). Prints per-rank times, the split, the exchange volume, and the speed-up bound
ideal/max-rank. usage: python tools/dist_probe.py [--workload rmat|lap7] [--ranks 8]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from g4s_amd import capi, dist as gdist, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat")
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--row-weight", type=int, default=1, help="work per row on top of its nonzeros in the equal-work partition (g4s_row_partition)")
args = ap.parse_args()
lib = capi.load()
A = bench.build_matrix(args.workload, host, False)
n = A.rows
x = host.synth_vector(7, n)
y = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


full = timed(lambda: A.spmv(x, y))
print(f"{args.workload}: whole matrix on one GPU {full:.4f} ms (path {A.info()['spmv_path']}), nnz {A.nnz}")
W = args.ranks
offs = gdist.row_partition(A.rowptr, W, row_weight=args.row_weight)
print(f"row partition: work = nnz + {args.row_weight} per row")
worst = 0.0
for r in range(W):
    r0, r1 = offs[r], offs[r + 1]
    rp, ci, va = gdist.slice_rows(A.rowptr, A.colids, A.values, r0, r1)
    h = C.c_void_p()
    o = (C.c_int64 * (W + 1))(*offs)
    torch.cuda.synchronize()
    capi.check(lib.g4s_spmv_dist_create(C.byref(h), r, W, o, n, host._ptr(rp), host._ptr(ci), host._ptr(va), capi.DEVICE_POINTERS))
    for k in range(W):
        if k != r:
            capi.check(lib.g4s_spmv_dist_set_give(h, k, 0, None, 0))
    info = capi.DistInfo()
    capi.check(lib.g4s_spmv_dist_get_info(h, C.byref(info)))
    xl, yl = x[r0:r1].contiguous(), torch.empty(r1 - r0, dtype=torch.float64, device="cuda")

    def step():
        capi.check(lib.g4s_spmv_dist_begin(h, host._ptr(xl), host._ptr(yl), host._stream()))
        capi.check(lib.g4s_spmv_dist_finish(h, host._ptr(yl), host._stream()))
    ms = timed(step)
    worst = max(worst, ms)
    print(f"  rank {r}: rows {r1 - r0:9d} nnz own {info.nnz_own:10d} (path {info.own_path}) + remote {info.nnz_rem:10d} (path {info.rem_path}), referenced remote columns "
          f"{info.n_ref:8d} = {info.recv_bytes / 1e6:6.1f} MB per step: {ms:.4f} ms = {full / W / ms:.2f} of ideal")
    lib.g4s_spmv_dist_destroy(h)
    del rp, ci, va
print(f"slowest rank {worst:.4f} ms -> at most {full / worst:.2f}x over one GPU at {W} ranks before the exchange (which overlaps the own-column product)")
