#!/usr/bin/env python3
"""A/B of the blocked SpMV's launch forms on BASELINE configs[1] in ONE process: the three-launch form (G4S_PB_FUSED=0) against the one-launch
persistent form with different row-group counts / lags (read from the environment when the plan is built). Interleaved rounds, median and min.
usage: python tools/sweep_fused.py [--variants legacy,16:auto,8:auto,24:auto,16:2] [--rounds 5] [--iters 40] [--small]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="legacy,1:auto,8:auto,16:auto,24:auto,32:auto,16:1,16:2,16:6")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--small", action="store_true")
ap.add_argument("--env", default="", help="extra KEY=VAL,KEY=VAL applied to every variant")
args = ap.parse_args()

A0 = bench.build_matrix("rmat", host, args.small)
x = host.synth_vector(7, A0.cols)
x2 = host.synth_vector(11, A0.cols)
handles = {}
for kv in filter(None, args.env.split(",")):
    k, v = kv.split("=")
    os.environ[k] = v
for v in args.variants.split(","):
    for k in ("G4S_PB_FUSED", "G4S_PB_GROUPS", "G4S_PB_LAG"):
        os.environ.pop(k, None)
    if v == "legacy":
        os.environ["G4S_PB_FUSED"] = "0"
    else:
        g, lag = v.split(":")
        os.environ["G4S_PB_GROUPS"] = g
        if lag != "auto":
            os.environ["G4S_PB_LAG"] = lag
    os.environ["G4S_DEBUG"] = "1"
    handles[v] = host.CSR(A0.rowptr, A0.colids, A0.values, A0.rows, A0.cols, spmv_flags=capi.SPMV_BLOCKED)
    handles[v].handle                                               # the plan is built lazily: build it while this variant's environment is set
y = torch.empty(A0.rows, dtype=torch.float64, device="cuda")
ref = None
for v, A in handles.items():                                        # every variant agrees with the first one on a second vector (stale hand-offs would show)
    for _ in range(3):
        A.spmv(x, y)
    A.spmv(x2, y)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
        scale = float(ref.abs().max())
    else:
        d = float((y - ref).abs().max())
        print(f"variant {v}: max |y - y_first| = {d:.3e} (scale {scale:.3e})", flush=True)
        if d > 1e-9 * scale:
            print(f"WARNING: variant {v} disagrees with the first one", flush=True)
times = {v: [] for v in handles}
for r in range(args.rounds):
    for v, A in handles.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            A.spmv(x, y)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / args.iters)
info = A0.info()
for v, ts in times.items():
    ts = sorted(ts)
    med, mn = ts[len(ts) // 2], ts[0]
    gbs = info["algorithmic_bytes"] / (med * 1e-3) / 1e9
    print(f"{v:>10s}  median {med:.4f} ms  min {mn:.4f} ms  {info['nnz'] / med / 1e6:8.1f} GEdges/s  {gbs:7.1f} GB/s  frac {gbs / 8000:.4f}", flush=True)
