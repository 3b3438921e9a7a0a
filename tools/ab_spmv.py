#!/usr/bin/env python3
"""A/B timing of SpMV variants in ONE process (rule 24 of the HIP guide: interleaved rounds, median and min).
usage: python tools/ab_spmv.py [--workloads rmat,rmat12m,banded,lap7] [--rounds 5] [--iters 50]
rmat12m: a 1.25 M × 1.25 M R-MAT of ≈ 12 M entries — the size of one rank's slab of configs[1] in the 8-way partition (what lifts a slab lifts
every matrix of this size)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="rmat,banded,lap7")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--variants", default="0,4")
args = ap.parse_args()

for w in args.workloads.split(","):
    A0 = host.rmat_csr(1_250_000, 21, 12_500_000, 20240523) if w == "rmat12m" else bench.build_matrix(w, host, False)
    x = host.synth_vector(7, A0.cols)
    variants = {}
    for v in args.variants.split(","):
        variants[v] = host.CSR(A0.rowptr, A0.colids, A0.values, A0.rows, A0.cols, spmv_flags=int(v))
    y = torch.empty(A0.rows, dtype=torch.float64, device="cuda")
    times = {v: [] for v in variants}
    for A in variants.values():
        for _ in range(5):
            A.spmv(x, y)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        for v, A in variants.items():
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
        print(f"{w:7s} flags={v:>4s} median {med:.4f} ms  min {mn:.4f} ms  {info['nnz'] / med / 1e6:8.1f} GEdges/s  "
              f"{gbs:7.1f} GB/s  frac {gbs / 8000:.3f}  blocks={info['stream_blocks']} long={info['long_rows']}/{info['long_chunks']}", flush=True)
    del variants, A0, x, y
    torch.cuda.empty_cache()
