#!/usr/bin/env python3
"""The 431^3 operator through the diagonal kernel with three XCD walks IN ONE PROCESS, on one handle and one pair of vectors (G4S_SPMV_DIA_WALK is read per launch):
s = plane-sliced with staggered starting planes, l = plane-sliced in lock-step, c = contiguous eighths. Process-to-process the same walk varies by ±8 % (page
placement), so A/Bs across processes say little."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ["G4S_SPMV_LIVE_ENV"] = "1"                             # the library then reads the A/B switches at every launch
from g4s_amd import host

n = 431
A = host.laplacian_csr(7, n, n, n)
x = host.synth_vector(7, A.cols)
y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
A.spmv(x, y)
ref = None
res = {}
for rnd in range(4):
    for w in ("s", "l", "c"):
        os.environ["G4S_SPMV_DIA_WALK"] = w
        for _ in range(3):
            A.spmv(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            A.spmv(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        res.setdefault(w, []).append(round(ms, 4))
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y), "the walks must give the same bits"
print({"staggered": res["s"], "lockstep": res["l"], "contiguous": res["c"]})
