#!/usr/bin/env python3
"""The diagonal kernel with two rows per lane (default) against one row per lane (G4S_SPMV_DIA_ONE_ROW=1, read per launch) IN ONE PROCESS on one handle and one
pair of vectors — for the 431^3 operator and the 10 M banded matrix. (Across processes the same kernel varies by ±8 % on the large operator.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ["G4S_SPMV_LIVE_ENV"] = "1"                             # the library then reads the A/B switches at every launch
from g4s_amd import host

for name, A in (("lap7 431^3", host.laplacian_csr(7, 431, 431, 431)), ("banded 10M hb5", host.banded_csr(10_000_000, 5, 20240521))):
    x = host.synth_vector(7, A.cols)
    y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    A.spmv(x, y)
    ref, res = None, {}
    for rnd in range(4):
        for mode in ("two", "one"):
            if mode == "one":
                os.environ["G4S_SPMV_DIA_ONE_ROW"] = "1"
            else:
                os.environ.pop("G4S_SPMV_DIA_ONE_ROW", None)
            for _ in range(3):
                A.spmv(x, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                A.spmv(x, y)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(round(e0.elapsed_time(e1) / 30, 4))
            if ref is None:
                ref = y.clone()
            assert torch.equal(ref, y)
    os.environ.pop("G4S_SPMV_DIA_ONE_ROW", None)
    print(name, {"two_rows_per_lane": res["two"], "one_row_per_lane": res["one"]})
    del A, x, y
    torch.cuda.empty_cache()
