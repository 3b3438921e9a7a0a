#!/usr/bin/env python3
"""End-to-end latency of the two entry points on small and mid-size problems — the scale of the reference's own examples:
can_24 by default (mm/src/mkl_spgemm.cpp:8-9), patents_main 240 547², nnz 560 943 (mm/README.md:9).

One case per process (so that rocprofv3 can count its dispatches and host waits):
    python tools/small_sizes.py --case rmat5e5 --op spmv      [--reps 200]
    python tools/small_sizes.py --case rmat5e5 --op spgemm    [--reps 20]
Cases: rmat5e5 (2^18 rows, 5e5 draws), rmat2e6 (2^19 rows, 2e6 draws), rmat12m (1.25 M rows, 1.25e7 draws: an 8-way slab of configs[1]), lap120 (120^3 7-point stencil),
er5e5 (240 547 rows, 560 943 uniform draws: the shape of patents_main), can24 (24 rows, 160 uniform draws: the shape of can_24).
Prints one JSON line: wall time per call (host clock around a synchronised loop) and, for SpMV, the stream time per call (events around the loop).
tools/small_sizes.sh runs every case, once plain and once under rocprofv3, and writes the table (profiles/r05_small_sizes.txt)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", required=True)
ap.add_argument("--op", required=True, choices=["spmv", "spgemm"])
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--flags", type=int, default=0, help="g4s_csr_create flags (path forcing)")
a = ap.parse_args()


def build(case):
    if case == "rmat5e5":
        return host.rmat_csr(1 << 18, 18, 500_000, 20240524)
    if case == "rmat2e6":
        return host.rmat_csr(1 << 19, 19, 2_000_000, 20240525)
    if case == "rmat12m":
        return host.rmat_csr(1_250_000, 21, 12_500_000, 20240523)
    if case == "lap120":
        return host.laplacian_csr(7, 120, 120, 120)
    if case in ("er5e5", "can24"):                                 # uniform draws in the shape of the reference's own examples: patents_main (mm/README.md:9), can_24 (mkl_spgemm.cpp:8-9)
        import ctypes as C
        n, draws = (240547, 560943) if case == "er5e5" else (24, 160)
        g = torch.Generator(device="cuda"); g.manual_seed(20240526)
        keys = torch.unique(torch.randint(0, n * n, (draws,), dtype=torch.int64, device="cuda", generator=g), sorted=True)
        nnz = keys.numel()
        rowptr = torch.empty(n + 1, dtype=torch.int32, device="cuda"); colids = torch.empty(nnz, dtype=torch.int32, device="cuda"); values = torch.empty(nnz, dtype=torch.float64, device="cuda")
        capi.check(capi.load().g4s_synth_csr_from_keys(7, n, n, C.c_void_p(keys.data_ptr()), nnz, C.c_void_p(rowptr.data_ptr()), C.c_void_p(colids.data_ptr()), C.c_void_p(values.data_ptr()), None))
        torch.cuda.synchronize()
        return host.CSR(rowptr, colids, values, n, n)
    raise SystemExit(f"unknown case {case}")


A = build(a.case)
out = {"case": a.case, "op": a.op, "rows": A.rows, "nnz": A.nnz}
if a.op == "spmv":
    reps = a.reps or 200
    B = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols, spmv_flags=a.flags)
    x = host.synth_vector(7, A.cols)
    y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    B.handle
    torch.cuda.synchronize()
    out["create_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    for _ in range(10):
        B.spmv(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        B.spmv(x, y)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    info = B.info()
    dev = e0.elapsed_time(e1) * 1e-3 / reps
    out.update({"reps": reps, "path": {0: "stream", 1: "blocked", 3: "diagonal", 4: "block-row"}.get(info["spmv_path"], info["spmv_path"]),
                "us_per_call_stream": round(dev * 1e6, 2), "us_per_call_wall": round(wall * 1e6, 2),
                "algorithmic_bytes": info["algorithmic_bytes"], "frac_of_8TBs": round(info["algorithmic_bytes"] / dev / 8e12, 4),
                "GEdges_s": round(A.nnz / dev / 1e9, 2)})
else:
    reps = a.reps or 20
    flop = host.get_flop(A, A)
    c = host.HashSpGEMM(A, A)
    cnnz = c.nnz
    del c
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c = host.HashSpGEMM(A, A)                                  # g4s_spgemm_csr_i32_f64, device pointers: the call the reference times (mkl_spgemm.cpp:67-81)
        del c
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    out.update({"reps": reps, "flop": flop, "nnz_C": cnnz, "ms_per_call": round(wall * 1e3, 4), "GFLOPS": round(2 * flop / wall / 1e9, 2)})
print(json.dumps(out))
