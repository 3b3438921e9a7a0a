#!/usr/bin/env python3
"""Cost of g4s_csr_update_values next to one product and one plan build, on BASELINE configs[1] (blocked path, created with G4S_SPMV_UPDATABLE), the 431^3
stencil slab (diagonal path) and the banded matrix. Usage: python tools/update_values_time.py [--small]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import capi, host

ap = argparse.ArgumentParser()
ap.add_argument("--small", action="store_true")
a = ap.parse_args()


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
n = 1_000_000 if a.small else 10_000_000
cases = [("R-MAT configs[1]", lambda f: host.rmat_csr(n, 24, 10 * n, 20240521, spmv_flags=f | capi.SPMV_UPDATABLE)),
         ("7-pt stencil 431x431x60", lambda f: host.laplacian_csr(7, 431, 431, 60 if not a.small else 8, spmv_flags=f)),
         ("banded 10M hb 5", lambda f: host.banded_csr(n, 5, 3, spmv_flags=f))]
for name, make in cases:
    A = make(0)
    t0 = time.perf_counter()
    A.handle
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t0) * 1e3
    x = host.synth_vector(7, A.cols)
    y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    A.spmv(x, y)
    prod = timed(lambda: A.spmv(x, y), 20)
    v2 = A.values * 1.25
    upd = timed(lambda: A.update_values(v2), 10)
    y2 = A.spmv(x).clone()
    ok = bool(torch.allclose(y2, 1.25 * y, rtol=1e-12, atol=1e-12))
    out.append({"matrix": name, "path": A.info()["spmv_path"], "nnz": A.nnz, "plan_ms": round(plan_ms, 2), "product_ms": round(prod, 4), "update_values_ms": round(upd, 4),
                "update_in_products": round(upd / prod, 2), "plan_in_products": round(plan_ms / prod, 1), "plan_bytes": A.info()["plan_bytes"], "scaled_result_ok": ok})
    print(json.dumps(out[-1]), flush=True)
    A.close()
    del A, x, y, v2, y2
