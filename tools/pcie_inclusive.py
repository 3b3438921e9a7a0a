#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot host-pointer SpMV call (g4s_spmv_csr_i32_f64: upload A and x, build the plan, run, download y)
on BASELINE configs[1]. Never the bench `value` — recorded in DESIGN.md only."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from g4s_amd import capi, host
import bench
lib = capi.load()
A = bench.build_matrix("rmat", host, False)
rp, ci, va = A.to_host()
x = host.synth_vector(7, A.cols).cpu().numpy()
y = np.zeros(A.rows)
out = {}
for name, flags in (("auto(blocked plan)", 0), ("stream plan", capi.SPMV_STREAM)):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        capi.check(lib.g4s_spmv_csr_i32_f64(A.rows, A.cols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, x.ctypes.data, y.ctypes.data, 1.0, 0.0, flags))
        ts.append(time.perf_counter() - t0)
    out[name] = {"ms": round(min(ts) * 1e3, 1), "GEdges/s": round(A.nnz / min(ts) / 1e9, 3)}
print(json.dumps({"workload": "R-MAT 10M, nnz %d, host arrays in and out" % A.nnz, "one_shot": out}))
