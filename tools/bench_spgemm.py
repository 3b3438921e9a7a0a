#!/usr/bin/env python3
"""SpGEMM benchmark in the reference's protocol (mm/src/mkl_spgemm.cpp:60-85): 1 warm-up + mean of N runs of C = A·A, GFLOPS = 2·flop/t.
BASELINE configs[2]: R-MAT scale 21 (n = 2 097 152); edge factor 3 is the largest whose nnz(C) fits the reference's int32 crpt.
usage: python tools/bench_spgemm.py [--ef 3] [--runs 3] [--scale 21]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ef", type=float, default=3.0)
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--two-phase", action="store_true", help="time g4s_spgemm_symbolic and g4s_spgemm_numeric as two calls instead of the one-call form")
ap.add_argument("--cpu-sample-ef", type=float, default=0.0, help="also time the oracle's hash SpGEMM (1 host thread) on this smaller edge factor")
ap.add_argument("--mkl", type=int, default=0, metavar="THREADS",
                help="also time the REFERENCE's call sequence (mm/inc/mkl_mult.h:40-111 on oneMKL, oracle/mkl_ref.py) on the SAME input with this many threads "
                     "(the reference hard-codes 14, mm/src/mkl_spgemm.cpp:61); 1 warm-up + 2 runs; skipped where the MKL runtime is absent")
args = ap.parse_args()
lib = capi.load()
n = 1 << args.scale
A = host.rmat_csr(n, args.scale, int(args.ef * n), 20240522)
A.values.abs_()
flop = host.get_flop(A, A)
crpt = torch.empty(n + 1, dtype=torch.int32, device="cuda")
cnnz = C.c_int64()


def run():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not args.two_phase:
        # the call the reference times: mkl(A, B, C, timing) / HashSpGEMM(A, B, C) as one unit (mkl_spgemm.cpp:67-81)
        c = host.HashSpGEMM(A, A)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cnnz.value = c.nnz
        del c
        return 0.0, (t1 - t0) * 1e3
    capi.check(lib.g4s_spgemm_symbolic(n, n, n, A.rowptr.data_ptr(), A.colids.data_ptr(), A.rowptr.data_ptr(), A.colids.data_ptr(), crpt.data_ptr(),
                                       C.byref(cnnz), None))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ccol = torch.empty(cnnz.value, dtype=torch.int32, device="cuda")
    cval = torch.empty(cnnz.value, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    capi.check(lib.g4s_spgemm_numeric(n, n, n, A.rowptr.data_ptr(), A.colids.data_ptr(), A.values.data_ptr(), A.rowptr.data_ptr(), A.colids.data_ptr(),
                                      A.values.data_ptr(), crpt.data_ptr(), ccol.data_ptr(), cval.data_ptr(), capi.DEVICE_POINTERS | capi.SORT_OUTPUT, None))
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    del ccol, cval
    return (t1 - t0) * 1e3, (t3 - t2) * 1e3


run()
sym, num = zip(*[run() for _ in range(args.runs)])
s, m = sum(sym) / len(sym), sum(num) / len(num)
cpu = None
if args.cpu_sample_ef > 0:
    from tests import oracle_lib
    o = oracle_lib.load()
    As = host.rmat_csr(n, args.scale, int(args.cpu_sample_ef * n), 20240522)
    As.values.abs_()
    fs = host.get_flop(As, As)
    rp, ci, va = As.to_host()
    t0 = time.perf_counter()
    crp, cci, cva = o.spgemm((rp, ci, va), (rp, ci, va), n)
    dt = time.perf_counter() - t0
    cpu = {"value": round(2 * fs / dt / 1e9, 3), "unit": "GFLOPS", "cores": 1, "kind": "port",
           "sample": f"R-MAT scale {args.scale}, edge factor {args.cpu_sample_ef}: flop {fs}, nnz(C) {len(cci)}, {dt:.1f} s (oracle_spgemm_symbolic + numeric)"}
ref = None
if args.mkl > 0:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mkl_ref
    if mkl_ref.available():
        mkl_ref.load(threading="gnu")
        rp, ci, va = A.to_host()
        tm, tot = {}, []
        for i in range(3):
            crp, cci, cva = mkl_ref.mkl_spgemm((rp, ci, va), (rp, ci, va), n, n, n, timings=tm, threads=args.mkl)
            if i:
                tot.append(dict(tm))
            nnzc = len(cci)
            if i < 2:
                del crp, cci, cva
        mean = {k: sum(t[k] for t in tot) / len(tot) for k in tot[0]}
        # the GPU result against the reference library's, whole arrays: index arrays bit for bit, values relative (all terms positive)
        import numpy as np
        c = host.HashSpGEMM(A, A)
        grp, gci, gva = c.to_host()
        del c
        cmp = {"crpt_equal": bool(np.array_equal(grp, crp)), "ccol_equal": bool(len(gci) == len(cci) and np.array_equal(gci, cci))}
        if cmp["ccol_equal"]:
            cmp["values_max_rel_err"] = float(np.max(np.abs(gva - cva) / cva))
        del grp, gci, gva, crp, cci, cva
        ref = {"value": round(2 * flop / (mean["total"] * 1e-3) / 1e9, 3), "unit": "GFLOPS", "cores": args.mkl, "kind": "reference call sequence on oneMKL " + mkl_ref.version()[35:52].strip(),
               "sample": f"the same matrix (flop {flop}, nnz(C) {nnzc}), mean of 2 runs after 1 warm-up, stage times ms: " + ", ".join(f"{k} {v:.1f}" for k, v in mean.items()),
               "spmm_only_GFLOPS": round(2 * flop / (mean["spmm"] * 1e-3) / 1e9, 3), "nnz_C_equal_to_gpu": nnzc == cnnz.value,
               "gpu_result_against_it": cmp}
    else:
        ref = {"skipped": "libmkl_rt.so not found on this box"}
# SURVEY.md §8d byte model: A once, every B row once per use (12 B per product), C once, the row pointers
model_bytes = 12 * A.nnz + 4 * n + 12 * flop + 12 * cnnz.value + 8 * n
print(json.dumps({"cpu_baseline": cpu, "reference_baseline": ref,
                  "roofline": {"bound": "hbm (row re-reads of B served by L2/MALL) / LDS atomics", "model": "12*nnz(A) + 4*rows + 12*flop + 12*nnz(C) + 8*rows (SURVEY 8d)",
                               "model_bytes": model_bytes, "achieved": round(model_bytes / ((s + m) * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(model_bytes / ((s + m) * 1e-3) / 1e9 / 8000.0, 4)}, "metric": "fp64 SpGEMM A*A GFLOPS (2*flop/t)", "value": round(2 * flop / ((s + m) * 1e-3) / 1e9, 3), "unit": "GFLOPS",
                  "config": {"workload": f"R-MAT scale {args.scale}, edge factor {args.ef}, C = A*A", "rows": n, "nnz_A": A.nnz, "flop": flop, "nnz_C": cnnz.value,
                             "compression": round(flop / max(cnnz.value, 1), 3)},
                  **({"symbolic_ms": round(s, 2), "numeric_ms": round(m, 2)} if args.two_phase else {"call_ms": round(m, 2), "form": "one call (g4s_spgemm_csr_i32_f64, device pointers)"}), "runs": args.runs,
                  "compulsory_bytes": 12 * (2 * A.nnz + cnnz.value), "compulsory_GBps": round(12 * (2 * A.nnz + cnnz.value) / ((s + m) * 1e-3) / 1e9, 1)}))
