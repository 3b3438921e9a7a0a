#!/usr/bin/env python3
"""The latency table of VERDICT r4 item 4 (profiles/r05_small_sizes.txt): every case of tools/small_sizes.py, timed plain, then run twice under rocprofv3
(kernel + HIP API trace) with N and 2N calls — the differences of the two runs are the dispatches and the host waits of N calls, set-up excluded.
Usage (on the GPU box): python tools/small_sizes_table.py [--out gpurun_out/r05_small_sizes.txt] [--cases can24,er5e5,rmat5e5,rmat2e6,rmat12m,lap120]"""
import argparse, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_small_sizes.txt"))
ap.add_argument("--cases", default="can24,er5e5,rmat5e5,rmat2e6,rmat12m,lap120")
ap.add_argument("--ops", default="spmv,spgemm")
a = ap.parse_args()
WAITS = ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyDtoH", "hipMemcpyHtoD", "hipStreamWaitEvent")
env = dict(os.environ, TMPDIR="/tmp")


def run_plain(case, op, extra=()):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "small_sizes.py"), "--case", case, "--op", op, *extra], capture_output=True, text=True, cwd=ROOT, timeout=600)
    if r.returncode:
        return None
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def counts(case, op, reps):
    d = f"/tmp/ss_{case}_{op}_{reps}"
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--hip-trace", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "tools", "small_sizes.py"),
                        "--case", case, "--op", op, "--reps", str(reps)], capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
    if r.returncode:
        return None
    nk = sum(1 for f in glob.glob(d + "/*/*kernel_trace.csv") for _ in csv.DictReader(open(f)))
    waits = {}
    for f in glob.glob(d + "/*/*hip_api_trace.csv"):
        for row in csv.DictReader(open(f)):
            fn = row.get("Function", "")
            if fn in WAITS:
                waits[fn] = waits.get(fn, 0) + 1
    shutil.rmtree(d, ignore_errors=True)
    return nk, waits


lines = ["# end-to-end latency on small and mid-size problems (tools/small_sizes_table.py): time per call, kernel dispatches and host waits per call",
         "# (dispatches / waits = difference of two profiled runs with N and 2N calls, divided by N)", ""]
for case in a.cases.split(","):
    for op in a.ops.split(","):
        if op == "spgemm" and case == "rmat12m":
            continue                                              # (an 8-way SLAB of configs[1] is an SpMV workload; its square overflows the reference's int32 crpt)
        t = run_plain(case, op)
        if t is None:
            lines.append(f"{case:8s} {op:7s} FAILED")
            continue
        n = 40 if op == "spmv" else 6
        c1, c2 = counts(case, op, n), counts(case, op, 2 * n)
        disp = waits = "?"
        if c1 and c2:
            disp = f"{(c2[0] - c1[0]) / n:.1f}"
            keys = sorted(set(c1[1]) | set(c2[1]))
            w = {k: (c2[1].get(k, 0) - c1[1].get(k, 0)) / n for k in keys}
            waits = ", ".join(f"{k} {v:.1f}" for k, v in w.items() if v > 0) or "none"
        if op == "spmv":
            lines.append(f"{case:8s} spmv    rows {t['rows']:>9d} nnz {t['nnz']:>10d} path {t['path']:9s} {t['us_per_call_stream']:9.2f} us/call (stream) {t['us_per_call_wall']:9.2f} us/call (host loop)  "
                         f"{t['GEdges_s']:8.2f} GEdges/s  frac {t['frac_of_8TBs']:.3f}  create {t['create_ms']:.2f} ms | dispatches/call {disp} | host waits/call: {waits}")
        else:
            lines.append(f"{case:8s} spgemm  rows {t['rows']:>9d} nnz {t['nnz']:>10d} flop {t['flop']:>12d} nnz(C) {t['nnz_C']:>11d} {t['ms_per_call']:9.4f} ms/call  {t['GFLOPS']:8.2f} GFLOPS"
                         f" | dispatches/call {disp} | host waits/call: {waits}")
        print(lines[-1], flush=True)
os.makedirs(os.path.dirname(a.out), exist_ok=True)
open(a.out, "w").write("\n".join(lines) + "\n")
