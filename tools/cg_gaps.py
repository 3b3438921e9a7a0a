#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/bench_cg.py: what one CG iteration is made of — kernel durations and the gaps between
dependent launches (is the loop host-bound, and what would a hipGraph remove?). usage: python tools/cg_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
cg = [r for r in rows if any(k in r["Kernel_Name"] for k in ("cg_direction", "cg_pAp", "cg_update", "elem_matvec", "spmv_", "node_blocks"))]
dur, gaps = defaultdict(list), []
for a, b in zip(cg, cg[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 50:                       # inside a batch of enqueued iterations (a host read-back ends a batch)
        gaps.append(g)
for r in cg:
    name = next(k for k in ("cg_direction", "cg_pAp", "cg_update", "elem_matvec", "spmv_", "node_blocks") if k in r["Kernel_Name"])
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
it = sum(sum(v) / len(v) for v in dur.values())
print("mean kernel durations (us):", {k: round(sum(v) / len(v), 2) for k, v in dur.items()}, "-> sum per iteration", round(it, 2))
gaps.sort()
print(f"gaps between consecutive CG kernels inside a batch (us): n {len(gaps)}, median {gaps[len(gaps) // 2]:.2f}, mean {sum(gaps) / len(gaps):.2f}, p90 {gaps[int(0.9 * len(gaps))]:.2f}")
print(f"one iteration = 4 kernels + 4 gaps = {it + 4 * gaps[len(gaps) // 2]:.1f} us (median gap); a graph replay keeps the dependent-kernel boundary, it only removes host enqueue time")
