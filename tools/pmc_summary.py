#!/usr/bin/env python3
"""Average each PMC counter per kernel over the dispatches in rocprofv3 counter_collection CSVs under a directory."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            short = None
            for key in ("spmv_csr_adaptive", "spmv_long_fixup", "pb_producer", "pb_consumer", "pb_scale_rows", "pb_gather_hot", "pb_prepare"):
                if key in k:
                    short = key
            if short is None:
                import re
                m = re.search(r"(\w+_kernel)(<[^>]*>)?", k)
                if m and any(key in m.group(1) for key in ("spgemm", "hub_", "row_flop", "elem_", "dense_", "tb_", "spmv_dia")):
                    short = m.group(1) + (m.group(2) or "")
            if short is None:
                continue
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    d = out[k]
    if "FETCH_SIZE" in d:
        # gfx950: FETCH_SIZE (KiB) counts 128-B requests at 64 B on wide coalesced streams → ×2 for streamed reads (MI355X_MICROARCH.md §HBM)
        d["fetch_bytes_raw"] = d["FETCH_SIZE"]["mean"] * 1024
        d["fetch_bytes_x2"] = d["FETCH_SIZE"]["mean"] * 2048
    if "WRITE_SIZE" in d:
        d["write_bytes"] = d["WRITE_SIZE"]["mean"] * 1024
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        h, m = d["TCC_HIT_sum"]["mean"], d["TCC_MISS_sum"]["mean"]
        d["l2_hit_rate"] = h / (h + m) if h + m else None
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
