"""Print the headline fields of a bench.py JSON line. Usage: python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "max_rel_err", d.get("max_rel_err"))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", {k: v for k, v in cb.items() if k not in ("reference_library", "sample")})
print("reference_library", json.dumps(cb.get("reference_library")))
if "also" in d:
    print("also", {k: d["also"].get(k) for k in ("value", "ms_per_step", "frac_of_n_gpus_x_8TBs", "spmv_path", "error")})
