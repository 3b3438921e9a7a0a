"""Host-side gaps of one g4s_csr_create: HIP API calls longer than 0.2 ms between the first and the last plan kernel, from a rocprofv3 --kernel-trace --hip-trace run of
tools/plan_create_trace.py. Usage: python tools/plan_trace_report.py <rocprof output dir>"""
import csv, glob, sys
d = sys.argv[1]
k = sorted(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in k if "pb_sample_kernel" in r["Kernel_Name"]]
ends = [int(r["End_Timestamp"]) for r in k if "pb_fill_consumer" in r["Kernel_Name"]]
for n, (t0, t1) in enumerate(zip(starts, ends)):
    print(f"create {n}: {(t1 - t0) / 1e6:.2f} ms from the first to the last plan kernel")
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 - 2_000_000 and e <= t1 + 3_000_000 and e - s > 200_000:
            print(f"   at {(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e6:7.3f} ms  {r['Function']}")
