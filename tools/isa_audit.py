#!/usr/bin/env python3
"""Which inner loops wait for memory once per load? Compiles every .hip of g4s_amd/csrc to gfx950 assembly and lists the inner loops that hold at most three vector
loads and an `s_waitcnt vmcnt(0)` — the shape of a loop the compiler did not unroll into "all loads, one wait" (round 4: the producer's staging of a band of x was
sixteen such round trips per work item). Most hits are harmless (a loop that runs once or twice, a pointer chase that cannot be batched); read them.
Usage: python tools/isa_audit.py [file.hip ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "g4s_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(SRC) if f.endswith(".hip"))
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fvisibility=hidden",
               "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", os.path.join(SRC, f), "-o", tmp.name]
        if subprocess.run(cmd, capture_output=True).returncode:
            print(f, ": does not compile stand-alone"); continue
        lines = open(tmp.name).read().split("\n")
    kernel = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m: kernel = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", l)
        if not m: continue
        lab, loads, waits, j = m.group(1), 0, 0, i + 1
        while j < len(lines) and j < i + 400 and not re.match(r"^_Z", lines[j]):
            t = lines[j]
            loads += bool(re.search(r"global_load|buffer_load", t))
            waits += "s_waitcnt vmcnt(0)" in t
            if re.search(r"s_cbranch\w+ " + re.escape(lab) + r"\b", t): break
            j += 1
        if 0 < loads <= 3 and waits:
            try: name = subprocess.run(["c++filt", kernel], capture_output=True, text=True).stdout.strip()
            except OSError: name = kernel
            print(f"{f}: {name[:100]}  {lab}: {loads} load(s), {waits} × vmcnt(0), {j - i} lines")
