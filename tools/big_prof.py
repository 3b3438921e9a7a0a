"""Per-section cycle split of spgemm_numeric_big_kernel (library built with EXTRA=-DG4S_PROFILE_BIG). Usage: python tools/big_prof.py [--ef 3]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from g4s_amd import capi, host

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--ef", type=float, default=3.0)
a = ap.parse_args()
lib = capi.load()
n = 1 << a.scale
A = host.rmat_csr(n, a.scale, int(a.ef * n), 20240522)
host.HashSpGEMM(A, A)
buf = (C.c_ulonglong * 64)()
lib.g4s_debug_big_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.g4s_debug_big_prof(buf, 1)
host.HashSpGEMM(A, A)
lib.g4s_debug_big_prof(buf, 0)
names = ["row start: barrier", "p2 open: columns arrive + LDS writes (later chunks)", "round: lookup + issue", "row start: metadata loads", "round: load wait", "round: accumulate", "p2 load K/zero", "p2 bucket index", "p2 walk", "p2 long list", "p2 store", "(count) rounds of the reporting wavefront", "(count) halving steps in those rounds", "flat map", "flat rounds"]
tot = sum(buf[:11])
names = names[:15] + ["-"]
names[9] = "p2 open: columns arrive + LDS writes (first chunk of a row)"
names[6] = "p2 open: prefetch issue + barrier + span"
for n, v in zip(names, buf):
    print(f"{n:60s} {v:16d} ticks {100.0 * v / max(tot, 1):6.2f} %")
