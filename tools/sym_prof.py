"""Per-section cycle split of spgemm_symbolic_window_kernel (library built with EXTRA=-DG4S_PROFILE_BIG): the symbolic phase alone.
Usage: python tools/sym_prof.py [--ef 3]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
host.HashSpGEMM(A, A)                                            # the one-call form: the symbolic window kernels also emit the sorted columns
lib.g4s_debug_big_prof(buf, 0)
names = {16: "zero bitmap + barrier", 17: "flat products (mark)", 19: "emit (all)", 22: "row start / end",
         24: "  emit: words + popcount", 25: "  emit: scan + barrier", 26: "  emit: list build + barrier", 29: "  emit: write columns (wave 0's own loops)", 27: "  emit: barrier behind the column writes", 28: "  emit: last barrier"}
tot = sum(buf[k] for k in (16, 17, 19, 22))
for k, nme in names.items():
    print(f"{nme:26s} {buf[k]:16d} ticks {100.0 * buf[k] / max(tot, 1):6.2f} %")
