"""Per-section cycle split of spgemm_numeric_rank_kernel (library built with -DG4S_PROFILE_BIG: tools/build_variant.sh prof spgemm.hip -DG4S_PROFILE_BIG). Usage: G4S_LIB=… python tools/rank_prof.py [--ef 3]"""
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
host.HashSpGEMM(A, A)
lib.g4s_debug_big_prof(buf, 0)
if os.environ.get("G4S_RANK_PROF_SIZES"):                           # library built with -DG4S_PROFILE_BIG -DG4S_PROFILE_SIZES: chunks and their clock by output count
    sz = buf[48:56]
    tot_t = sum(sz[1::2]) or 1
    for b, lab in enumerate(["<= 512 outputs", "<= 2048", "<= 6144", "more"]):
        n, tk = sz[2 * b], sz[2 * b + 1]
        print(f"chunks of {lab:15s}: {n:8d} ({100.0 * n / max(sum(sz[0::2]), 1):5.1f} %)  clock share {100.0 * tk / tot_t:5.1f} %  ticks per chunk {tk / max(n, 1):9.0f}")
    sys.exit(0)
names = ["mark (round 0 from registers + further rounds)", "barrier 1 (marks done)", "ranks: read words, scan, barrier 2", "ranks: write words, barrier 3", "accumulate (+ lgkmcnt drain)",
         "next chunk's round requested", "barrier 4 (sums done)", "store (+ clean)", "barrier 5"]
buf = buf[32:]
tot = sum(buf[:9]) + sum(buf[12:16])
names += ["(count) chunks", "(count) units", "(count) outputs", "  mark: wait for the chunk's first-round columns", "  mark: loop top (next item / descriptors requested)", "  mark: round 0's marks",
          "  accumulate: round 0 (its records' wait included)"]
for nm, v in zip(names, buf):
    print(f"{nm:55s} {v:16d} ticks {100.0 * v / max(tot, 1):6.2f} %")
ch = max(buf[9], 1)
print(f"chunks seen by the reporting wavefronts: {buf[9]}, units per chunk {buf[10] / ch:.1f}, outputs per chunk {buf[11] / ch:.1f}, ticks per chunk {tot / ch:.0f} (100 MHz: {tot / ch / 100:.2f} us)")
