"""Randomised SpGEMM cases against scipy (index arrays bit for bit, values to 1e-10·Σ|terms|): shapes chosen to hit the flat kernels' seams —
A rows of exactly T / T+1 / 2T entries (entry tiles), B rows of 63 / 64 / 65 / 128 entries (unit boundaries), empty B rows, column counts just
above the column map's threshold, every workgroup shape, both call forms. Usage: python tools/stress_spgemm.py [--cases 40] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--wide", action="store_true", help="B with 5 M … 40 M columns: past the window kernels' mid-row limit (key tables), more than 16 window pieces (no splits)")
args = ap.parse_args()
from g4s_amd import host  # noqa: E402

rng = np.random.default_rng(args.seed)
SHAPE_VARS = ("SYM_MED", "SYM_LARGE", "SYM_WINDOW", "NUM_MED", "NUM_LARGE", "NUM_M2", "NUM_M3")


def random_case():
    K = int(rng.choice([300, 1500, 5000]))
    N = int(rng.choice([70_000, 131_073, 400_000, 1_100_000])) if not args.wide else int(rng.choice([5_000_000, 17_000_000, 40_000_000]))
    used = N if rng.random() < 0.4 else int(N * rng.uniform(0.3, 0.8)) if not args.wide else int(N * rng.choice([0.05, 0.5, 1.0]))          # share of B's columns that hold entries (column map on / off)
    cols_pool = np.sort(rng.choice(N, used, replace=False))
    blens = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, 300, 1000, 4000], K, p=[.05, .1, .1, .1, .1, .1, .1, .1, .15, .07, .03])
    blens = np.minimum(blens, used)
    # skew: rows prefer the low end of the pool (a crowded head, like a power-law graph)
    brows = []
    for l in blens:
        if l == 0:
            brows.append(np.empty(0, np.int64)); continue
        w = rng.random()
        idx = np.unique((rng.random(int(l * 1.3) + 4) ** (1 + 3 * w) * used).astype(np.int64))[:l]
        brows.append(cols_pool[idx])
    brp = np.concatenate([[0], np.cumsum([len(r) for r in brows])]).astype(np.int32)
    bci = np.concatenate(brows).astype(np.int32) if brp[-1] else np.empty(0, np.int32)
    bva = rng.uniform(0.5, 1.0, brp[-1])
    alens = list(rng.choice([0, 1, 2, 5, 40, 255, 256, 257, 1023, 1024, 1025, 2048], 24))
    alens = [min(int(a), K) for a in alens]
    arows = [np.sort(rng.choice(K, a, replace=False)) for a in alens]
    arp = np.concatenate([[0], np.cumsum(alens)]).astype(np.int32)
    aci = np.concatenate(arows).astype(np.int32) if arp[-1] else np.empty(0, np.int32)
    ava = rng.uniform(0.5, 1.0, arp[-1])
    return (arp, aci, ava), (brp, bci, bva), len(alens), K, N


bad = 0
for case in range(args.cases):
    A, B, M, K, N = random_case()
    shape = str(rng.choice(["default", "256", "512", "1024"]))
    two_phase = bool(rng.random() < 0.3)
    for v in SHAPE_VARS:
        if shape == "default":
            os.environ.pop("G4S_SPGEMM_T_" + v, None)
        else:
            os.environ["G4S_SPGEMM_T_" + v] = shape
    if rng.random() < 0.3:
        os.environ["G4S_SPGEMM_STATIC_ROWS"] = "1"
    else:
        os.environ.pop("G4S_SPGEMM_STATIC_ROWS", None)
    a = host.CSR.from_host(*A, M, K)
    b = host.CSR.from_host(*B, K, N)
    c = host.HashSpGEMM(a, b, two_phase=two_phase)
    crpt, ccol, cval = c.to_host()
    S = (sp.csr_matrix((A[2], A[1], A[0]), shape=(M, K)) @ sp.csr_matrix((B[2], B[1], B[0]), shape=(K, N))).tocsr()
    S.sort_indices()
    ok = np.array_equal(S.indptr, crpt) and np.array_equal(S.indices, ccol) and bool(np.all(np.abs(S.data - cval) <= 1e-10 * S.data + 1e-300))
    flop = int(np.diff(B[0])[A[1]].sum()) if len(A[1]) else 0
    print(f"case {case:3d}: K={K} N={N} nnzB={B[0][-1]} flop={flop} nnzC={len(ccol)} shape={shape} two_phase={two_phase} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
