"""Randomised SpMV cases against scipy in extended precision (tolerance 1e-10·Σ|a_ij·x_j| per row): power-law rows with hubs past the long-row
limit, bands with stray entries, stencils with missing entries, empty rows / columns, alpha / beta, every path (auto, stream, blocked) and both
plan builders. Usage: python tools/stress_spmv.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
from g4s_amd import capi, host  # noqa: E402

rng = np.random.default_rng(args.seed)


def power_law(n, m):
    deg = np.minimum((rng.pareto(1.2, n) * 3).astype(np.int64), m)
    deg[rng.integers(0, n, 3)] = rng.integers(2049, min(m, 30000) + 1, 3) if m > 2049 else m      # hubs: long rows
    deg[rng.integers(0, n, n // 10)] = 0
    rows = np.repeat(np.arange(n), deg)
    cols = (rng.random(len(rows)) ** 3 * m).astype(np.int64)
    return sp.csr_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, m))


def banded(n, hb, strays):
    d = [rng.uniform(-1, 1, n - abs(k)) for k in range(-hb, hb + 1)]
    A = sp.diags(d, list(range(-hb, hb + 1)), shape=(n, n), format="lil")
    for _ in range(strays):
        A[rng.integers(0, n), rng.integers(0, n)] = rng.uniform(-1, 1)
    return A.tocsr()


def stencil(s, holes):
    n = s * s * s
    idx = np.arange(n)
    offs = [0, 1, -1, s, -s, s * s, -s * s]
    rows, cols = [], []
    for o in offs:
        ok = (idx + o >= 0) & (idx + o < n)
        rows.append(idx[ok]); cols.append(idx[ok] + o)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    keep = rng.random(len(rows)) >= holes
    return sp.csr_matrix((rng.uniform(-1, 1, keep.sum()), (rows[keep], cols[keep])), shape=(n, n))


bad = 0
for case in range(args.cases):
    kind = str(rng.choice(["power_law", "banded", "stencil", "rect"]))
    if kind == "power_law":
        n = int(rng.choice([3000, 70000, 300000])); A = power_law(n, n)
    elif kind == "rect":
        A = power_law(int(rng.choice([500, 40000])), int(rng.choice([90000, 400000])))
    elif kind == "banded":
        A = banded(int(rng.choice([5000, 300000])), int(rng.choice([1, 3, 8])), int(rng.choice([0, 0, 1, 5])))
    else:
        A = stencil(int(rng.choice([12, 40, 66])), float(rng.choice([0.0, 0.0, 0.02, 0.3])))
    A.sum_duplicates(); A.sort_indices()
    path = str(rng.choice(["auto", "stream", "blocked"]))
    planner = str(rng.choice(["default", "host", "device"]))
    for v in ("G4S_PLAN_HOST", "G4S_PLAN_DEVICE"):
        os.environ.pop(v, None)
    if planner != "default":
        os.environ["G4S_PLAN_" + planner.upper()] = "1"
    flags = {"auto": 0, "stream": capi.SPMV_STREAM, "blocked": capi.SPMV_BLOCKED}[path]
    rp = torch.from_numpy(A.indptr.astype(np.int32)).cuda(); ci = torch.from_numpy(A.indices.astype(np.int32)).cuda(); va = torch.from_numpy(A.data).cuda()
    G = host.CSR(rp, ci, va, A.shape[0], A.shape[1], spmv_flags=flags)
    x = rng.uniform(-1, 1, A.shape[1]); y0 = rng.uniform(-1, 1, A.shape[0])
    alpha, beta = (1.0, 0.0) if rng.random() < 0.5 else (float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, -0.5])))
    y = torch.from_numpy(y0.copy()).cuda()
    G.spmv(torch.from_numpy(x).cuda(), y, alpha, beta)
    got = y.cpu().numpy()
    Al = A.astype(np.longdouble)
    want = alpha * (Al @ x.astype(np.longdouble)) + beta * y0.astype(np.longdouble)
    scale = abs(alpha) * (abs(A) @ np.abs(x)) + abs(beta) * np.abs(y0)
    ok = bool(np.all(np.abs(got - want.astype(np.float64)) <= 1e-10 * scale + 1e-300))
    print(f"case {case:3d}: {kind:9s} {A.shape[0]}x{A.shape[1]} nnz={A.nnz} path={path} -> {G.info()['spmv_path']} planner={planner} alpha={alpha:.2f} beta={beta:.2f} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    del G
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
