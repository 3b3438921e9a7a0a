#!/usr/bin/env python3
"""Where do the products and outputs of a SpGEMM sit? Rows of C by the numeric classes of g4s_amd/csrc/spgemm.hip (kNumLimits; the rank rows are those whose product bound
exceeds kSymLimits' 32 768): rows, products (flop) and outputs (nnz) per class. Usage: python tools/class_split.py [--scale 21 --ef 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--ef", type=float, default=3.0)
a = ap.parse_args()
n = 1 << a.scale
A = host.rmat_csr(n, a.scale, int(a.ef * n), 20240522)
C = host.HashSpGEMM(A, A)
rp = A.rowptr.long()
blen = (rp[1:] - rp[:-1])
per_entry = blen[A.colids.long()]
cs = torch.zeros(A.nnz + 1, dtype=torch.long, device="cuda")
cs[1:] = torch.cumsum(per_entry, 0)
flop = cs[rp[1:]] - cs[rp[:-1]]
crp = C.rowptr.long()
nz = crp[1:] - crp[:-1]
rank = flop > 32768
edges = [0, 32, 512, 1024, 2048, 4096, 8192, 1 << 62]
print(f"rows {n}, products {int(flop.sum())}, outputs {int(nz.sum())}")
print(f"{'class':>14s} {'rows':>9s} {'products':>12s} {'outputs':>12s} {'prod/out':>8s}")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (~rank) & (nz > lo) & (nz <= hi)
    f, z = int(flop[m].sum()), int(nz[m].sum())
    print(f"{'<=' + str(hi) if hi < 1 << 62 else 'more':>14s} {int(m.sum()):9d} {f:12d} {z:12d} {f / max(z, 1):8.2f}")
f, z = int(flop[rank].sum()), int(nz[rank].sum())
print(f"{'rank':>14s} {int(rank.sum()):9d} {f:12d} {z:12d} {f / max(z, 1):8.2f}")
for lo, hi in [(0, 8192), (8192, 32768), (32768, 131072), (131072, 1 << 62)]:
    m = rank & (nz > lo) & (nz <= hi)
    f, z = int(flop[m].sum()), int(nz[m].sum())
    print(f"  rank, {lo} < nz <= {hi if hi < 1 << 62 else 'inf'}: rows {int(m.sum())}, products {f}, outputs {z}, ratio {f / max(z, 1):.2f}")
