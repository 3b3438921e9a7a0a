"""Structure of C = A·A by row class: rows, A-entries, products, outputs by product-bound bucket (the symbolic classes) and by output bucket
(the numeric classes), plus where the products sit by B-row length. Usage: python tools/row_hist.py [--ef 3] [--scale 21]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from g4s_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--ef", type=float, default=3.0)
a = ap.parse_args()
n = 1 << a.scale
A = host.rmat_csr(n, a.scale, int(a.ef * n), 20240522)
Cm = host.HashSpGEMM(A, A)
nz = torch.diff(Cm.rowptr.to(torch.int64))
deg = torch.diff(A.rowptr.to(torch.int64))
rowid = torch.repeat_interleave(torch.arange(n, device=deg.device), deg)
blen = deg[A.colids.long()]
flop = torch.zeros(n, dtype=torch.int64, device=deg.device).index_add_(0, rowid, blen)
maxb = torch.zeros(n, dtype=torch.int64, device=deg.device).scatter_reduce_(0, rowid, blen, "amax")
nz, flop, deg, maxb = nz.cpu().numpy(), flop.cpu().numpy(), deg.cpu().numpy(), maxb.cpu().numpy()
edges = [0, 32, 128, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 1 << 20, 1 << 40]
for name, key in (("products (flop bound)", flop), ("outputs (nz)", nz)):
    print(f"by {name}")
    print(f"{'range':>22s} {'rows':>8s} {'A entries':>10s} {'products':>12s} {'outputs':>12s} {'mean na':>8s} {'max na':>7s} {'mean longest B':>14s} {'compr':>6s}")
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (key > lo) & (key <= hi)
        if not m.any(): continue
        print(f"{f'({lo}, {hi}]':>22s} {m.sum():8d} {deg[m].sum():10d} {flop[m].sum():12d} {nz[m].sum():12d} {deg[m].mean():8.1f} {deg[m].max():7d} {maxb[m].mean():14.1f} {flop[m].sum() / max(nz[m].sum(), 1):6.2f}")
print(f"total rows {n} nnz(A) {deg.sum()} products {flop.sum()} outputs {nz.sum()}")
bl = blen.cpu().numpy()
print("products by B-row length")
for lo, hi in zip([0, 16, 64, 256, 1024, 4096, 16384], [16, 64, 256, 1024, 4096, 16384, 1 << 30]):
    m = (bl > lo) & (bl <= hi)
    print(f"  B rows of ({lo}, {hi}] entries: {m.sum():9d} uses, {bl[m].sum():12d} products ({100.0 * bl[m].sum() / bl.sum():5.1f} %)")
