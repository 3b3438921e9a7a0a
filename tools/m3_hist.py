"""Where the outputs and products of C = A·A sit by output-row size (chunks of 8192 of the big-row kernel). Usage: python tools/m3_hist.py [--ef 3]"""
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
flop = torch.zeros(n, dtype=torch.int64, device=deg.device).index_add_(0, rowid, deg[A.colids.long()])
nz, flop, deg = nz.cpu().numpy(), flop.cpu().numpy(), deg.cpu().numpy()
edges = [0, 32, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 1 << 30]
print(f"{'nz range':>22s} {'rows':>8s} {'outputs':>12s} {'products':>12s} {'mean na':>8s} {'chunk passes':>12s}")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (nz > lo) & (nz <= hi)
    if not m.any(): continue
    chunks = ((nz[m] + 8191) // 8192).sum()
    print(f"{f'({lo}, {hi}]':>22s} {m.sum():8d} {nz[m].sum():12d} {flop[m].sum():12d} {deg[m].mean():8.1f} {chunks:12d}")
