"""How crowded are the 128-column groups the symbolic emit step writes? Outputs of C = A·A by the number of outputs that share their (row, 128-column group),
over the rows of the window classes (more than 512 products). Usage: python tools/group_density.py [--ef 3] [--scale 21]"""
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
Cm = host.HashSpGEMM(A, A)
rp = Cm.rowptr.to(torch.int64)
nz = torch.diff(rp)
deg = torch.diff(A.rowptr.to(torch.int64))
rowid_a = torch.repeat_interleave(torch.arange(n, device=deg.device), deg)
flop = torch.zeros(n, dtype=torch.int64, device=deg.device).index_add_(0, rowid_a, deg[A.colids.long()])
big = flop > 512
rows = torch.repeat_interleave(torch.arange(n, device=nz.device), nz)
keep = big[rows]
key = (rows[keep] << 14) | (Cm.colids[keep].long() >> 7)      # 2^21 / 128 = 2^14 groups per row
del rows, keep
_, cnt = torch.unique_consecutive(key, return_counts=True)
total = int(cnt.sum())
print(f"outputs in rows of more than 512 products: {total}, non-empty groups {cnt.numel()}, mean {total / cnt.numel():.2f} per group")
edges = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128]
lo = 1
for e in edges:
    m = (cnt >= lo) & (cnt <= e)
    print(f"  groups with {lo:3d}..{e:3d} outputs: {int(m.sum()):11d} groups ({100.0 * int(m.sum()) / cnt.numel():5.1f} %), {int(cnt[m].sum()):11d} outputs ({100.0 * int(cnt[m].sum()) / total:5.1f} %)")
    lo = e + 1
