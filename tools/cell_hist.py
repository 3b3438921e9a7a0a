#!/usr/bin/env python3
"""Distribution of nonzeros over (row band, column band) cells of the R-MAT benchmark matrix, for several band sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import host
import bench
A = bench.build_matrix("rmat", host, False)
n = A.rows
rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), (A.rowptr[1:] - A.rowptr[:-1]).long())
cols = A.colids.long()
for bits in (13, 14, 15, 16):
    nb = (n + (1 << bits) - 1) >> bits
    cell = (rows >> bits) * nb + (cols >> bits)
    cnt = torch.bincount(cell, minlength=nb * nb)
    tot = cnt.sum().item()
    line = f"band 2^{bits}: {nb}x{nb} cells, nonempty {(cnt > 0).sum().item()}, max {cnt.max().item()}"
    for T in (1024, 4096, 16384, 65536, 262144):
        line += f" | >= {T}: {cnt[cnt >= T].sum().item() / tot:.3f} ({(cnt >= T).sum().item()} cells)"
    print(line)
    # distinct (row, colband) pairs = products after run compaction
    pair = rows * nb + (cols >> bits)
    print(f"   distinct (row, column-band) pairs / nnz = {torch.unique(pair).numel() / tot:.3f}")
# column degree concentration
deg = torch.bincount(cols, minlength=n)
sd, _ = torch.sort(deg, descending=True)
cs = torch.cumsum(sd, 0).double() / sd.sum().item()
for H in (4096, 8192, 16384, 65536, 262144, 524288):
    print(f"top {H} columns by degree hold {cs[H - 1].item():.3f} of the nonzeros")
