#!/usr/bin/env python3
"""What would popularity-selected "hot" column bands buy the blocked SpMV? Columns are ranked by degree; the top H·16K go to H hot
bands (in rank order), the rest keep their natural 16 K bands. Prints distinct (row, band) pairs per nonzero = the floor of the
partial sums the producer hands to the consumer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import host
import bench
A = bench.build_matrix("rmat", host, False)
n = A.rows
rows = torch.repeat_interleave(torch.arange(n, device="cuda", dtype=torch.int64), (A.rowptr[1:] - A.rowptr[:-1]).long())
cols = A.colids.long()
nnz = cols.numel()
bits = 14
deg = torch.bincount(cols, minlength=n)
order = torch.argsort(deg, descending=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device="cuda")
nat = cols >> bits
nb = (n + (1 << bits) - 1) >> bits
for H in (0, 1, 2, 4, 8, 16, 32):
    hot = rank[cols] < H * (1 << bits)
    band = torch.where(hot, rank[cols] >> bits, nat + H)
    pair = rows * (nb + H) + band
    d = torch.unique(pair).numel()
    print(f"H={H:3d}: nnz in hot bands {hot.double().mean().item():.3f}, distinct (row, band) pairs / nnz = {d / nnz:.4f}")
