#!/usr/bin/env python3
"""Micro-runs per nonzero of the blocked SpMV plan as a function of the span length (entries summed by one lane group before a
partial sum is written), with H hot column bands. A micro-run = same row, same (row band, column band) cell, same span."""
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
bits, H = 14, int(sys.argv[1]) if len(sys.argv) > 1 else 27
deg = torch.bincount(cols, minlength=n)
order = torch.argsort(deg, descending=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device="cuda")
hot = rank[cols] < H * (1 << bits)
band = torch.where(hot, rank[cols] >> bits, (cols >> bits) + H)
nb = ((n + (1 << bits) - 1) >> bits)
cell = band * nb + (rows >> bits)
key = cell * n + rows                         # (cell, row): CSR order is row-major so a stable sort by cell keeps rows ascending
sk, _ = torch.sort(key)
scell = sk // n
first = torch.ones_like(scell, dtype=torch.bool)
first[1:] = scell[1:] != scell[:-1]
idx = torch.arange(nnz, device="cuda")
start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
pos = idx - start
ncells = int(first.sum().item())
for S in (8, 16, 32, 64):
    trip = sk * 64 + (pos // S) % 64 + 0      # (cell,row) with span id folded in (span ids of one (cell,row) run differ by < 64 for these S on C2's cells? no: use unique on pairs)
    span = pos // S
    new = torch.ones(nnz, dtype=torch.bool, device="cuda")
    new[1:] = (sk[1:] != sk[:-1]) | (span[1:] != span[:-1])
    runs = int(new.sum().item())
    pads = ncells * (S - 1) // 2
    print(f"span {S:2d}: micro-runs / nnz = {runs / nnz:.4f}; pad entries ~ {pads / nnz:.4f} of nnz; bytes/nnz ~ {10 * (1 + pads / nnz) + 5 / S * 8 / 8 + 18 * runs / nnz:.2f}")
print(f"distinct (cell,row) pairs / nnz = {torch.unique(sk).numel() / nnz:.4f}, non-empty cells {ncells}")
