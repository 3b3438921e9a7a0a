#!/usr/bin/env python3
"""The byte table behind VERDICT r3 item 5 (configs[1], blocked SpMV): what each remaining layout idea would take off the 2.00 GB a product moves.
Hot bands as the plan picks them (27 groups of 16 K columns by degree); everything else is a "cold" entry in a natural 16 K band.
  (1) non-empty-column numbering of the natural bands: bands over the non-empty cold columns only — fewer, denser bands: fewer distinct (row, band)
      pairs (= partial sums that cross HBM twice, 18 B each) and fewer bands of x to stage;
  (2) u8 row deltas instead of u16 local rows in the consumer stream: 1 B per micro-run;
  (3) window-aligned hot cells: one slot base per 32-entry window instead of one per 8-entry span in the hot cells.
Usage: python tools/byte_table.py"""
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
bits, H = 14, 27
deg = torch.bincount(cols, minlength=n)
order = torch.argsort(deg, descending=True)
rank = torch.empty_like(order)
rank[order] = torch.arange(n, device="cuda")
hot = rank[cols] < H * (1 << bits)
cold = ~hot
nb = (n + (1 << bits) - 1) >> bits
# the plan today: natural bands for the cold entries
pairs_hot = torch.unique(rows[hot] * H + (rank[cols[hot]] >> bits)).numel()
pairs_cold_nat = torch.unique(rows[cold] * nb + (cols[cold] >> bits)).numel()
# (1): cold columns renumbered over the non-empty ones
is_cold_col = torch.zeros(n, dtype=torch.bool, device="cuda")
is_cold_col[cols[cold]] = True
newid = torch.cumsum(is_cold_col.long(), 0) - 1
ncold_cols = int(is_cold_col.sum())
nb2 = (ncold_cols + (1 << bits) - 1) >> bits
pairs_cold_cmp = torch.unique(rows[cold] * nb2 + (newid[cols[cold]] >> bits)).numel()
n_hot, n_cold = int(hot.sum()), int(cold.sum())
print(f"nnz {nnz}: hot {n_hot} ({n_hot / nnz:.3f}) in {H} bands, cold {n_cold} in {nb} natural bands; non-empty cold columns {ncold_cols} of {n} -> {nb2} bands")
print(f"distinct (row, band) pairs: hot {pairs_hot} ({pairs_hot / n_hot:.3f} per entry), cold natural {pairs_cold_nat} ({pairs_cold_nat / n_cold:.3f}), cold compacted {pairs_cold_cmp} ({pairs_cold_cmp / n_cold:.3f})")
d_pairs = pairs_cold_nat - pairs_cold_cmp
x_nat, x_cmp = nb * (1 << bits) * 8, nb2 * (1 << bits) * 8
b1 = d_pairs * 18 + (x_nat - x_cmp)
micro = 0.349 * nnz
b2 = micro * 1.0
hot_spans = n_hot / 8
b3 = hot_spans * 4 * (1 - 8 / 32)                       # slot index per 8-entry span (4 B) -> per 32-entry window
tot = 1999193093
print(f"(1) compacted cold bands: {d_pairs} fewer partial sums x 18 B = {d_pairs * 18 / 1e6:.1f} MB, x staging {x_nat / 1e6:.1f} -> {x_cmp / 1e6:.1f} MB: total -{b1 / 1e6:.1f} MB = {100 * b1 / tot:.2f} %")
print(f"(2) u8 row deltas in the consumer stream: -{b2 / 1e6:.1f} MB = {100 * b2 / tot:.2f} %")
print(f"(3) window-aligned hot cells (slot index per window): -{b3 / 1e6:.1f} MB = {100 * b3 / tot:.2f} %")
print(f"sum: -{(b1 + b2 + b3) / 1e6:.1f} MB = {100 * (b1 + b2 + b3) / tot:.2f} % of the {tot / 1e9:.2f} GB a product moves (stop rule: build only if >= 8 %)")
