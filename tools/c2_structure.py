#!/usr/bin/env python3
"""Structure of BASELINE configs[1] (R-MAT 10M, 1e8 draws, seed 20240521) that decides what a blocked SpMV can do — run on the CPU with the oracle's
generator (no GPU): empty rows / columns, how the nonzeros concentrate in the popular columns, how dense (y tile × x tile) cells are once
the empty rows are squeezed out, how many cold entries merge, and how uneven 8 192-row tiles are. Output: profiles/r02_c2_structure.txt.
usage: python tools/c2_structure.py [--cache DIR]   (caches the 1e8 keys as .npy, ~800 MB)"""
import argparse
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cache", default="/tmp/c2")
args = ap.parse_args()
os.makedirs(args.cache, exist_ok=True)
n = 10_000_000
rf, cf = os.path.join(args.cache, "rows.npy"), os.path.join(args.cache, "cols.npy")
if os.path.exists(rf) and os.path.exists(cf):
    rows, cols = np.load(rf), np.load(cf)
else:
    keys = np.unique(oracle_lib.load().rmat_keys(20240521, 24, n, 0, 100_000_000))
    rows, cols = (keys // n).astype(np.int32), (keys % n).astype(np.int32)
    np.save(rf, rows), np.save(cf, cols)
nnz = rows.size
print(f"configs[1]: n {n}, nnz {nnz}")
rdeg, cdeg = np.bincount(rows, minlength=n), np.bincount(cols, minlength=n)
print(f"empty rows {int((rdeg == 0).sum())} ({(rdeg == 0).mean():.3f}), empty columns {int((cdeg == 0).sum())} ({(cdeg == 0).mean():.3f}); "
      f"nnz in rows of more than 256 entries {rdeg[rdeg > 256].sum() / nnz:.3f}, of at most 16 entries {rdeg[rdeg <= 16].sum() / nnz:.3f}; longest row {rdeg.max()}")
order = np.argsort(-cdeg, kind="stable")
rank = np.empty(n, np.int32)
rank[order] = np.arange(n, dtype=np.int32)
cs = np.cumsum(cdeg[order])
print("share of the nonzeros in the k most popular columns: " + ", ".join(f"{k >> 10}K {cs[k - 1] / nnz:.3f}" for k in (16384, 65536, 131072, 262144, 442368, 524288, 1048576, 2097152)))
ne = rdeg > 0
cr = (np.cumsum(ne) - 1)[rows].astype(np.int64)
rk = rank[cols].astype(np.int64)
NR = int(ne.sum())
print(f"non-empty rows {NR}; cells of 8192 compact rows x 4096 ranked columns:")
for Hc in (131072, 262144, 524288):
    hot = rk < Hc
    cnt = np.bincount((cr[hot] // 8192) * (Hc // 4096) + rk[hot] // 4096, minlength=((NR + 8191) // 8192) * (Hc // 4096))
    cold = ~hot
    pairs = np.unique(rows[cold].astype(np.int64) * 1024 + (cols[cold] >> 14)).size
    print(f"  {Hc >> 10}K hot columns hold {hot.mean():.3f} of the nonzeros in {cnt.size} cells: mean {cnt.mean():.0f}, median {np.median(cnt):.0f} entries; "
          f"share in cells under 256 / 1024 entries {cnt[cnt < 256].sum() / cnt.sum():.3f} / {cnt[cnt < 1024].sum() / cnt.sum():.3f}; "
          f"the {int(cold.sum())} cold entries form {pairs} (row, 16K-column band) pairs = {pairs / cold.sum():.3f} partial sums per entry")
# 64-row chunks, tiles of at most 8192 non-empty rows
cnt64, wk64 = np.add.reduceat(ne.astype(np.int64), np.arange(0, n, 64)), np.add.reduceat(rdeg.astype(np.int64), np.arange(0, n, 64))
tiles, comp, work = [], 0, 0
quota = (nnz + 255) // 256
for w in range(cnt64.size):
    if comp and (comp + cnt64[w] > 8192 or work >= quota):
        tiles.append(work)
        comp = work = 0
    comp += cnt64[w]
    work += wk64[w]
tiles.append(work)
t = np.array(tiles)


def lpt(items, P=256):
    h = [0] * P
    for w in sorted(items, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + w)
    return max(h)


split = []
for w in t:
    k = max(1, int(np.ceil(w / 300000)))
    split += [w / k] * k
print(f"y tiles of <= 8192 non-empty rows: {t.size} tiles, mean {t.mean():.0f} entries, max {t.max()} (the 64 rows that hold the largest hubs); "
      f"greedy schedule on 256 CUs: {lpt(t) / (t.sum() / 256):.2f} x ideal whole, {lpt(split) / (sum(split) / 256):.2f} x ideal with tiles over 300 K entries split")
