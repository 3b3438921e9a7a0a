#!/usr/bin/env python3
"""Is the diagonal kernel slower on the 431^3 operator (5.3 TB/s of its own bytes) than on the 10 M banded matrix (6.1) because of the operand SIZE or because of
the far planes of x? Same kernel, 7 diagonals each: 431^3 stencil (offsets ±1, ±431, ±431²), an 80 M-row band (offsets −3 … 3), a 10 M-row band (−3 … 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from g4s_amd import host

def run(name, A):
    x = host.synth_vector(7, A.cols)
    y = torch.empty(A.rows, dtype=torch.float64, device="cuda")
    for _ in range(5):
        A.spmv(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        A.spmv(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    own = 8 * A.rows * 7 + 4 * A.rows + 8 * A.rows + 8 * A.cols     # diagonals + masks + y + x once
    print(f"{name:28s} rows {A.rows:9d} path {A.info()['spmv_path']} {ms:7.4f} ms  own bytes {own / 1e9:5.2f} GB -> {own / ms / 1e9:6.2f} TB/s")

run("7-point 431^3", host.laplacian_csr(7, 431, 431, 431))
torch.cuda.empty_cache()
run("band hb 3, 80 M rows", host.banded_csr(80_062_991, 3, 20240521))
torch.cuda.empty_cache()
run("band hb 3, 10 M rows", host.banded_csr(10_000_000, 3, 20240521))
