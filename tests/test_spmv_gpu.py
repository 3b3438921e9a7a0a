"""GPU parity: the HIP SpMV (through the C-ABI) against the oracle on the same inputs.

Tolerance (fp64): |y_gpu − y_oracle|_i ≤ 1e-10 · Σ_k |a_ik x_k|  (north_star: 1e-10 relative). Rows reduced by one lane
(blocks with > 128 rows) are summed in the oracle's order and must be BIT-identical.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.helpers import power_law_csr, random_csr

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _check(oracle, A, rp, ci, va, x, alpha=1.0, beta=0.0, y0=None, exact=False):
    xd = torch.from_numpy(x).cuda()
    yd = None if y0 is None else torch.from_numpy(y0.copy()).cuda()
    y = A.spmv(xd, yd, alpha, beta).cpu().numpy()
    want = oracle.spmv(rp, ci, va, x, y0, alpha, beta)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    scale = abs(alpha) * asum + (abs(beta) * np.abs(y0) if y0 is not None else 0.0)
    err = np.abs(y - want)
    assert np.all(err <= TOL * scale + 1e-300), f"max rel err {np.max(err / (scale + 1e-300))}"
    if exact:
        assert np.array_equal(y, want)
    return y


@pytest.mark.parametrize("rows,cols,density,seed", [(1, 1, 1.0, 0), (5, 7, 0.5, 1), (300, 300, 0.05, 2), (5000, 4000, 0.004, 3),
                                                    (20000, 20000, 0.0005, 4)])
def test_spmv_random(oracle, rows, cols, density, seed):
    from g4s_amd import host
    rp, ci, va = random_csr(rows, cols, density, seed, empty_rows=[0, rows // 2] if rows > 4 else [])
    A = host.CSR.from_host(rp, ci, va, rows, cols)
    x = np.random.default_rng(seed).uniform(-1, 1, cols)
    _check(oracle, A, rp, ci, va, x)
    _check(oracle, A, rp, ci, va, x, alpha=-2.5, beta=0.75, y0=np.random.default_rng(9).uniform(-1, 1, rows))


def test_spmv_short_rows_bit_exact(oracle):
    # ≥ 129 rows per block → one lane per row, oracle's left-to-right order → bit-identical
    from g4s_amd import host
    rp, ci, va = oracle.laplacian5(200, 150)
    A = host.CSR.from_host(rp, ci, va, 30000, 30000)
    x = np.random.default_rng(0).uniform(-1, 1, 30000)
    _check(oracle, A, rp, ci, va, x, exact=True)
    inf = A.info()
    assert inf["long_rows"] == 0 and inf["stream_blocks"] >= 30000 * 5 // 2048


def test_spmv_heavy_rows_inside_many_row_blocks(oracle):
    # A block of the streaming path with hundreds of short rows AND a row of several hundred entries (what a power-law matrix looks like between its hubs):
    # the short rows are summed by one lane each in the oracle's order — bit-identical — the heavy one by a whole wavefront (within the tolerance, run to run equal).
    from g4s_amd import capi, host
    rng = np.random.default_rng(5)
    rows = cols = 4000
    lens = np.full(rows, 2)
    heavy = [7, 700, 701, 1500, 3999]
    for r, n in zip(heavy, [900, 65, 64, 1999, 300]):
        lens[r] = n
    lens[[0, 100, 2000]] = 0
    rp = np.zeros(rows + 1, dtype=np.int32)
    rp[1:] = np.cumsum(lens)
    ci = np.concatenate([np.sort(rng.choice(cols, n, replace=False)) for n in lens]).astype(np.int32)
    va = rng.uniform(-1, 1, ci.size)
    A = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=capi.SPMV_STREAM)
    assert A.info()["spmv_path"] == 0 and A.info()["long_rows"] == 0
    x = rng.uniform(-1, 1, cols)
    y = _check(oracle, A, rp, ci, va, x)
    want = oracle.spmv(rp, ci, va, x, None, 1.0, 0.0)
    short = lens <= 64
    assert np.array_equal(y[short], want[short])
    _check(oracle, A, rp, ci, va, x, alpha=-0.5, beta=2.0, y0=rng.uniform(-1, 1, rows))
    xd = torch.from_numpy(x).cuda()
    assert torch.equal(A.spmv(xd), A.spmv(xd))


def test_warm_up_runs_every_family_once(g4s):
    # g4s_warm_up: one small matrix through the plan builders, the SpMV paths and its own square; callable again
    from g4s_amd import capi
    capi.check(g4s.g4s_warm_up())
    capi.check(g4s.g4s_warm_up())


def test_spmv_power_law_with_hubs(oracle):
    # empty rows, short rows, medium rows (shuffle path) and hubs > TILE_NNZ / > LONG_CHUNK (chunked path)
    from g4s_amd import host
    rows = cols = 30000
    rp, ci, va = power_law_csr(rows, cols, 17, 25000)
    lens = np.diff(rp)
    assert lens.max() > 8192 and (lens == 0).sum() > 100 and ((lens > 2048) & (lens <= 8192)).sum() >= 0
    A = host.CSR.from_host(rp, ci, va, rows, cols)
    inf = A.info()
    assert inf["long_rows"] == int((lens > inf["tile_nnz"]).sum()) and inf["long_chunks"] >= inf["long_rows"]
    x = np.random.default_rng(1).uniform(-1, 1, cols)
    _check(oracle, A, rp, ci, va, x)
    _check(oracle, A, rp, ci, va, x, alpha=0.5, beta=-1.0, y0=np.ones(rows))
    # NT and plain-load variants agree bit for bit (same arithmetic, different cache policy)
    B = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=4)
    xd = torch.from_numpy(x).cuda()
    assert torch.equal(A.spmv(xd), B.spmv(xd))
    # reproducible run to run (no atomics)
    assert torch.equal(A.spmv(xd), A.spmv(xd))


def test_spmv_edge_shapes(oracle):
    from g4s_amd import host
    # all rows empty
    rp = np.zeros(11, np.int32)
    A = host.CSR.from_host(rp, np.zeros(0, np.int32), np.zeros(0), 10, 10)
    y0 = np.arange(10, dtype=np.float64)
    y = A.spmv(torch.ones(10, dtype=torch.float64, device="cuda"), torch.from_numpy(y0.copy()).cuda(), 1.0, 2.0).cpu().numpy()
    assert np.array_equal(y, 2.0 * y0)
    yn = A.spmv(torch.ones(10, dtype=torch.float64, device="cuda"), torch.full((10,), float("nan"), dtype=torch.float64, device="cuda"))
    assert torch.all(yn == 0)                                   # beta == 0 never reads y
    # one fully dense row among empties; a single row exactly TILE_NNZ and TILE_NNZ+1 long
    for n in (2048, 2049, 8192, 8193):
        rp = np.array([0, 0, n, n], np.int32)
        ci = np.arange(n, dtype=np.int32)
        va = np.random.default_rng(n).uniform(-1, 1, n)
        A = host.CSR.from_host(rp, ci, va, 3, n)
        _check(oracle, A, rp, ci, va, np.random.default_rng(1).uniform(-1, 1, n))
    # exactly TILE_ROWS+1 single-entry rows (row cap boundary)
    n = 1025
    rp = np.arange(n + 1, dtype=np.int32)
    A = host.CSR.from_host(rp, np.arange(n, dtype=np.int32)[::-1].copy(), np.ones(n), n, n)
    x = np.arange(n, dtype=np.float64)
    _check(oracle, A, rp, np.arange(n, dtype=np.int32)[::-1].copy(), np.ones(n), x, exact=True)


def test_spmv_rejects_bad_input(g4s):
    from g4s_amd import capi
    h = C.c_void_p()
    rp = np.array([0, 2, 1], np.int32)           # decreasing
    ci = np.array([0, 1], np.int32)
    va = np.ones(2)
    st = g4s.g4s_csr_create(C.byref(h), 2, 2, 1, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, 0)
    assert st == capi.ERR_INVALID
    rp = np.array([0, 1, 2], np.int32)
    ci = np.array([0, 5], np.int32)              # column out of range: must be refused, not gathered
    st = g4s.g4s_csr_create(C.byref(h), 2, 2, 2, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, 0)
    assert st == capi.ERR_INVALID and b"column" in g4s.g4s_last_error()


def test_spmv_one_shot_host_pointers(oracle, g4s):
    # the mv.c-shaped call: caller-owned host arrays in and out, synchronous
    from g4s_amd import capi
    rp, ci, va = random_csr(400, 300, 0.03, 12)
    x = np.random.default_rng(2).uniform(-1, 1, 300)
    y = np.zeros(400)
    capi.check(g4s.g4s_spmv_csr_i32_f64(400, 300, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, x.ctypes.data, y.ctypes.data, 1.0, 0.0, 0))
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    assert np.all(np.abs(y - want) <= TOL * asum + 1e-300)


def test_spmv_golden_fixture():
    import os
    from g4s_amd import host
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "spgemm_rmat8.npz"))
    n = int(g["n"])
    A = host.CSR.from_host(g["arpt"], g["acol"], g["aval"], n, n)
    y = A.spmv(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert np.allclose(y, g["y"], rtol=0, atol=1e-10 * np.abs(g["y"]).max())


def test_spmv_full_size_properties():
    """BASELINE config 2 scale (10M×10M R-MAT, ≈1e8 nnz): size-independent properties instead of the oracle —
    linearity A(ax+bz) = aAx + bAz, row-sum identity A·1 = Σ_k values (segment sums), reproducibility."""
    from g4s_amd import host
    n = 10_000_000
    A = host.rmat_csr(n, 24, 100_000_000, 20240521)
    assert 0.9e8 < A.nnz <= 1.0e8
    x = host.synth_vector(7, n)
    z = host.synth_vector(8, n)
    yx, yz = A.spmv(x), A.spmv(z)
    ycomb = A.spmv(2.0 * x - 0.5 * z)
    # scale for the tolerance: |A|·(|2x| + |0.5z|)
    Aabs = host.CSR(A.rowptr, A.colids, A.values.abs(), n, n)
    scale = Aabs.spmv(2.0 * x.abs() + 0.5 * z.abs())
    assert torch.all((ycomb - (2.0 * yx - 0.5 * yz)).abs() <= 1e-10 * scale + 1e-300)
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    rowsum = torch.zeros(n, dtype=torch.float64, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (A.rowptr[1:] - A.rowptr[:-1]).long())
    rowsum.index_add_(0, rows, A.values)
    abs_rowsum = Aabs.spmv(ones)
    assert torch.all((A.spmv(ones) - rowsum).abs() <= 1e-10 * abs_rowsum + 1e-300)
    assert torch.all((A.spmv(x) - yx).abs() <= 1e-10 * Aabs.spmv(x.abs()) + 1e-300)   # blocked path: LDS atomics, last bits may differ
    inf = A.info()
    assert inf["algorithmic_bytes"] == 12 * A.nnz + 4 * (n + 1) + 16 * n


@pytest.mark.parametrize("kind", ["random", "powerlaw", "lap5", "tiny", "wide"])
def test_spmv_blocked_path(oracle, kind):
    """The blocked path (G4S_SPMV_BLOCKED, propagation blocking): same parity bar as the streaming path. Matrices larger than one 16K band in both directions, with
    empty rows, hubs, alpha/beta, and rows/cols that are not multiples of the band."""
    from g4s_amd import capi, host
    if kind == "random":
        rows, cols = 40000, 50000
        rp, ci, va = random_csr(rows, cols, 0.0004, 3, empty_rows=[0, 17000, 39999])
    elif kind == "powerlaw":
        rows = cols = 70000
        rp, ci, va = power_law_csr(rows, cols, 29, 30000)
    elif kind == "lap5":
        rows = cols = 200 * 150
        rp, ci, va = oracle.laplacian5(200, 150)
    elif kind == "tiny":
        rows, cols = 3, 5
        rp, ci, va = random_csr(rows, cols, 0.8, 1)
    else:
        rows, cols = 100, 100000
        rp, ci, va = random_csr(rows, cols, 0.01, 5)
    A = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=capi.SPMV_BLOCKED)
    assert A.info()["spmv_path"] == 1
    x = np.random.default_rng(2).uniform(-1, 1, cols)
    _check(oracle, A, rp, ci, va, x)
    _check(oracle, A, rp, ci, va, x, alpha=-1.5, beta=0.25, y0=np.random.default_rng(3).uniform(-1, 1, rows))
    yn = A.spmv(torch.from_numpy(x).cuda(), torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda"))
    assert not torch.isnan(yn).any()                                   # beta == 0 never reads y
    S = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=capi.SPMV_STREAM)
    assert S.info()["spmv_path"] == 0


def test_spmv_path_selection():
    from g4s_amd import host
    # banded 10M: gathers are local → streaming path; R-MAT 10M: no locality → blocked path
    assert host.banded_csr(6_000_000, 5, 1).info()["spmv_path"] == 3      # diagonal-structured → index-free path
    assert host.banded_csr(6_000_000, 5, 1, spmv_flags=16).info()["spmv_path"] == 0   # G4S_SPMV_STREAM forces the CSR kernel
    A = host.rmat_csr(6_000_000, 23, 30_000_000, 5)
    assert A.info()["spmv_path"] == 1


@pytest.mark.parametrize("flags", [16, 8])
def test_spmv_nonfinite_inputs_propagate_like_the_oracle(oracle, flags):
    """Inf / NaN in x reach exactly the rows the oracle says they reach (pad slots and tail lanes must not leak 0·Inf = NaN)."""
    from g4s_amd import host
    rows, cols = 40000, 40000
    rp, ci, va = power_law_csr(rows, cols, 41, 5000)
    x = np.random.default_rng(4).uniform(-1, 1, cols)
    x[[5, 16383, 16384, 39999]] = [np.inf, -np.inf, np.nan, np.inf]
    A = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=flags)
    y = A.spmv(torch.from_numpy(x).cuda()).cpu().numpy()
    want = oracle.spmv(rp, ci, va, x)
    assert np.array_equal(np.isnan(y), np.isnan(want))
    assert np.array_equal(np.isposinf(y), np.isposinf(want)) and np.array_equal(np.isneginf(y), np.isneginf(want))
    fin = np.isfinite(want)
    xf = np.where(np.isfinite(x), x, 0.0)
    _, asum = oracle.spmv_ld(rp, ci, va, xf)
    assert np.all(np.abs(y[fin] - want[fin]) <= TOL * asum[fin] + 1e-300)


def test_row_slabs_reproduce_the_full_product():
    """The multi-GPU bench gives every rank a row slab (all columns) chosen by the equal-work rule and its own SpMV plan. On one
    GPU: the slabs of a 6 M-vertex R-MAT (each takes the blocked path on its own) stacked equal the product of the whole matrix."""
    from g4s_amd import dist as gdist, host
    n = 6_000_000
    A = host.rmat_csr(n, 23, 30_000_000, 11)
    x = host.synth_vector(3, n)
    y_full = A.spmv(x)
    Aabs = host.CSR(A.rowptr, A.colids, A.values.abs(), n, n)
    scale = Aabs.spmv(x.abs())
    for parts in (2, 3):
        offs = gdist.row_partition(A.rowptr, parts)
        assert offs[0] == 0 and offs[-1] == n and all(b > a for a, b in zip(offs, offs[1:]))
        work = [int(A.rowptr[b].item()) - int(A.rowptr[a].item()) + (b - a) for a, b in zip(offs, offs[1:])]   # nnz + rows (dist.row_partition)
        assert max(work) - min(work) <= 0.02 * (A.nnz + n)                          # equal work, up to one hub row
        ys = []
        for a, b in zip(offs, offs[1:]):
            rp, ci, va = gdist.slice_rows(A.rowptr, A.colids, A.values, a, b)
            S = host.CSR(rp, ci, va, b - a, n)
            assert S.info()["spmv_path"] in (1, 2)
            ys.append(S.spmv(x))
        y = torch.cat(ys)
        assert torch.all((y - y_full).abs() <= 1e-10 * scale + 1e-300)


@pytest.mark.parametrize("hot", ["0", "1", "3"])
def test_spmv_blocked_hot_column_bands(oracle, monkeypatch, hot):
    """The blocked path ranks columns by degree and gives the most popular ones their own bands (G4S_PB_HOT_BANDS forces how many;
    the default decides from the degree distribution). Same parity bar with none, one and three hot bands, on a matrix whose hot
    columns are scattered over the natural bands, with empty rows, a hub row and a rectangular shape."""
    from g4s_amd import capi, host
    monkeypatch.setenv("G4S_PB_HOT_BANDS", hot)
    rows, cols = 90000, 120000
    rng = np.random.default_rng(41)
    popular = rng.choice(cols, 3000, replace=False)
    lens = rng.integers(0, 12, rows)
    lens[5] = 40000                                                   # hub row
    lens[[0, 1000, rows - 1]] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.empty(rp[-1], np.int32)
    for r in range(rows):
        n = lens[r]
        if n == 0:
            continue
        k = min(n, cols)
        if n > 64:
            c = rng.choice(cols, k, replace=False)
        else:
            c = np.unique(np.where(rng.random(k) < 0.6, popular[rng.integers(0, len(popular), k)], rng.integers(0, cols, k)))
            while len(c) < k:                                         # top up after duplicates were merged
                c = np.unique(np.concatenate([c, rng.integers(0, cols, k - len(c))]))
        ci[rp[r]:rp[r + 1]] = np.sort(c)
    va = rng.uniform(-1, 1, rp[-1])
    A = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=capi.SPMV_BLOCKED)
    assert A.info()["spmv_path"] == 1
    x = rng.uniform(-1, 1, cols)
    _check(oracle, A, rp, ci, va, x)
    _check(oracle, A, rp, ci, va, x, alpha=0.75, beta=-2.0, y0=rng.uniform(-1, 1, rows))


def test_spmv_diagonal_path(oracle):
    """The index-free path for diagonal-structured matrices: chosen for stencils and bands, refused for anything else, bit-identical to the
    oracle on every row (one lane per row, products added in ascending column order), alpha/beta, rectangular shapes, boundary rows."""
    from g4s_amd import capi, host
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    cases = []
    for (nx, ny, nz) in [(50, 40, 30), (300, 7, 5)]:
        rp, ci, va = oracle.laplacian7(nx, ny, nz)
        cases.append((rp, ci, rng.uniform(-1, 1, len(ci)), nx * ny * nz, nx * ny * nz, 7))
    rp, ci, va = oracle.banded(40000, 9, 4)
    cases.append((rp, ci, va, 40000, 40000, 19))
    rp, ci, va = oracle.laplacian7(21, 13, 9)                          # an ODD number of rows: the two-rows-per-lane kernel + the one-row kernel for the last row
    cases.append((rp, ci, rng.uniform(-1, 1, len(ci)), 21 * 13 * 9, 21 * 13 * 9, 7))
    rp, ci, va = oracle.banded(30001, 2, 9)                            # odd, and the last row's entries reach back (offset −2, −1)
    cases.append((rp, ci, va, 30001, 30001, 5))
    # planes of >= 16 K rows, >= 8 of them (far diagonals), an odd plane and an even one
    for (nx, ny, nz) in [(131, 127, 9), (128, 130, 8)]:
        rp, ci, va = oracle.laplacian7(nx, ny, nz)
        cases.append((rp, ci, rng.uniform(-1, 1, len(ci)), nx * ny * nz, nx * ny * nz, 7))
    # rectangular: rows × (rows + 50), offsets {0, 3, 50}, the last rows lose entries
    rows, cols = 30000, 30020
    M = sp.diags([rng.uniform(-1, 1, rows), rng.uniform(-1, 1, rows), rng.uniform(-1, 1, rows)], [0, 3, 50], shape=(rows, cols), format="csr")
    M.sort_indices()
    cases.append((M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64), rows, cols, 3))
    for rp, ci, va, rows, cols, nd in cases:
        A = host.CSR.from_host(rp, ci, va, rows, cols)
        assert A.info()["spmv_path"] == 3
        # y that is only 8-byte aligned takes the one-row kernel for every row: same bits
        xm = torch.from_numpy(rng.uniform(-1, 1, cols)).cuda()
        buf = torch.zeros(rows + 1, dtype=torch.float64, device="cuda")
        assert buf.data_ptr() % 16 == 0
        assert torch.equal(A.spmv(xm, buf[1:]), A.spmv(xm))
        x = rng.uniform(-1, 1, cols)
        _check(oracle, A, rp, ci, va, x, exact=True)
        _check(oracle, A, rp, ci, va, x, alpha=-0.5, beta=2.0, y0=rng.uniform(-1, 1, rows), exact=True)
    # not diagonal-structured: a random matrix, and a band with one stray entry → the CSR kernel
    rp, ci, va = random_csr(20000, 20000, 0.0005, 3)
    assert host.CSR.from_host(rp, ci, va, 20000, 20000).info()["spmv_path"] == 0
    rp, ci, va = oracle.banded(40000, 2, 4)
    B = sp.csr_matrix((va, ci, rp), shape=(40000, 40000)).tolil()
    for stray_row, want_path in ((5000, 0), (20000, 3)):
        # a stray entry in a row the candidate scan does not sample (first / middle / last 2048 rows) is found by the fill pass → CSR kernel;
        # in a sampled row its offset simply becomes one more (nearly empty) diagonal
        Bs = B.copy()
        Bs[stray_row, 7] = 1.5
        Bs = Bs.tocsr()
        Bs.sort_indices()
        arrs = (Bs.indptr.astype(np.int32), Bs.indices.astype(np.int32), Bs.data)
        S = host.CSR.from_host(*arrs, 40000, 40000)
        assert S.info()["spmv_path"] == want_path
        _check(oracle, S, *arrs, rng.uniform(-1, 1, 40000), exact=(want_path == 3))


def test_spmv_randomised_structures():
    """tools/stress_spmv.py, 24 cases: power-law rows with hubs past the long-row limit, bands with stray entries, stencils with holes, rectangular
    shapes, alpha / beta, every path and both plan builders — against scipy in extended precision (1e-10·Σ|terms| per row)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_spmv.py"), "--cases", "24", "--seed", "7"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_spmv_blocked_fresh_vectors_every_launch(oracle):
    """The blocked path keeps state between launches (the gathered hot x values, the partial-sum buffer, the pre-scaled split row bands): a launch
    that read any of it from the previous product would go unnoticed while x stays the same, so every launch gets a NEW x and is compared with
    the oracle; alpha / beta alternate so that the split row bands' pre-scale is exercised too."""
    from g4s_amd import capi, host
    n = 1_500_000
    G = host.rmat_csr(n, 21, 24_000_000, 77)
    A = host.CSR(G.rowptr, G.colids, G.values, n, n, spmv_flags=capi.SPMV_BLOCKED)
    assert A.info()["spmv_path"] == 1
    rp, ci, va = A.to_host()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    y0 = host.synth_vector(99, n)
    for it in range(4):
        x = host.synth_vector(100 + it, n)
        xh = x.cpu().numpy()
        alpha, beta = (1.0, 0.0) if it % 2 == 0 else (-0.5, 1.25)
        y.copy_(y0)
        A.spmv(x, y, alpha, beta)
        want = oracle.spmv_mt_y(rp, ci, va, xh)
        _, asum = oracle.spmv_ld(rp, ci, va, xh)
        want = alpha * want + (beta * y0.cpu().numpy() if beta != 0.0 else 0.0)
        scale = abs(alpha) * asum + abs(beta) * np.abs(y0.cpu().numpy())
        err = np.abs(y.cpu().numpy() - want)
        assert np.all(err <= TOL * scale + 1e-300), f"launch {it}: max rel err {np.max(err / (scale + 1e-300))}"


def test_spmv_block_row_path_for_assembled_fe_matrices(oracle):
    """An assembled finite-element stiffness matrix handed over as plain CSR (BASELINE configs[4]: 3 unknowns per node, so the matrix is made of
    aligned 3×3 blocks) takes the block-row path (spmv_path 4: one block-column id per block, one lane per row) and every row is BIT-identical to
    the oracle — the CSR kernel reduces 81-entry rows with several lanes. Matrices that only look similar (a perturbed column, rows of unequal
    length, unsorted block columns) must stay on the CSR kernel; 2×2 and 4×4 blocks are recognised too."""
    from g4s_amd import capi, host
    from tests.helpers import assemble_csr, hex_mesh, spd_blocks
    ien, idmap, nno, neq = hex_mesh(12, 10, 6)
    K = spd_blocks(len(ien), 24, 3)
    rp, ci, va = assemble_csr(ien, idmap, K, neq)
    A = host.CSR.from_host(rp, ci, va, neq, neq)
    assert A.info()["spmv_path"] == 4 and A.info()["plan_bytes"] > 8 * len(ci)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, neq)
    _check(oracle, A, rp, ci, va, x, exact=True)
    _check(oracle, A, rp, ci, va, x, alpha=-0.75, beta=1.5, y0=rng.uniform(-1, 1, neq), exact=True)
    S = host.CSR.from_host(rp, ci, va, neq, neq, spmv_flags=capi.SPMV_STREAM)
    assert S.info()["spmv_path"] == 0
    _check(oracle, S, rp, ci, va, x)
    # near misses stay on the CSR kernel
    ci2 = ci.copy()
    k = rp[300]                                                       # swap two columns inside one row: the b rows of a node no longer agree
    ci2[k], ci2[k + 1] = ci2[k + 1], ci2[k]
    assert host.CSR.from_host(rp, ci2, va, neq, neq).info()["spmv_path"] == 0
    keep = np.ones(len(ci), bool)
    keep[rp[7]] = False                                               # one entry less in one row
    rp3 = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), rp[:-1]))]).astype(np.int32)
    assert host.CSR.from_host(rp3, ci[keep], va[keep], neq, neq).info()["spmv_path"] == 0
    # generic block sizes: a random block-sparse pattern expanded to b×b dense blocks
    for b in (2, 4):
        nb = 3000
        brp, bci, _ = random_csr(nb, nb, 0.002, 11 + b)
        lens = np.diff(brp)
        rows = nb * b
        rp_b = np.concatenate([[0], np.cumsum(np.repeat(lens * b, b))]).astype(np.int32)
        ci_b = np.concatenate([np.tile((bci[brp[n]:brp[n + 1], None] * b + np.arange(b)[None, :]).ravel(), b) for n in range(nb)]).astype(np.int32)
        va_b = rng.uniform(-1, 1, len(ci_b))
        B = host.CSR.from_host(rp_b, ci_b, va_b, rows, rows)
        if len(ci_b) >= 2 * b * rows:
            assert B.info()["spmv_path"] == 4
        _check(oracle, B, rp_b, ci_b, va_b, rng.uniform(-1, 1, rows), exact=B.info()["spmv_path"] == 4)
