"""GPU parity against Intel oneMKL's own results (tests/golden/mkl_spgemm.npz, mkl_dense.npz — generated on the build box by
tests/golden/make_golden.py through the reference's call sequence, mm/inc/mkl_mult.h:40-111; the GPU box needs only the fixtures).
Index arrays bit-exact; values within 1e-10 · Σ|terms| (north_star). The two *_at_full_size tests run the same call sequence live on the
GPU box's host where the oneMKL runtime exists (it ships in this image) and compare BASELINE configs[1] and configs[2] whole."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "mkl_spgemm.npz")
TOL = 1e-10


def _abs(A):
    return (A[0], A[1], np.abs(A[2]))


@pytest.mark.parametrize("name", ["tri4", "rmat8", "rmat10", "rect", "plaw", "spmv_rmat8"])
def test_spgemm_equals_mkl(oracle, name):
    from g4s_amd import host
    g = np.load(GOLD)
    A = tuple(g[f"{name}_a{k}"] for k in ("rpt", "col", "val"))
    B = tuple(g[f"{name}_b{k}"] for k in ("rpt", "col", "val"))
    M, K, N = (int(v) for v in g[f"{name}_mkn"])
    c = host.HashSpGEMM(host.CSR.from_host(*A, M, K), host.CSR.from_host(*B, K, N))
    crpt, ccol, cval = c.to_host()
    assert np.array_equal(crpt, g[f"{name}_crpt"]) and np.array_equal(ccol, g[f"{name}_ccol"])
    _, _, scale = oracle.spgemm(_abs(A), _abs(B), N, sort_output=True)
    assert np.all(np.abs(cval - g[f"{name}_cval"]) <= TOL * scale + 1e-300)


@pytest.mark.parametrize("flags", ["stream", "blocked"])
def test_spmv_equals_mkl_one_column_spgemm(oracle, flags):
    from g4s_amd import capi, host
    g = np.load(GOLD)
    A = tuple(g[f"spmv_rmat8_a{k}"] for k in ("rpt", "col", "val"))
    x = g["spmv_rmat8_bval"]
    M, K, _ = (int(v) for v in g["spmv_rmat8_mkn"])
    y_mkl = np.zeros(M)
    y_mkl[np.nonzero(np.diff(g["spmv_rmat8_crpt"]))[0]] = g["spmv_rmat8_cval"]
    a = host.CSR.from_host(*A, M, K, spmv_flags=capi.SPMV_STREAM if flags == "stream" else capi.SPMV_BLOCKED)
    y = a.spmv(torch.from_numpy(x).cuda()).cpu().numpy()
    _, asum = oracle.spmv_ld(*A, x)
    assert np.all(np.abs(y - y_mkl) <= TOL * asum + 1e-300)


def test_spgemm_config3_equals_mkl_at_full_size():
    """BASELINE config 3 itself (R-MAT scale 21, edge factor 3, C = A·A: nnz(C) 1.94e9) against the reference's call sequence run live on
    oneMKL on this box's host (oracle/mkl_ref.py; skipped where the runtime is absent): crpt and ccol bit for bit, values to 1e-10 relative
    (every term is positive: Σ|terms| is the value itself)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mkl_ref
    if not mkl_ref.available():
        pytest.skip("libmkl_rt.so not on this box")
    from g4s_amd import capi, host
    mkl_ref.load(threading="gnu")
    n = 1 << 21
    A = host.rmat_csr(n, 21, 3 * n, 20240522)
    A.values.abs_()
    rp, ci, va = A.to_host()
    crp, cci, cva = mkl_ref.mkl_spgemm((rp, ci, va), (rp, ci, va), n, n, n, threads=32)
    c = host.HashSpGEMM(A, A)
    grp, gci, gva = c.to_host()
    del c
    assert np.array_equal(grp, crp), "row pointer differs from oneMKL's"
    assert len(gci) == len(cci) == 1942743230 and np.array_equal(gci, cci), "column ids differ from oneMKL's"
    assert float(np.max(np.abs(gva - cva) / cva)) <= TOL
    del grp, gci, gva, crp, cci, cva
    capi.check(capi.load().g4s_trim())


def test_spmv_config1_equals_mkl_at_full_size(oracle):
    """configs[1] itself (R-MAT 10M×10M, nnz 98 736 299): y = A·x from the blocked path the benchmark times against oneMKL run live on this
    box — x as a 10M×1 CSR matrix, A·x as the reference's own mkl_sparse_spmm call sequence (oracle/mkl_ref.py); skipped where the runtime
    is absent. Rows that hold entries come back as C's rows; tolerance 1e-10 · Σ|a_ij·x_j|."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import mkl_ref
    if not mkl_ref.available():
        pytest.skip("libmkl_rt.so not on this box")
    import bench
    from g4s_amd import host
    mkl_ref.load(threading="gnu")
    A = bench.build_matrix("rmat", host, False)
    assert A.nnz == 98_736_299 and A.info()["spmv_path"] == 1
    x = host.synth_vector(7, A.cols)
    y = A.spmv(x).cpu().numpy()
    rp, ci, va = A.to_host()
    xh = x.cpu().numpy()
    n = A.cols
    B = (np.arange(n + 1, dtype=np.int32), np.zeros(n, dtype=np.int32), xh)
    crp, cci, cva = mkl_ref.mkl_spgemm((rp, ci, va), B, A.rows, n, 1, threads=32)
    y_mkl = np.zeros(A.rows)
    has = np.diff(crp) > 0
    assert np.array_equal(has, np.diff(rp) > 0)                   # a row of C exists exactly where A's row holds entries
    y_mkl[has] = cva
    _, asum = oracle.spmv_ld(rp, ci, va, xh)
    assert np.all(np.abs(y - y_mkl) <= TOL * asum + 1e-300)
