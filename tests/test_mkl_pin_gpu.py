"""GPU parity against Intel oneMKL's own results (tests/golden/mkl_spgemm.npz, mkl_dense.npz — generated on the build box by
tests/golden/make_golden.py through the reference's call sequence, mm/inc/mkl_mult.h:40-111; the GPU box needs only the fixtures).
Index arrays bit-exact; values within 1e-10 · Σ|terms| (north_star)."""
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
