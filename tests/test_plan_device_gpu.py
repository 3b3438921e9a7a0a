"""g4s_csr_create plans large device-resident matrices on the device (csrc/spmv.hip: build_plan_device); the host builder stays for small
ones and for host arrays. Both must give the same products: bit-identical wherever rows are summed by one lane (stencil, band — including the
short blocks the device builder's forced cuts leave), within 1e-10·Σ|terms| elsewhere; the long rows found must be the same."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _two_plans(monkeypatch, A, host, capi, flags):
    monkeypatch.setenv("G4S_PLAN_HOST", "1")
    H = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols, spmv_flags=flags)
    ih = H.info()
    monkeypatch.delenv("G4S_PLAN_HOST")
    monkeypatch.setenv("G4S_PLAN_DEVICE", "1")
    D = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols, spmv_flags=flags)
    idv = D.info()
    monkeypatch.delenv("G4S_PLAN_DEVICE")
    return H, D, ih, idv


@pytest.mark.parametrize("kind", ["lap7", "banded", "rmat"])
def test_device_plan_equals_host_plan(oracle, monkeypatch, kind):
    from g4s_amd import capi, host
    if kind == "lap7":
        A = host.laplacian_csr(7, 70, 70, 70)
    elif kind == "banded":
        A = host.banded_csr(400_000, 5, 11)
    else:
        n = 1 << 18
        A = host.rmat_csr(n, 18, 24 * n, 9)                        # hubs of > 2048 entries: long rows
    H, D, ih, idv = _two_plans(monkeypatch, A, host, capi, capi.SPMV_STREAM)
    assert ih["spmv_path"] == 0 and idv["spmv_path"] == 0
    assert ih["long_rows"] == idv["long_rows"] and ih["long_chunks"] == idv["long_chunks"]
    assert idv["stream_blocks"] >= ih["stream_blocks"]              # forced cuts every 4096 rows add blocks, never remove any
    x = host.synth_vector(5, A.cols)
    yh, yd = H.spmv(x), D.spmv(x)
    rp, ci, va = A.to_host()
    xh = x.cpu().numpy()
    yo = oracle.spmv_mt_y(rp, ci, va, xh)                          # the oracle's fp64 left-to-right sums
    _, asum = oracle.spmv_ld(rp, ci, va, xh)
    assert np.all(np.abs(yd.cpu().numpy() - yo) <= TOL * asum + 1e-300)
    if kind != "rmat":
        assert torch.equal(yh, yd) and np.array_equal(yd.cpu().numpy(), yo)    # rows of a stencil / band: one lane per row in every block
    else:
        assert ih["long_rows"] > 0
    # the default choice: device plan from 2^18 rows on
    Z = host.CSR(A.rowptr, A.colids, A.values, A.rows, A.cols, spmv_flags=capi.SPMV_STREAM)
    assert Z.info()["stream_blocks"] == (idv["stream_blocks"] if A.rows >= (1 << 18) else ih["stream_blocks"])


def test_device_plan_rejects_a_decreasing_row_pointer(monkeypatch):
    from g4s_amd import capi, host
    monkeypatch.setenv("G4S_PLAN_DEVICE", "1")
    rp = torch.tensor([0, 3, 2, 5], dtype=torch.int32, device="cuda")
    ci = torch.tensor([0, 1, 2, 0, 1], dtype=torch.int32, device="cuda")
    va = torch.ones(5, dtype=torch.float64, device="cuda")
    h = C.c_void_p()
    st = capi.load().g4s_csr_create(C.byref(h), 3, 3, 5, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), capi.DEVICE_POINTERS)
    assert st == capi.ERR_INVALID
