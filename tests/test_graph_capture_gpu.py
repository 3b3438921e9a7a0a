"""The per-product entry points are capture-safe: no allocation, no synchronisation, no host read inside g4s_spmv / g4s_elem_op_apply —
a solver may record its launch-bound inner loop in a hipGraph (here through torch.cuda.CUDAGraph, which captures the current stream)
and replay it. Replays must reproduce the eager result (bit for bit on the reproducible paths, 1e-10 on the blocked one)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", ["stream", "blocked", "diagonal"])
def test_spmv_sequence_in_a_hipgraph(path):
    from g4s_amd import capi, host
    if path == "diagonal":
        A = host.laplacian_csr(7, 40, 40, 40)
        assert A.info()["spmv_path"] == 3
    else:
        n = 1 << 16
        R = host.rmat_csr(n, 16, 12 * n, 5)
        A = host.CSR(R.rowptr, R.colids, R.values, n, n, spmv_flags=capi.SPMV_STREAM if path == "stream" else capi.SPMV_BLOCKED)
        assert A.info()["spmv_path"] == (0 if path == "stream" else 1)
    x = host.synth_vector(3, A.cols)
    y = torch.zeros(A.rows, dtype=torch.float64, device="cuda")
    z = torch.zeros_like(y)
    A.spmv(x, y)                                                   # plan, workspaces: everything allocated before the capture
    eager = y.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(5):                                     # y = A·x five times, then z = A·y: a dependent chain of products
                A.spmv(x, y)
            A.spmv(y, z)
    torch.cuda.current_stream().wait_stream(side)
    y.zero_()
    z.zero_()
    g.replay()
    torch.cuda.synchronize()
    if path == "blocked":
        scale = eager.abs().max().item()
        assert float((y - eager).abs().max()) <= 1e-10 * scale
    else:
        assert torch.equal(y, eager)
    ref_z = torch.zeros_like(z)
    A.spmv(eager, ref_z)
    torch.cuda.synchronize()
    assert float((z - ref_z).abs().max()) <= 1e-10 * max(ref_z.abs().max().item(), 1e-300)
    g.replay()                                                     # and again: the workspaces are reusable across replays
    torch.cuda.synchronize()
    assert float((z - ref_z).abs().max()) <= 1e-10 * max(ref_z.abs().max().item(), 1e-300)
