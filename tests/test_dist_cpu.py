"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the row partition + vector exchange, local SpMV by the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import power_law_csr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, kind, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from g4s_amd import dist as gdist
    from tests import oracle_lib
    o = oracle_lib.load()
    if kind == "powerlaw":
        rp, ci, va = power_law_csr(3000, 3000, 5, 900)
    else:
        rp, ci, va = o.laplacian7(9, 8, 7)
    n = len(rp) - 1
    rpt, cit, vat = torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va)
    offs = gdist.row_partition(rpt, world)
    r0, r1 = offs[rank], offs[rank + 1]
    lrp, lci, lva = gdist.slice_rows(rpt, cit, vat, r0, r1)
    x = torch.from_numpy(o.vector(7, n))
    ex = gdist.VectorExchange(offs, rank, world, colids=lci, mode=mode)
    x_full = torch.full((n,), float("nan"), dtype=torch.float64)
    ex(x[r0:r1].clone(), x_full)
    # every referenced column must have arrived
    assert not torch.isnan(x_full[lci.long()]).any()
    y_local = o.spmv(lrp.numpy(), lci.numpy(), lva.numpy(), np.nan_to_num(x_full.numpy()))
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y_local)
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([r0, r1, ex.recv_bytes]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,kind", [(2, "allgatherv", "powerlaw"), (2, "needed", "lap7"), (3, "needed", "powerlaw"),
                                             (3, "allgatherv", "lap7"), (3, "allgather", "powerlaw")])
def test_partitioned_spmv_matches_single(tmp_path, oracle, world, mode, kind):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, mode, kind, str(tmp_path)), nprocs=world, join=True)
    if kind == "powerlaw":
        rp, ci, va = power_law_csr(3000, 3000, 5, 900)
    else:
        rp, ci, va = oracle.laplacian7(9, 8, 7)
    n = len(rp) - 1
    want = oracle.spmv(rp, ci, va, oracle.vector(7, n))
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    assert np.array_equal(got, want)                       # same per-row arithmetic → bit-identical
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    assert metas[0][0] == 0 and metas[-1][1] == n and all(metas[i][1] == metas[i + 1][0] for i in range(world - 1))
    if kind == "lap7" and mode == "needed":
        # halo-only: a slab of the 7-point stencil needs at most one 9×8 plane from each neighbour
        assert all(m[2] <= 2 * 72 * 8 for m in metas)


def test_row_partition_equals_the_reference_rule(oracle):
    """a5: dist.row_partition is BIN::set_rows_offset (mm/inc/BIN.h:101-122) with work = nnz + 1 per row — offsets equal, integer for
    integer, to the oracle's restatement on the same work vector, for regular, power-law, empty-row and more-parts-than-rows inputs."""
    from g4s_amd import dist as gdist
    from tests.helpers import power_law_csr, random_csr
    cases = [random_csr(1000, 800, 0.01, 0, empty_rows=[0, 1, 2, 500, 999]), power_law_csr(5000, 5000, 3, 3000), oracle.laplacian5(40, 30),
             random_csr(5, 5, 0.5, 1), (np.zeros(8, np.int32), np.zeros(0, np.int32), np.zeros(0))]
    for rp, _, _ in cases:
        rows = len(rp) - 1
        work = (np.diff(rp).astype(np.int64) + 1)
        for parts in (1, 2, 3, 4, 7, 8, 14, 64):
            want = oracle.rows_offset(work, parts)
            got = gdist.row_partition(torch.from_numpy(np.ascontiguousarray(rp)), parts)
            # BIN.h leaves the offsets as lower_bound gives them (non-decreasing by construction); row_partition also clamps to rows
            assert got == [min(int(v), rows) for v in want], (rows, parts)


def test_row_partition_balances_work():
    from g4s_amd import dist as gdist
    rp, ci, va = power_law_csr(5000, 5000, 3, 2000)
    offs = gdist.row_partition(torch.from_numpy(rp), 4)
    work = np.diff(rp) + 1
    shares = [work[offs[i]:offs[i + 1]].sum() for i in range(4)]
    assert offs[0] == 0 and offs[-1] == 5000
    assert max(shares) <= sum(shares) / 4 + work.max()     # no part exceeds the average by more than one row's work


def _compact_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from g4s_amd import dist as gdist
    from tests import oracle_lib
    o = oracle_lib.load()
    rp, ci, va = power_law_csr(4000, 4000, 9, 1200)
    n = len(rp) - 1
    rpt, cit, vat = torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va)
    offs = gdist.row_partition(rpt, world)
    r0, r1 = offs[rank], offs[rank + 1]
    lrp, lci, lva = gdist.slice_rows(rpt, cit, vat, r0, r1)
    ex = gdist.CompactExchange(offs, rank, world, lci)
    assert ex.n_ref == len(np.unique(lci.numpy())) and int(ex.local_colids.max()) < ex.n_ref
    x = torch.from_numpy(o.vector(7, n))
    xc = torch.full((ex.n_ref,), float("nan"), dtype=torch.float64)
    for _ in range(2):                                             # twice: the send buffers are reused
        ex(x[r0:r1].clone(), xc)
    assert not torch.isnan(xc).any()
    y_local = o.spmv(lrp.numpy(), ex.local_colids.numpy(), lva.numpy(), xc.numpy())
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y_local)
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([r0, r1, ex.recv_bytes, ex.n_ref]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_compact_exchange_matches_single(tmp_path, oracle, world):
    """Columns renumbered per rank, only the referenced entries of x travel: same per-row arithmetic → bit-identical product."""
    mp.spawn(_compact_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rp, ci, va = power_law_csr(4000, 4000, 9, 1200)
    want = oracle.spmv(rp, ci, va, oracle.vector(7, 4000))
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    assert np.array_equal(got, want)
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    assert all(m[3] <= 4000 for m in metas)
    if world > 1:
        assert all(m[2] < 8 * 4000 for m in metas)                 # fewer bytes than the whole-slab all-gather
