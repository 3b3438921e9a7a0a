"""The multi-GPU SpMV of the C-ABI (g4s_spmv_dist_*, csrc/dist.hip): own-column / remote-column split, packed exchange of the referenced x
entries, own-column product overlapped with the exchange.

A one-GPU test box cannot run RCCL between ranks, so two things are rehearsed separately:
  * the partition logic, the want/give wiring, begin / buffers / finish — with 1, 2 and 3 ranks that share the GPU and carry the packed
    buffers through gloo (the "any other transport" half of the API);
  * the RCCL half (g4s_comm_create, g4s_spmv_dist_connect_rccl, ncclSend/ncclRecv inside g4s_spmv_dist_apply on a side stream with the
    event hand-over) — with ONE rank in loopback mode: half of its own slab is treated as remote and travels rank 0 → rank 0.
Every y must equal the single-rank oracle within 1e-10 · Σ|terms|."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import power_law_csr, init_gloo

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _matrix(kind, oracle):
    if kind == "powerlaw":
        n = 60000
        return (*power_law_csr(n, n, 31, 20000), n)
    if kind == "lap7":
        rp, ci, va = oracle.laplacian7(40, 30, 25)
        return rp, ci, va, 40 * 30 * 25
    n = 5000
    rp, ci, va = oracle.banded(n, 3, 5)
    return rp, ci, va, n


def _worker(rank, world, port, kind, out_dir):
    import torch.distributed as dist
    if kind.endswith("+merged"):
        os.environ["G4S_DIST_MERGE"] = "1"                          # force the one-product form (own columns inside the compact x)
        kind = kind[:-7]
    exchange = "packed"
    if kind.endswith("+allgather"):
        exchange, kind = "allgather", kind[:-10]
    init_gloo(rank, world, port)                                   # (a rendezvous FILE, tests/helpers.py)
    torch.cuda.set_device(0)
    from g4s_amd import dist as gdist
    from tests import oracle_lib
    rp, ci, va, n = _matrix(kind, oracle_lib.load())
    rpt = torch.from_numpy(rp).cuda()
    offs = gdist.row_partition(rpt, world)
    r0, r1 = offs[rank], offs[rank + 1]
    lrp, lci, lva = gdist.slice_rows(rpt, torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda(), r0, r1)
    D = gdist.DistSpMV(offs, rank, world, lrp, lci, lva, n, exchange=exchange)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    xl = torch.from_numpy(x[r0:r1]).cuda()
    y1 = D(xl).clone()
    y2 = D(xl)                                                      # a second product on the same handle
    info = D.info()
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y1.cpu().numpy())
    np.save(os.path.join(out_dir, f"z{rank}.npy"), y2.cpu().numpy())
    np.save(os.path.join(out_dir, f"i{rank}.npy"), np.array([r0, r1, info["n_ref"], info["nnz_own"], info["nnz_rem"], info["recv_bytes"], info["send_bytes"], info["reserved"]]))
    dist.barrier()
    D.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(1, "powerlaw"), (2, "powerlaw"), (3, "powerlaw"), (2, "lap7"), (3, "banded"), (3, "powerlaw+merged"), (2, "lap7+merged"),
                                        (2, "powerlaw+allgather"), (3, "lap7+allgather"), (3, "powerlaw+allgather+merged"), (1, "powerlaw+allgather"),
                                        (5, "powerlaw+merged"), (5, "lap7")])   # five ranks + this process: the box's limit of six GPU processes
def test_dist_spmv_capi_matches_oracle(tmp_path, oracle, world, kind):
    mp.spawn(_worker, args=(world, os.path.join(str(tmp_path), "rendezvous"), kind, str(tmp_path)), nprocs=world, join=True)
    merged = kind.endswith("+merged")
    kind = kind[:-7] if merged else kind
    allgather = kind.endswith("+allgather")
    kind = kind[:-10] if allgather else kind
    rp, ci, va, n = _matrix(kind, oracle)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    again = np.concatenate([np.load(tmp_path / f"z{r}.npy") for r in range(world)])
    assert np.all(np.abs(got - want) <= TOL * asum + 1e-300) and np.all(np.abs(again - want) <= TOL * asum + 1e-300)
    metas = [np.load(tmp_path / f"i{r}.npy") for r in range(world)]
    assert sum(int(m[3] + m[4]) for m in metas) == len(ci)          # every nonzero is in exactly one of the two parts
    assert sum(int(m[5]) for m in metas) == sum(int(m[6]) for m in metas)   # what is received was sent
    assert all(int(m[7]) == (1 if merged else 0) + (2 if allgather else 0) for m in metas)
    if merged:
        assert all(int(m[3]) == 0 for m in metas)                   # everything is in the one compact product
    if world == 1:
        assert int(metas[0][5]) == 0 and (allgather or int(metas[0][2]) == 0)
    if allgather:
        pad = max(int(m[1] - m[0]) for m in metas)
        assert all(int(m[2]) == pad * world and int(m[5]) == 8 * pad * (world - 1) for m in metas)
    if kind == "lap7" and world == 2 and not merged:
        # the halo of a slab cut along z is one plane of 40·30 columns per neighbour
        assert [int(m[2]) for m in metas] == [1200, 1200]


def test_dist_spmv_rccl_loopback_single_rank(oracle):
    """RCCL itself, on one GPU: the library makes its own communicator of one rank; with G4S_DIST_LOOPBACK the upper half of the slab is
    'remote', so every product packs, ncclSend/ncclRecv's (to itself) on the side stream, and finishes with the remote-column part."""
    from g4s_amd import dist as gdist
    n = 50000
    rp, ci, va = power_law_csr(n, n, 41, 9000)
    x = np.random.default_rng(8).uniform(-1, 1, n)
    D = gdist.DistSpMV([0, n], 0, 1, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda(), n, loopback=True)
    info = D.info()
    assert info["connected"] == 1 and info["n_ref"] > 0 and info["recv_bytes"] == info["send_bytes"] == 8 * info["n_ref"]
    assert info["nnz_own"] + info["nnz_rem"] == len(ci) and info["nnz_rem"] > 0
    xl = torch.from_numpy(x).cuda()
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    for _ in range(3):
        y = D(xl).cpu().numpy()
        assert np.all(np.abs(y - want) <= TOL * asum + 1e-300)
    # the remote half travels as SEVEN segments by default (seven ncclSend / ncclRecv pairs per product in one group, as one rank of an 8-GPU node has);
    # one segment and sixteen give the same product
    for peers in ("1", "16"):
        os.environ["G4S_DIST_LOOPBACK_PEERS"] = peers
        try:
            E = gdist.DistSpMV([0, n], 0, 1, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda(), n, loopback=True)
        finally:
            del os.environ["G4S_DIST_LOOPBACK_PEERS"]
        assert E.info()["n_ref"] == info["n_ref"] and E.info()["connected"] == 1
        y = E(xl).cpu().numpy()
        assert np.all(np.abs(y - want) <= TOL * asum + 1e-300)
        E.close()
    # the all-gather exchange through RCCL itself (one rank: the collective is a copy, the event hand-over and the side stream are real)
    G = gdist.DistSpMV([0, n], 0, 1, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda(), n, loopback=True, exchange="allgather")
    gi = G.info()
    assert gi["reserved"] == 2 and gi["nnz_rem"] > 0 and gi["n_ref"] == n
    for _ in range(2):
        y = G(xl).cpu().numpy()
        assert np.all(np.abs(y - want) <= TOL * asum + 1e-300)
    G.close()
    # the all-reduce the Krylov dots use
    t = torch.arange(8, dtype=torch.float64, device="cuda")
    D.allreduce_sum(t)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64))
    D.close()


def test_dist_rect_rank_without_rows():
    """A rectangular operator whose rank owns entries of x but no rows (the coarse side of a restriction, a tail rank of the divergence operator):
    begin / finish accept y = NULL and do nothing."""
    import ctypes as C
    from g4s_amd import capi
    lib = capi.load()
    n = 64
    row_off = (C.c_int64 * 2)(0, 0)
    col_off = (C.c_int64 * 2)(0, n)
    rp = (C.c_int32 * 1)(0)
    h = C.c_void_p()
    capi.check(lib.g4s_spmv_dist_create_rect(C.byref(h), 0, 1, row_off, col_off, C.cast(rp, C.c_void_p), None, None, 0))
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_spmv_dist_begin(h, C.c_void_p(x.data_ptr()), None, None))
    capi.check(lib.g4s_spmv_dist_finish(h, None, None))
    capi.check(lib.g4s_spmv_dist_apply(h, C.c_void_p(x.data_ptr()), None, None))
    torch.cuda.synchronize()
    capi.check(lib.g4s_spmv_dist_destroy(h))


_STALL_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch
from g4s_amd import dist as gdist, capi
from tests.helpers import power_law_csr
n = 20000
rp, ci, va = power_law_csr(n, n, 41, 3000)
free0 = torch.cuda.mem_get_info()[0]
os.environ["G4S_DIST_TEST_STALL"] = "4"          # the side stream is held up for 4 s in front of the set-up exchange: a peer that enters late or never
os.environ["G4S_DIST_TIMEOUT_S"] = "0.5"
t0 = time.time()
try:
    gdist.DistSpMV([0, n], 0, 1, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda(), n, loopback=True)
    print("NO ERROR"); sys.exit(3)
except capi.G4SError as e:
    dt = time.time() - t0
    print("ERROR:", e)
    print("SECONDS", round(dt, 2))
# the constructor has destroyed the poisoned handle on its way out: that must not have waited for the stuck exchange either
print("DONE", round(time.time() - t0, 2))
sys.stdout.flush()
os._exit(0)                                        # (the aborted communicator's process is not expected to live on: exit without the runtime's teardown)
"""


def test_dist_connect_time_out_poisons_the_handle_and_returns(tmp_path):
    """the deadline of the set-up waits used to return while the grouped ncclSend / ncclRecv was still pending — the error path then blocked in hipFree
    and hipDeviceSynchronize behind the same stuck operation. Now the communicator is aborted, the handle poisoned, nothing of it is freed or synchronised: with
    a 0.5 s deadline and an exchange that is held up for 4 s (loopback rehearsal), the constructor raises after the deadline, not after the hold-up."""
    import subprocess
    import sys
    script = tmp_path / "stall.py"
    script.write_text(_STALL_SCRIPT.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "no completion within" in r.stdout and "aborted" in r.stdout
    secs = float([l for l in r.stdout.splitlines() if l.startswith("DONE")][0].split()[1])
    # ncclCommAbort itself waits until RCCL's kernel — queued behind the rehearsal's 4 s hold-up — has started and seen the abort flag; with a peer that really
    # never arrives that kernel is already running. What must not happen is a wait without end (hipFree / hipDeviceSynchronize behind a receive nobody answers).
    assert secs < 20.0, r.stdout


def test_dist_setup_failure_releases_the_native_handle(monkeypatch):
    """a set-up phase that fails after 'create' used to leave the native handle (two device copies of the slab, plans, index lists) allocated with no
    object to close(). The constructor now releases what it built before the exception leaves it: free device memory returns to where it was."""
    from g4s_amd import dist as gdist
    n = 400000
    rp, ci, va = power_law_csr(n, n, 43, 200)
    d = [torch.from_numpy(a).cuda() for a in (rp, ci, va)]
    torch.cuda.synchronize()
    before_handles = len(gdist.DistSpMV._by_handle)
    free0 = torch.cuda.mem_get_info()[0]
    for hook in ("create:0", "pre:create:0"):
        monkeypatch.setenv("G4S_DIST_FAIL", hook)
        with pytest.raises(RuntimeError, match="G4S_DIST_FAIL"):
            gdist.DistSpMV([0, n], 0, 1, *d, n)
        torch.cuda.synchronize()
        assert len(gdist.DistSpMV._by_handle) == before_handles
        assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20), "the failed set-up left device memory behind"
    D = gdist.DistSpMV([0, n], 0, 1, *d, n)                    # (the hook is one-shot: the next set-up succeeds)
    assert D.h
    D.close()


def _column_worker(rank, world, port, kind, out_dir):
    """One rank of the column partition: the columns of its slab (every row), the partial y of all rows, the sum over the ranks by the process group."""
    import ctypes as C
    import scipy.sparse as sp
    import torch.distributed as dist
    init_gloo(rank, world, port)
    torch.cuda.set_device(0)
    from g4s_amd import capi
    from tests import oracle_lib
    lib = capi.load()
    rp, ci, va, n = _matrix(kind, oracle_lib.load())
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    cuts = [n * k // world for k in range(world + 1)]
    S = A[:, cuts[rank]:cuts[rank + 1]].tocsr()
    S.sort_indices()
    lrp, lci, lva = S.indptr.astype(np.int32), (S.indices + cuts[rank]).astype(np.int32), S.data.astype(np.float64)
    h = C.c_void_p()
    offs = (C.c_int64 * (world + 1))(*cuts)
    capi.check(lib.g4s_spmv_dist_create_columns(C.byref(h), rank, world, offs, n, lrp.ctypes.data, lci.ctypes.data if len(lci) else None, lva.ctypes.data if len(lva) else None, capi.HOST_POINTERS))
    info = capi.DistInfo()
    capi.check(lib.g4s_spmv_dist_get_info(h, C.byref(info)))
    assert info.reserved & 8 and info.connected == 1 and info.local_rows == n
    x = np.random.default_rng(3).uniform(-1, 1, n)
    xl = torch.from_numpy(x[cuts[rank]:cuts[rank + 1]]).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        capi.check(lib.g4s_spmv_dist_begin(h, C.c_void_p(xl.data_ptr()) if xl.numel() else None, y.data_ptr(), None))
        torch.cuda.synchronize()
        yc = y.cpu()                                               # the caller's transport: all-reduce of the partial products (gloo on the host)
        dist.all_reduce(yc)
        y.copy_(yc)
        capi.check(lib.g4s_spmv_dist_finish(h, y.data_ptr(), None))
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y.cpu().numpy())
    # wrong input: a column outside the slab is refused
    if len(lci):
        bad = lci.copy()
        bad[0] = cuts[rank + 1] if rank + 1 < world else cuts[rank] - 1 if rank > 0 else n
        h2 = C.c_void_p()
        assert lib.g4s_spmv_dist_create_columns(C.byref(h2), rank, world, offs, n, lrp.ctypes.data, bad.ctypes.data, lva.ctypes.data, capi.HOST_POINTERS) == capi.ERR_INVALID
    dist.barrier()
    capi.check(lib.g4s_spmv_dist_destroy(h))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(1, "powerlaw"), (2, "powerlaw"), (3, "powerlaw"), (3, "lap7")])
def test_dist_column_partition_matches_oracle(tmp_path, oracle, world, kind):
    """north_star's "all-reduce of partial products" (a correctness variant): y = Σ_g A[:, g]·x_g over a column partition, every rank ending with the
    whole y — equal to the oracle's product on every rank."""
    mp.spawn(_column_worker, args=(world, os.path.join(str(tmp_path), "rendezvous"), kind, str(tmp_path)), nprocs=world, join=True)
    rp, ci, va, n = _matrix(kind, oracle)
    x = np.random.default_rng(3).uniform(-1, 1, n)
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    for r in range(world):
        got = np.load(tmp_path / f"y{r}.npy")
        assert np.all(np.abs(got - want) <= TOL * asum + 1e-300), r


def test_dist_column_partition_rccl_all_reduce_single_rank(oracle):
    """The library's own path for the column partition: g4s_comm_create + g4s_spmv_dist_connect_rccl + g4s_spmv_dist_apply (one rank: the all-reduce is skipped,
    the wiring and the entry points are the multi-rank ones)."""
    import ctypes as C
    from g4s_amd import capi
    lib = capi.load()
    rp, ci, va, n = _matrix("powerlaw", oracle)
    h, comm = C.c_void_p(), C.c_void_p()
    offs = (C.c_int64 * 2)(0, n)
    capi.check(lib.g4s_spmv_dist_create_columns(C.byref(h), 0, 1, offs, n, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, capi.HOST_POINTERS))
    raw = (C.c_char * 128)()
    capi.check(lib.g4s_comm_unique_id(raw))
    capi.check(lib.g4s_comm_create(C.byref(comm), 1, 0, raw))
    capi.check(lib.g4s_spmv_dist_connect_rccl(h, comm))
    x = np.random.default_rng(3).uniform(-1, 1, n)
    xd, y = torch.from_numpy(x).cuda(), torch.empty(n, dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_spmv_dist_apply(h, xd.data_ptr(), y.data_ptr(), None))
    torch.cuda.synchronize()
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    assert np.all(np.abs(y.cpu().numpy() - want) <= TOL * asum + 1e-300)
    capi.check(lib.g4s_spmv_dist_destroy(h))
    capi.check(lib.g4s_comm_destroy(comm))
