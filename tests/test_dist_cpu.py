"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the library's own set-up logic — g4s_row_partition and g4s_dist_split_rows, the host
half of g4s_spmv_dist_create (csrc/dist.hip) — with the exchange carried by gloo and the local products done by the oracle. What runs on the
GPU box through RCCL is the same split, the same want lists and the same buffer layout; only the transport and the product kernels differ."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import power_law_csr, init_gloo


def _matrix(kind, o):
    if kind == "powerlaw":
        return power_law_csr(3000, 3000, 5, 900)
    if kind == "hubs":                                             # few own-column entries per slab → the merged (one product) form
        return power_law_csr(4000, 4000, 9, 1200)
    return o.laplacian7(9, 8, 7)


def _worker(rank, world, port, mode, kind, out_dir):
    init_gloo(rank, world, port)                                   # (a rendezvous FILE, tests/helpers.py)
    from g4s_amd import dist as gdist
    from tests import oracle_lib
    o = oracle_lib.load()
    rp, ci, va = _matrix(kind, o)
    n = len(rp) - 1
    rpt, cit, vat = torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(va)
    offs = gdist.row_partition(rpt, world)
    r0, r1 = offs[rank], offs[rank + 1]
    lrp, lci, lva = gdist.slice_rows(rpt, cit, vat, r0, r1)
    S = gdist.split_rows(offs, rank, world, lrp.numpy(), lci.numpy(), lva.numpy(), n, allgather=(mode == "allgather"))
    x = torch.from_numpy(o.vector(7, n))
    x_local = x[r0:r1].clone()
    x_rem = torch.full((max(S["n_ref"], 1),), float("nan"), dtype=torch.float64)
    cut = S["recv_cut"]
    if mode == "allgather":
        pad = int(S["pad"])
        mine = torch.zeros(pad, dtype=torch.float64)
        mine[:r1 - r0] = x_local
        slots = [torch.empty(pad, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(slots, mine)
        x_rem = torch.cat(slots)
        recv_bytes = 8 * pad * (world - 1)
    else:
        # every rank tells every owner which entries it wants (indices local to the owner's slab), then the entries travel
        want = [torch.from_numpy(S["want"][cut[k]:cut[k + 1]].copy()) for k in range(world)]
        counts = torch.tensor([w.numel() for w in want], dtype=torch.int64)
        allc = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
        give = [None] * world
        ops = []
        for k in range(world):
            if k == rank:
                continue
            if want[k].numel():
                ops.append(dist.P2POp(dist.isend, want[k], k))
            if int(allc[k][rank]):
                give[k] = torch.empty(int(allc[k][rank]), dtype=torch.int32)
                ops.append(dist.P2POp(dist.irecv, give[k], k))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        ops, send = [], {}
        for k in range(world):
            if k == rank:
                if cut[k + 1] > cut[k]:                            # merged form: own entries of the compact x, a local gather
                    x_rem[cut[k]:cut[k + 1]] = x_local[want[k].long()]
                continue
            if give[k] is not None:
                assert int(give[k].min()) >= 0 and int(give[k].max()) < r1 - r0
                send[k] = x_local[give[k].long()].contiguous()
                ops.append(dist.P2POp(dist.isend, send[k], k))
            if cut[k + 1] > cut[k]:
                ops.append(dist.P2POp(dist.irecv, x_rem[cut[k]:cut[k + 1]], k))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        recv_bytes = 8 * int(cut[world] - (cut[rank + 1] - cut[rank]))
    orp, oci, ova = S["own"]
    rrp, rci, rva = S["rem"]
    assert len(rci) == 0 or not torch.isnan(x_rem[torch.from_numpy(rci).long()]).any()   # every referenced entry has arrived
    xr = np.nan_to_num(x_rem.numpy())
    y_local = o.spmv(rrp, rci, rva, xr) if S["merged"] else o.spmv(orp, oci, ova, x_local.numpy()) + o.spmv(rrp, rci, rva, xr)
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y_local)
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([r0, r1, recv_bytes, S["n_ref"], int(S["merged"]), len(oci), len(rci)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,kind", [(2, "packed", "powerlaw"), (2, "packed", "lap7"), (3, "packed", "powerlaw"), (3, "packed", "hubs"),
                                             (1, "packed", "powerlaw"), (2, "allgather", "lap7"), (3, "allgather", "powerlaw"), (3, "allgather", "hubs"),
                                             (8, "packed", "powerlaw"), (8, "packed", "lap7"), (8, "allgather", "hubs")])   # the driver's widest launch: eight ranks
def test_partitioned_spmv_matches_single(tmp_path, oracle, world, mode, kind):
    mp.spawn(_worker, args=(world, os.path.join(str(tmp_path), "rendezvous"), mode, kind, str(tmp_path)), nprocs=world, join=True)
    rp, ci, va = _matrix(kind, oracle)
    n = len(rp) - 1
    x = oracle.vector(7, n)
    want = oracle.spmv(rp, ci, va, x)
    _, asum = oracle.spmv_ld(rp, ci, va, x)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    assert metas[0][0] == 0 and metas[-1][1] == n and all(metas[i][1] == metas[i + 1][0] for i in range(world - 1))
    assert sum(int(m[5] + m[6]) for m in metas) == len(ci)        # every entry is in exactly one part
    if all(m[4] for m in metas) or world == 1:
        assert np.array_equal(got, want)                           # merged form: one product, the CSR row's own order → bit-identical
    else:
        assert np.all(np.abs(got - want) <= 1e-10 * asum + 1e-300)  # own + remote part: two partial sums per row
    if kind == "lap7" and mode == "packed":
        # halo-only: a slab of the 7-point stencil needs at most one 9×8 plane from each neighbour
        assert all(m[2] <= 2 * 72 * 8 for m in metas)
    if mode == "packed" and world > 1:
        assert all(m[2] < 8 * n for m in metas)                    # fewer bytes than the whole-vector all-gather


def test_row_partition_equals_the_reference_rule(oracle):
    """a5: g4s_row_partition is BIN::set_rows_offset (mm/inc/BIN.h:101-122) — offsets equal, integer for integer, to the oracle's restatement
    on the same work vector (work = nnz + 1 per row, and an arbitrary per-row cost), for regular, power-law, empty-row and
    more-parts-than-rows inputs. Runs on the host: no GPU."""
    from g4s_amd import dist as gdist
    from tests.helpers import power_law_csr, random_csr
    cases = [random_csr(1000, 800, 0.01, 0, empty_rows=[0, 1, 2, 500, 999]), power_law_csr(5000, 5000, 3, 3000), oracle.laplacian5(40, 30),
             random_csr(5, 5, 0.5, 1), (np.zeros(8, np.int32), np.zeros(0, np.int32), np.zeros(0))]
    rng = np.random.default_rng(0)
    for rp, _, _ in cases:
        rows = len(rp) - 1
        work = (np.diff(rp).astype(np.int64) + 1)
        cost = rng.integers(0, 50, rows).astype(np.int64)
        for parts in (1, 2, 3, 4, 7, 8, 14, 64):
            want = oracle.rows_offset(work, parts)
            got = gdist.row_partition(torch.from_numpy(np.ascontiguousarray(rp)), parts)
            # BIN.h leaves the offsets as lower_bound gives them (non-decreasing by construction); g4s_row_partition also clamps to rows
            assert got == [min(int(v), rows) for v in want], (rows, parts)
            want = oracle.rows_offset(cost, parts)
            got = gdist.row_partition(torch.from_numpy(np.ascontiguousarray(rp)), parts, row_work=cost)
            assert got == [min(int(v), rows) for v in want], (rows, parts)


def test_row_partition_balances_work():
    from g4s_amd import dist as gdist
    rp, ci, va = power_law_csr(5000, 5000, 3, 2000)
    offs = gdist.row_partition(torch.from_numpy(rp), 4)
    work = np.diff(rp) + 1
    shares = [work[offs[i]:offs[i + 1]].sum() for i in range(4)]
    assert offs[0] == 0 and offs[-1] == 5000
    assert max(shares) <= sum(shares) / 4 + work.max()     # no part exceeds the average by more than one row's work


def test_split_rows_rejects_bad_input():
    """the host set-up validates before it indexes — a non-monotone rowptr, a negative nnz or an out-of-range column is
    G4S_ERR_INVALID, not a host crash (the same checks g4s_csr_create makes)."""
    from g4s_amd import capi
    lib = capi.load()
    offs = (C.c_int64 * 3)(0, 2, 4)
    sp = capi.DistSplit()
    ci = np.array([0, 1, 2, 3], np.int32)
    va = np.ones(4)

    def call(rp, cols=ci, n_cols=4, flags=0):
        rp = np.asarray(rp, np.int32)
        return lib.g4s_dist_split_rows(0, 2, offs, n_cols, rp.ctypes.data, cols.ctypes.data, va.ctypes.data, flags, C.byref(sp))

    assert call([0, 2, 4]) == capi.OK
    assert sp.local_rows == 2 and sp.nnz_own + sp.nnz_rem == 4
    lib.g4s_dist_split_free(C.byref(sp))
    assert call([1, 2, 4]) == capi.ERR_INVALID                     # rowptr[0] != 0
    assert call([0, 3, 2]) == capi.ERR_INVALID and b"decreases" in lib.g4s_last_error()
    assert call([0, 2, -1]) == capi.ERR_INVALID
    assert call([0, 2, 4], cols=np.array([0, 1, 2, 4], np.int32)) == capi.ERR_INVALID and b"column" in lib.g4s_last_error()
    assert call([0, 2, 4], cols=np.array([0, -1, 2, 3], np.int32)) == capi.ERR_INVALID
    assert call([0, 2, 4], n_cols=5) == capi.ERR_INVALID           # row_offsets[world] != n_cols
    assert call([0, 2, 4], flags=capi.DEVICE_POINTERS) == capi.ERR_INVALID
    bad = (C.c_int64 * 3)(0, 3, 2)
    assert lib.g4s_dist_split_rows(0, 2, bad, 2, np.zeros(4, np.int32).ctypes.data, None, None, 0, C.byref(sp)) == capi.ERR_INVALID
    part = (C.c_int64 * 3)()
    assert lib.g4s_row_partition(2, np.array([0, 3, 1], np.int32).ctypes.data, None, 1, 2, part, 0) == capi.ERR_INVALID
    assert lib.g4s_row_partition(2, None, np.array([1, -1], np.int64).ctypes.data, 0, 2, part, 0) == capi.ERR_INVALID
