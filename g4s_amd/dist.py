"""1-D row partition of a CSR matrix over ranks and the per-SpMV vector exchange (one process per GPU, torch.distributed).

The reference has no multi-device code on this path; its shared-memory analogue is the equal-work contiguous row split of
BIN::set_rows_offset (mm/inc/BIN.h:101-122: prefix-sum the per-row work, boundary t = lower_bound(prefix, ceil(total/parts)·t)),
and its distributed analogue is CitcomS's per-matvec neighbour exchange (citcoms/lib/Regional_parallel_related.c:744-789).
Here the same split rule assigns rows to GPUs; each rank owns rows [r0,r1) of A, of y and of x, and before each SpMV receives
the x entries it references from their owners:
  * "allgatherv": every slab to every rank (grouped send/recv — exact slab sizes, each peer's slab over its own xGMI link);
  * "needed": only the column ranges a rank's rows actually reference (computed once at setup; for stencil/banded matrices
    this is the halo, a few planes, instead of the whole vector — SURVEY.md §8e).
This module is device-agnostic torch code (the tests drive it with gloo on CPU tensors); the SpMV itself is g4s_amd.host.CSR.
"""
import os

import torch
import torch.distributed as dist


def row_partition(rowptr, parts):
    """Row offsets [parts+1] giving each part an equal share of work = nnz + rows (BIN.h:101-122 rule)."""
    rows = rowptr.numel() - 1
    rp = rowptr.to(torch.int64)
    prefix = rp + torch.arange(rows + 1, device=rowptr.device, dtype=torch.int64)   # work prefix: nnz before row + rows before row
    total = int(prefix[-1].item())
    avg = (total + parts - 1) // parts
    targets = torch.arange(1, parts + 1, device=rowptr.device, dtype=torch.int64) * avg
    cut = torch.searchsorted(prefix, targets, right=False)                            # std::lower_bound
    offs = [0] + [min(int(c), rows) for c in cut.tolist()]
    offs[parts] = rows                                                                # BIN.h:120
    for i in range(1, parts + 1):
        offs[i] = max(offs[i], offs[i - 1])
    return offs


def slice_rows(rowptr, colids, values, r0, r1):
    """Rows [r0,r1) as a local CSR with global column ids."""
    k0, k1 = int(rowptr[r0].item()), int(rowptr[r1].item())
    rp = (rowptr[r0:r1 + 1] - rowptr[r0]).to(torch.int32).contiguous()
    return rp, colids[k0:k1].contiguous(), values[k0:k1].contiguous()


def _gloo_on_device(t, group=None):
    """gloo moves device tensors from its own host threads with no regard for the HIP stream that produced them (the rehearsal set-up:
    several ranks on one GPU). Kernels that wrote the send buffers must have finished before the sends are posted; RCCL is
    stream-ordered and needs nothing."""
    return t.is_cuda and dist.is_initialized() and dist.get_backend(group) != "nccl"


class VectorExchange:
    """Brings the x entries this rank's rows reference into a full-length local buffer.

    offsets: the row partition (rank k owns x[offsets[k]:offsets[k+1]]).
    mode "allgatherv": receive every peer's whole slab (grouped send/recv, exact sizes). mode "allgather": the same data through one
    padded all_gather_into_tensor. mode "needed": receive, from each peer, only the contiguous range [lo,hi) of its slab that local
    columns touch (empty ranges are skipped)."""

    def __init__(self, offsets, rank, world, colids=None, mode="allgatherv", group=None):
        self.offsets, self.rank, self.world, self.group = list(offsets), rank, world, group
        self.mode = mode
        # want[k] = (lo,hi) global range this rank needs from rank k
        self.want = [(self.offsets[k], self.offsets[k + 1]) for k in range(world)]
        if mode == "needed":
            assert colids is not None
            self.want = []
            for k in range(world):
                lo, hi = self.offsets[k], self.offsets[k + 1]
                if k == rank or hi <= lo or colids.numel() == 0:
                    self.want.append((lo, lo))
                    continue
                m = (colids >= lo) & (colids < hi)
                if bool(m.any()):
                    c = colids[m]
                    self.want.append((int(c.min().item()), int(c.max().item()) + 1))
                else:
                    self.want.append((lo, lo))
        # give[k] = (lo,hi) global range rank k needs from this rank: exchange the wish lists once
        self.give = [(0, 0)] * world
        if world > 1:
            mine = torch.tensor([list(w) for w in self.want], dtype=torch.int64)
            allw = [torch.zeros_like(mine) for _ in range(world)]
            dev = None
            if dist.get_backend(group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
                mine = mine.to(dev)
                allw = [a.to(dev) for a in allw]
            dist.all_gather(allw, mine, group=group)
            self.give = [tuple(int(v) for v in allw[k][rank].tolist()) for k in range(world)]
        self.recv_bytes = sum(8 * (hi - lo) for k, (lo, hi) in enumerate(self.want) if k != rank)

    def __call__(self, x_local, x_full):
        """x_full[own slab] = x_local, and the wanted ranges of the peers' slabs arrive by grouped send/recv (mode "allgather": one
        all_gather_into_tensor of slabs padded to the longest one, then one copy per slab — a single collective, no point-to-point)."""
        r0, r1 = self.offsets[self.rank], self.offsets[self.rank + 1]
        x_full[r0:r1].copy_(x_local)
        if self.world == 1:
            return x_full
        if self.mode == "allgather":
            longest = max(self.offsets[k + 1] - self.offsets[k] for k in range(self.world))
            if getattr(self, "_pad", None) is None or self._pad.numel() != longest or self._pad.device != x_local.device:
                self._pad = torch.zeros(longest, dtype=x_local.dtype, device=x_local.device)
                self._all = torch.empty(longest * self.world, dtype=x_local.dtype, device=x_local.device)
            self._pad[:r1 - r0].copy_(x_local)
            if _gloo_on_device(x_local, self.group):
                torch.cuda.synchronize()
            dist.all_gather_into_tensor(self._all, self._pad, group=self.group)
            for k in range(self.world):
                if k != self.rank:
                    lo, hi = self.offsets[k], self.offsets[k + 1]
                    x_full[lo:hi].copy_(self._all[k * longest:k * longest + hi - lo])
            return x_full
        ops = []
        for k in range(self.world):
            if k == self.rank:
                continue
            lo, hi = self.give[k]
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, x_local[lo - r0:hi - r0], k, group=self.group))
            lo, hi = self.want[k]
            if hi > lo:
                ops.append(dist.P2POp(dist.irecv, x_full[lo:hi], k, group=self.group))
        if ops:
            if _gloo_on_device(x_local, self.group):
                torch.cuda.synchronize()
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return x_full


class CompactExchange:
    """Exchange for matrices without column locality (power-law graphs): a rank's rows reference only part of every peer's slab,
    scattered all over it, so the slab's columns are renumbered 0 … n_ref−1 in ascending global order (`local_colids`, to build the
    local CSR with `cols = n_ref`) and only the referenced entries travel: every owner gathers the entries a peer asked for at set-up
    (`x_local[give_idx]`) and sends them packed; they land in the contiguous segment of `x_compact` that belongs to that owner — no
    scatter on the receiving side. Against the whole-slab all-gather this moves n_ref·8 bytes per rank instead of cols·8, and the local
    product stages n_ref/16 K column bands of x instead of cols/16 K.

    offsets: the row partition (rank k owns x[offsets[k]:offsets[k+1]]); colids: the global column ids of this rank's rows."""

    def __init__(self, offsets, rank, world, colids, group=None):
        self.offsets, self.rank, self.world, self.group = list(offsets), rank, world, group
        dev = colids.device
        ref = torch.unique(colids.long())                                        # sorted referenced columns
        self.n_ref = int(ref.numel())
        self.local_colids = torch.bucketize(colids.long(), ref).to(torch.int32)   # position of every entry's column in ref
        bounds = torch.searchsorted(ref, torch.tensor(self.offsets, dtype=torch.int64, device=dev)).tolist()
        self.seg = [(bounds[k], bounds[k + 1]) for k in range(world)]            # segment of x_compact owned by rank k
        want_idx = [(ref[lo:hi] - self.offsets[k]).contiguous() for k, (lo, hi) in enumerate(self.seg)]   # local indices at the owner
        self.own_idx = want_idx[rank]
        self.give_idx = [None] * world
        if world > 1:
            nccl = dist.get_backend(group) == "nccl"
            cdev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
            mine = torch.tensor([w.numel() for w in want_idx], dtype=torch.int64, device=cdev)
            allw = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allw, mine, group=group)
            give_count = [int(allw[k][rank].item()) for k in range(world)]
            ops, keep = [], []
            for k in range(world):
                if k == rank:
                    continue
                if want_idx[k].numel():
                    buf = want_idx[k].to(cdev)
                    keep.append(buf)
                    ops.append(dist.P2POp(dist.isend, buf, k, group=group))
                if give_count[k]:
                    self.give_idx[k] = torch.empty(give_count[k], dtype=torch.int64, device=cdev)
                    ops.append(dist.P2POp(dist.irecv, self.give_idx[k], k, group=group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            self.give_idx = [g.to(dev) if g is not None else None for g in self.give_idx]
        self.recv_bytes = sum(8 * (hi - lo) for k, (lo, hi) in enumerate(self.seg) if k != rank)
        # one gather per step fills the send buffer of every peer: peer k's packed entries are _send_all[_cut[k]:_cut[k+1]]
        sizes = [g.numel() if g is not None else 0 for g in self.give_idx]
        self._cut = [0]
        for n in sizes:
            self._cut.append(self._cut[-1] + n)
        parts = [g for g in self.give_idx if g is not None]
        self._give_all = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=dev)
        self._send_all = torch.empty(self._cut[-1], dtype=torch.float64, device=dev)

    def __call__(self, x_local, x_compact):
        lo, hi = self.seg[self.rank]
        if hi > lo:
            torch.index_select(x_local, 0, self.own_idx, out=x_compact[lo:hi])
        if self.world == 1:
            return x_compact
        if self._cut[-1]:
            torch.index_select(x_local, 0, self._give_all, out=self._send_all)
        ops = []
        for k in range(self.world):
            if k == self.rank:
                continue
            if self._cut[k + 1] > self._cut[k]:
                ops.append(dist.P2POp(dist.isend, self._send_all[self._cut[k]:self._cut[k + 1]], k, group=self.group))
            lo, hi = self.seg[k]
            if hi > lo:
                ops.append(dist.P2POp(dist.irecv, x_compact[lo:hi], k, group=self.group))
        if ops:
            if _gloo_on_device(x_local, self.group):
                torch.cuda.synchronize()
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return x_compact


def _all_reduce_sum(t, group=None):
    """Element-wise sum over the ranks, in place. RCCL reduces device tensors directly; gloo (the CPU rehearsal backend) goes through host memory."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, group=group)
    else:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)


def dist_conj_grad(A_local, exchange, BI_local, F_local, zero_resid_local, acc, steps, group=None):
    """Jacobi-CG (conj_grad, citcoms/lib/General_matrix_functions.c:307-424) on a row-partitioned operator: this rank owns the rows
    [offsets[rank], offsets[rank+1]) of A (a host.CSR with all columns), the matching slabs of BI, F and the solution, and the local
    indices of its boundary rows. Per iteration: one exchange of the direction vector (`exchange`, a VectorExchange), one local
    g4s_spmv, two all-reduces of 256 partial sums. The vector kernels and the termination test are the library's
    (g4s_cg_* step API); torch only moves bytes. Returns (d0_local, iterations, residual)."""
    import ctypes as C
    from . import capi, host
    lib = capi.load()
    n = A_local.rows
    dev = F_local.device
    ws = C.c_void_p()
    capi.check(lib.g4s_cg_ws_create(C.byref(ws), n))
    try:
        d0 = torch.empty(n, dtype=torch.float64, device=dev)
        zr = zero_resid_local if zero_resid_local is not None and zero_resid_local.numel() else None
        nz = int(zr.numel()) if zr is not None else 0
        st = host._stream()
        capi.check(lib.g4s_cg_begin(ws, host._ptr(F_local), host._ptr(BI_local), host._ptr(d0), host._ptr(zr) if zr is not None else None, nz, st))
        p_ptr, Ap_ptr, part_ptr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(lib.g4s_cg_buffers(ws, C.byref(p_ptr), C.byref(Ap_ptr), C.byref(part_ptr)))
        part = host.view_f64(part_ptr, 768, dev)
        p_full = torch.zeros(A_local.cols, dtype=torch.float64, device=dev)
        count, done, residual = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        _all_reduce_sum(part, group)                              # r·z and r·r of the start vector
        while True:
            capi.check(lib.g4s_cg_direction(ws, int(steps), float(acc), st))
            capi.check(lib.g4s_cg_state(ws, C.byref(count), C.byref(done), C.byref(residual), st))
            if done.value:
                break
            capi.check(lib.g4s_cg_buffers(ws, C.byref(p_ptr), C.byref(Ap_ptr), None))
            p, Ap = host.view_f64(p_ptr, n, dev), host.view_f64(Ap_ptr, n, dev)
            exchange(p, p_full)
            A_local.spmv(p_full, Ap)
            capi.check(lib.g4s_cg_reduce_pAp(ws, st))
            _all_reduce_sum(part[256:512], group)
            capi.check(lib.g4s_cg_update(ws, host._ptr(BI_local), host._ptr(d0), st))
            _all_reduce_sum(part, group)                          # [0,256) r·z and [512,768) r·r; the middle third is rewritten before its next use
        capi.check(lib.g4s_cg_end(ws, host._ptr(d0), host._ptr(zr) if zr is not None else None, nz, st))
        return d0, count.value, residual.value
    finally:
        lib.g4s_cg_ws_destroy(ws)


class DistSpMV:
    """The row-partitioned product behind the C-ABI (g4s_spmv_dist_*, csrc/dist.hip): own-column / remote-column split, packed exchange of
    only the referenced x entries, own-column product overlapped with the exchange. This class only wires it to a transport:

      * backend "nccl": the library's own RCCL communicator (g4s_comm_create; rank 0's 128-byte id reaches the others through one
        torch.distributed broadcast) — ncclSend/ncclRecv issued by the library itself, `apply` is one C call;
      * any other backend (gloo on the test boxes): the packed buffers the library exposes travel through torch.distributed
        point-to-point between g4s_spmv_dist_begin and g4s_spmv_dist_finish.

    rowptr / colids / values: this rank's rows [offsets[rank], offsets[rank+1]) with GLOBAL column ids, device tensors."""

    def __init__(self, offsets, rank, world, rowptr, colids, values, n_cols, spmv_flags=0, group=None, loopback=False):
        import ctypes as C
        from . import capi, host
        self._C, self._capi, self._host = C, capi, host
        self.lib = capi.load()
        self.offsets, self.rank, self.world, self.group = [int(v) for v in offsets], rank, world, group
        self.n_local = self.offsets[rank + 1] - self.offsets[rank]
        self.h = C.c_void_p()
        offs = (C.c_int64 * (world + 1))(*self.offsets)
        torch.cuda.current_stream().synchronize()
        flags = capi.DEVICE_POINTERS | spmv_flags | (capi.DIST_LOOPBACK if loopback else 0)
        self.comm = None
        self.rccl = dist.is_initialized() and dist.get_backend(group) == "nccl" if (world > 1) else bool(loopback)
        # Set-up is a sequence of phases, the later ones collective. After each one the ranks agree (one all-reduce of a flag) whether it
        # succeeded EVERYWHERE; if not, every rank raises at the same point — none is left waiting inside a collective the others never enter.
        self._phase("create", lambda: capi.check(self.lib.g4s_spmv_dist_create(
            C.byref(self.h), rank, world, offs, int(n_cols), host._ptr(rowptr), host._ptr(colids), host._ptr(values), flags)))
        if self.rccl:
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_char * 128)()
                capi.check(self.lib.g4s_comm_unique_id(raw))
                idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            if world > 1:
                dev = torch.device("cuda", torch.cuda.current_device())
                t = idbuf.to(dev)
                dist.broadcast(t, 0, group=group)
                idbuf = t.cpu()
            raw = (C.c_char * 128).from_buffer_copy(bytes(idbuf.numpy().tobytes()))
            self.comm = C.c_void_p()
            self._phase("communicator", lambda: capi.check(self.lib.g4s_comm_create(C.byref(self.comm), world, rank, raw)))
            self._phase("connect", lambda: capi.check(self.lib.g4s_spmv_dist_connect_rccl(self.h, self.comm)))
        elif world > 1:
            self._wire_by_torch()
        self._views = None

    def _phase(self, name, fn):
        err = None
        try:
            if os.environ.get("G4S_DIST_FAIL") == f"{name}:{self.rank}":      # test hook: this phase "fails" on this rank
                raise RuntimeError("G4S_DIST_FAIL")
            fn()
        except Exception as e:                                               # noqa: BLE001 — re-raised below, on every rank
            err = e
        if self.world > 1 and dist.is_initialized():
            on_gpu = dist.get_backend(self.group) == "nccl"
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 0 and err is None:
                err = RuntimeError(f"g4s_spmv_dist set-up: phase '{name}' failed on another rank")
        if err is not None:
            raise err

    # -- set-up over torch.distributed point-to-point: every rank tells every owner which entries it wants
    def _wire_by_torch(self):
        C, capi = self._C, self._capi
        dev = torch.device("cuda", torch.cuda.current_device())
        cdev = torch.device("cpu") if dist.get_backend(self.group) != "nccl" else dev
        want = []
        for k in range(self.world):
            n, p = C.c_int64(), C.c_void_p()
            capi.check(self.lib.g4s_spmv_dist_want(self.h, k, C.byref(n), C.byref(p)))
            want.append(self._host.view_i32(p, n.value, dev).to(cdev) if n.value else torch.empty(0, dtype=torch.int32, device=cdev))
        counts = torch.tensor([w.numel() for w in want], dtype=torch.int64, device=cdev)
        allc = [torch.zeros_like(counts) for _ in range(self.world)]
        dist.all_gather(allc, counts, group=self.group)
        give_n = [int(allc[k][self.rank].item()) for k in range(self.world)]
        ops, recv = [], [None] * self.world
        for k in range(self.world):
            if k == self.rank:
                continue
            if want[k].numel():
                ops.append(dist.P2POp(dist.isend, want[k], k, group=self.group))
            if give_n[k]:
                recv[k] = torch.empty(give_n[k], dtype=torch.int32, device=cdev)
                ops.append(dist.P2POp(dist.irecv, recv[k], k, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for k in range(self.world):
            if k == self.rank:
                continue
            g = recv[k].cpu().contiguous() if recv[k] is not None else torch.empty(0, dtype=torch.int32)
            capi.check(self.lib.g4s_spmv_dist_set_give(self.h, k, g.numel(), C.c_void_p(g.data_ptr()) if g.numel() else None, capi.HOST_POINTERS))

    def info(self):
        i = self._capi.DistInfo()
        self._capi.check(self.lib.g4s_spmv_dist_get_info(self.h, self._C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def _buffers(self):
        if self._views is None:
            C = self._C
            sp, rp = C.c_void_p(), C.c_void_p()
            sc, rc = C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
            self._capi.check(self.lib.g4s_spmv_dist_buffers(self.h, C.byref(sp), C.byref(sc), C.byref(rp), C.byref(rc)))
            scut, rcut = [sc[k] for k in range(self.world + 1)], [rc[k] for k in range(self.world + 1)]
            dev = torch.device("cuda", torch.cuda.current_device())
            self._views = (self._host.view_f64(sp, max(scut[-1], 1), dev), scut, self._host.view_f64(rp, max(rcut[-1], 1), dev), rcut)
        return self._views

    def __call__(self, x_local, y_local=None):
        """y_local = (A·x)[own rows]."""
        host, capi = self._host, self._capi
        if y_local is None:
            y_local = torch.empty(self.n_local, dtype=torch.float64, device=x_local.device)
        st = host._stream()
        if self.rccl or self.world == 1:
            capi.check(self.lib.g4s_spmv_dist_apply(self.h, host._ptr(x_local), host._ptr(y_local), st))
            return y_local
        capi.check(self.lib.g4s_spmv_dist_begin(self.h, host._ptr(x_local), host._ptr(y_local), st))
        send, scut, recv, rcut = self._buffers()
        ops = []
        for k in range(self.world):
            if k == self.rank:
                continue
            if scut[k + 1] > scut[k]:
                ops.append(dist.P2POp(dist.isend, send[scut[k]:scut[k + 1]], k, group=self.group))
            if rcut[k + 1] > rcut[k]:
                ops.append(dist.P2POp(dist.irecv, recv[rcut[k]:rcut[k + 1]], k, group=self.group))
        if ops:
            if _gloo_on_device(x_local, self.group):
                torch.cuda.synchronize()
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        capi.check(self.lib.g4s_spmv_dist_finish(self.h, host._ptr(y_local), st))
        return y_local

    def allreduce_sum(self, t):
        """Sum over the ranks, in place: the library's RCCL communicator when there is one, torch.distributed otherwise."""
        if self.comm is not None and self.world > 1:
            self._capi.check(self.lib.g4s_comm_allreduce_sum_f64(self.comm, self._host._ptr(t), t.numel(), self._host._stream()))
        else:
            _all_reduce_sum(t, self.group)

    def conj_grad(self, BI_local, F_local, zero_resid_local, acc, steps):
        """Jacobi-CG on the partitioned operator in one C call (g4s_conj_grad_dist: product and dot products on the library's RCCL
        communicator). Needs the RCCL wiring (backend nccl, or loopback). Returns (d0_local, iterations, residual)."""
        C, capi, host = self._C, self._capi, self._host
        if self.comm is None:
            raise RuntimeError("DistSpMV.conj_grad needs the library's RCCL communicator (backend nccl); over gloo use dist_conj_grad")
        d0 = torch.empty(self.n_local, dtype=torch.float64, device=F_local.device)
        zr = zero_resid_local if zero_resid_local is not None and zero_resid_local.numel() else None
        cycles, res = C.c_int32(0), C.c_double(0.0)
        capi.check(self.lib.g4s_conj_grad_dist(self.h, self.comm, self.n_local, host._ptr(BI_local), host._ptr(zr) if zr is not None else None,
                                               int(zr.numel()) if zr is not None else 0, host._ptr(F_local), host._ptr(d0), float(acc), int(steps),
                                               C.byref(cycles), C.byref(res), host._stream()))
        return d0, cycles.value, res.value

    def close(self):
        if self.h:
            self.lib.g4s_spmv_dist_destroy(self.h)
            self.h = None
        if self.comm is not None:
            self.lib.g4s_comm_destroy(self.comm)
            self.comm = None
