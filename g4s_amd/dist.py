"""Row-partitioned SpMV over several ranks: thin wiring around the C-ABI (g4s_row_partition, g4s_dist_split_rows, g4s_spmv_dist_*, csrc/dist.hip).

The reference has no multi-device code on this path; its shared-memory analogue is the equal-work contiguous row split of
BIN::set_rows_offset (mm/inc/BIN.h:101-122) and its distributed analogue CitcomS's per-mat-vec neighbour exchange
(citcoms/lib/Regional_parallel_related.c:744-789). Everything that decides or moves data lives in the library; this module
  * asks it for the partition (row_partition) and, in tests, for one rank's own / remote split on the host (split_rows);
  * carries the library's packed buffers through torch.distributed when the process group is not RCCL (gloo on the test boxes);
  * drives the library's CG step API around the distributed product (dist_conj_grad) where there is no RCCL communicator.
torch only holds memory and provides the rendezvous.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist


def row_partition(rowptr, parts, row_weight=1, row_work=None):
    """Row offsets [parts+1] giving each part an equal share of work (g4s_row_partition = the rule of BIN.h:101-122): work per row =
    row_work[i] if given, else nnz_i + row_weight. rowptr: int32 tensor (host or device)."""
    from . import capi
    lib = capi.load()
    rows = rowptr.numel() - 1
    offs = (C.c_int64 * (parts + 1))()
    work = None if row_work is None else np.ascontiguousarray(row_work, dtype=np.int64)
    rp = rowptr.contiguous()
    if rp.is_cuda:
        torch.cuda.current_stream().synchronize()
    capi.check(lib.g4s_row_partition(rows, C.c_void_p(rp.data_ptr()), None if work is None else C.c_void_p(work.ctypes.data), int(row_weight), parts, offs,
                                     capi.DEVICE_POINTERS if rp.is_cuda else capi.HOST_POINTERS))
    return [int(v) for v in offs]


def slice_rows(rowptr, colids, values, r0, r1):
    """Local CSR (rows r0:r1, all columns) of a global CSR held by every rank."""
    k0, k1 = int(rowptr[r0].item()), int(rowptr[r1].item())
    lrp = (rowptr[r0:r1 + 1] - rowptr[r0]).contiguous()
    return lrp, colids[k0:k1].contiguous(), values[k0:k1].contiguous()


def split_rows(offsets, rank, world, rowptr, colids, values, n_cols, allgather=False, loopback=False):
    """g4s_dist_split_rows on host arrays (numpy): this rank's rows cut into the own-column and the remote-column part, as the library does it
    before it uploads. Needs no GPU. Returns a dict of numpy arrays and scalars."""
    from . import capi
    lib = capi.load()
    rp, ci, va = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colids, np.int32), np.ascontiguousarray(values, np.float64))
    offs = (C.c_int64 * (world + 1))(*[int(v) for v in offsets])
    sp = capi.DistSplit()
    flags = (capi.DIST_ALLGATHER if allgather else 0) | (capi.DIST_LOOPBACK if loopback else 0)
    capi.check(lib.g4s_dist_split_rows(rank, world, offs, int(n_cols), rp.ctypes.data, ci.ctypes.data, va.ctypes.data, flags, C.byref(sp)))
    try:
        m = sp.local_rows
        arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True)
        return {"local_rows": m, "n_ref": sp.n_ref, "merged": bool(sp.merged), "allgather": bool(sp.allgather), "pad": sp.pad,
                "own": (arr(sp.own_rowptr, m + 1, np.int32), arr(sp.own_colids, sp.nnz_own, np.int32), arr(sp.own_values, sp.nnz_own, np.float64)),
                "rem": (arr(sp.rem_rowptr, m + 1, np.int32), arr(sp.rem_colids, sp.nnz_rem, np.int32), arr(sp.rem_values, sp.nnz_rem, np.float64)),
                "want": arr(sp.want, 0 if sp.allgather else sp.n_ref, np.int32), "recv_cut": arr(sp.recv_cut, world + 1, np.int64)}
    finally:
        lib.g4s_dist_split_free(C.byref(sp))


def _gloo_on_device(t, group=None):
    """gloo moves device tensors through host staging on ITS OWN stream: the producing stream must be complete first."""
    return t.is_cuda and dist.is_initialized() and dist.get_backend(group) != "nccl"


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


def dist_conj_grad(D, BI_local, F_local, zero_resid_local, acc, steps):
    """Jacobi-CG (conj_grad, citcoms/lib/General_matrix_functions.c:307-424) on a row-partitioned operator WITHOUT an RCCL communicator (gloo
    rehearsals): the library's CG step API (g4s_cg_*) around the library's distributed product D (a DistSpMV), the 256 partial sums of every
    dot product summed over torch.distributed. With RCCL the whole solve is one C call: DistSpMV.conj_grad → g4s_conj_grad_dist.
    Returns (d0_local, iterations, residual)."""
    from . import capi, host
    lib = capi.load()
    n = D.n_local
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
        count, done, residual = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        _all_reduce_sum(part, D.group)                            # r·z and r·r of the start vector
        while True:
            capi.check(lib.g4s_cg_direction(ws, int(steps), float(acc), st))
            capi.check(lib.g4s_cg_state(ws, C.byref(count), C.byref(done), C.byref(residual), st))
            if done.value:
                break
            capi.check(lib.g4s_cg_buffers(ws, C.byref(p_ptr), C.byref(Ap_ptr), None))
            p, Ap = host.view_f64(p_ptr, n, dev), host.view_f64(Ap_ptr, n, dev)
            D(p, Ap)
            capi.check(lib.g4s_cg_reduce_pAp(ws, st))
            _all_reduce_sum(part[256:512], D.group)
            capi.check(lib.g4s_cg_update(ws, host._ptr(BI_local), host._ptr(d0), st))
            _all_reduce_sum(part, D.group)                        # [0,256) r·z and [512,768) r·r; the middle third is rewritten before its next use
        capi.check(lib.g4s_cg_end(ws, host._ptr(d0), host._ptr(zr) if zr is not None else None, nz, st))
        return d0, count.value, residual.value
    finally:
        lib.g4s_cg_ws_destroy(ws)


class TorchTransport:
    """g4s_transport over torch.distributed for process groups that are not RCCL (gloo rehearsals): the C loops of the library
    (g4s_conj_grad_dist_tr, g4s_stokes_uzawa_cg_dist) call back here for the dot products' all-reduce and for the exchange half of a
    distributed product. With RCCL use g4s_transport_rccl instead — no Python in the loop."""

    def __init__(self, group=None):
        from . import capi, host
        self.group = group

        def allreduce(ctx, buf, count, stream):
            try:
                t = host.view_f64(C.c_void_p(buf), int(count), torch.device("cuda", torch.cuda.current_device()))
                torch.cuda.current_stream().synchronize()
                _all_reduce_sum(t, self.group)
                return 0
            except Exception as e:                                 # noqa: BLE001 — a Python exception must not unwind through C
                print(f"TorchTransport.allreduce failed: {e}", flush=True)
                return capi.ERR_HIP

        def exchange(ctx, h, stream):
            try:
                DistSpMV._by_handle[h].exchange_by_torch()
                return 0
            except Exception as e:                                 # noqa: BLE001
                print(f"TorchTransport.exchange failed: {e}", flush=True)
                return capi.ERR_HIP

        self._cbs = (capi.ALLREDUCE_CB(allreduce), capi.EXCHANGE_CB(exchange))   # kept alive with the object
        self.struct = capi.Transport(None, self._cbs[0], self._cbs[1])


def rccl_transport(comm):
    from . import capi
    t = capi.Transport()
    capi.check(capi.load().g4s_transport_rccl(comm, C.byref(t)))
    return t


def stokes_uzawa_dist(K, D, Dt, transport, BI, BPI, vmass, area, volume, zero_resid, F, V, P, params, hist_lines=0):
    """g4s_stokes_uzawa_cg_dist on this rank's slabs (device tensors; K, D, Dt: DistSpMV over the equation / element partitions).
    Returns (StokesResult, hist array or None); V and P are updated in place."""
    from . import capi, host
    res = capi.StokesResult()
    hist = np.zeros(5 * hist_lines) if hist_lines else None
    zr = zero_resid if zero_resid is not None and zero_resid.numel() else None
    tr = transport.struct if hasattr(transport, "struct") else transport
    capi.check(capi.load().g4s_stokes_uzawa_cg_dist(K.h, D.h, Dt.h, C.byref(tr), K.n_local, D.n_local, host._ptr(BI), host._ptr(BPI), host._ptr(vmass), host._ptr(area),
                                                    float(volume), host._ptr(zr) if zr is not None else None, int(zr.numel()) if zr is not None else 0, host._ptr(F),
                                                    host._ptr(V), host._ptr(P), C.byref(params), C.byref(res), hist.ctypes.data if hist is not None else None, hist_lines,
                                                    host._stream()))
    return res, hist


class DistSpMV:
    """The row-partitioned product behind the C-ABI (g4s_spmv_dist_*, csrc/dist.hip): own-column / remote-column split, packed exchange of
    only the referenced x entries, own-column product overlapped with the exchange. This class only wires it to a transport:

      * backend "nccl": the library's own RCCL communicator (g4s_comm_create; rank 0's 128-byte id reaches the others through one
        torch.distributed broadcast) — ncclSend/ncclRecv issued by the library itself, `apply` is one C call;
      * any other backend (gloo on the test boxes): the packed buffers the library exposes travel through torch.distributed
        point-to-point between g4s_spmv_dist_begin and g4s_spmv_dist_finish.

    rowptr / colids / values: this rank's rows [offsets[rank], offsets[rank+1]) with GLOBAL column ids, device tensors."""

    _by_handle = {}                                                # handle value → DistSpMV, for the transport's exchange callback

    def __init__(self, offsets, rank, world, rowptr, colids, values, n_cols, spmv_flags=0, group=None, loopback=False, exchange="packed", col_offsets=None):
        """col_offsets: the partition of x when it differs from the rows' (a rectangular operator, g4s_spmv_dist_create_rect)."""
        from . import capi, host
        self._capi, self._host = capi, host
        self.lib = capi.load()
        self.offsets, self.rank, self.world, self.group = [int(v) for v in offsets], rank, world, group
        self.n_local = self.offsets[rank + 1] - self.offsets[rank]
        self.h = C.c_void_p()
        offs = (C.c_int64 * (world + 1))(*self.offsets)
        torch.cuda.current_stream().synchronize()
        assert exchange in ("packed", "allgather")
        self.allgather = exchange == "allgather"                   # ONE ncclAllGather of the padded slabs instead of packed point-to-point messages
        flags = capi.DEVICE_POINTERS | spmv_flags | (capi.DIST_LOOPBACK if loopback else 0) | (capi.DIST_ALLGATHER if self.allgather else 0)
        self.comm = None
        self._views = None
        self.rccl = dist.is_initialized() and dist.get_backend(group) == "nccl" if (world > 1) else bool(loopback)
        # Set-up is a sequence of phases, the later ones collective. BEFORE a collective phase the ranks agree that all of them are about to enter it, and
        # after each phase whether it succeeded everywhere (one all-reduce of a flag each); if not, every rank raises at the same point — none is left
        # waiting inside a collective the others never enter. Whatever a failed set-up had already built (the native handle with its two device copies of
        # the slab, the communicator) is released before the exception leaves the constructor: the caller never gets an object it could close().
        try:
            self._setup(offs, rank, world, rowptr, colids, values, n_cols, flags, col_offsets)
        except BaseException:
            self.close()
            raise
        DistSpMV._by_handle[self.h.value] = self                  # (only a fully wired handle is reachable from the transport callback)

    def _setup(self, offs, rank, world, rowptr, colids, values, n_cols, flags, col_offsets):
        capi, host, group = self._capi, self._host, self.group
        if col_offsets is None:
            self._phase("create", lambda: capi.check(self.lib.g4s_spmv_dist_create(
                C.byref(self.h), rank, world, offs, int(n_cols), host._ptr(rowptr), host._ptr(colids), host._ptr(values), flags)))
        else:
            assert int(col_offsets[-1]) == int(n_cols)
            coffs = (C.c_int64 * (world + 1))(*[int(v) for v in col_offsets])
            self._phase("create", lambda: capi.check(self.lib.g4s_spmv_dist_create_rect(
                C.byref(self.h), rank, world, offs, coffs, host._ptr(rowptr), host._ptr(colids), host._ptr(values), flags)))
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
            comm = C.c_void_p()
            self._phase("communicator", lambda: capi.check(self.lib.g4s_comm_create(C.byref(comm), world, rank, raw)), collective=True)
            self.comm = comm
            self._phase("connect", lambda: capi.check(self.lib.g4s_spmv_dist_connect_rccl(self.h, self.comm)), collective=True)
        elif world > 1 and not self.allgather:
            self._phase("wire", self._wire_by_torch, collective=True)

    def _agree(self, ok):
        """min over the ranks of a flag (True only if True everywhere); a no-op for one rank."""
        if self.world > 1 and dist.is_initialized():
            on_gpu = dist.get_backend(self.group) == "nccl"
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return int(t.item()) == 1
        return bool(ok)

    def _phase(self, name, fn, collective=False):
        """One set-up phase. G4S_DIST_FAIL (test hook) = "<phase>:<rank>" makes the phase fail on that rank after it ran (its collectives done), and
        "pre:<phase>:<rank>" BEFORE it runs — the real failure mode: a rank that raises in front of a collective its peers are about to enter."""
        err = None
        hook = os.environ.get("G4S_DIST_FAIL")
        fires_before, fires_after = hook == f"pre:{name}:{self.rank}", hook == f"{name}:{self.rank}"
        if fires_before or fires_after:
            os.environ.pop("G4S_DIST_FAIL")                        # one shot: the set-up a caller falls back to is not hit again
        if collective:
            # readiness: nobody enters the phase's collectives unless everybody is about to
            if not self._agree(not fires_before):
                raise RuntimeError("G4S_DIST_FAIL (before the phase)" if fires_before else f"g4s_spmv_dist set-up: another rank could not enter phase '{name}'")
        try:
            if not collective and fires_before:
                raise RuntimeError("G4S_DIST_FAIL (before the phase)")
            fn()
            if fires_after:
                raise RuntimeError("G4S_DIST_FAIL")
        except Exception as e:                                               # noqa: BLE001 — re-raised below, on every rank
            err = e
        if not self._agree(err is None) and err is None:
            err = RuntimeError(f"g4s_spmv_dist set-up: phase '{name}' failed on another rank")
        if err is not None:
            raise err

    # -- set-up over torch.distributed point-to-point: every rank tells every owner which entries it wants
    def _wire_by_torch(self):
        capi = self._capi
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
        self._capi.check(self.lib.g4s_spmv_dist_get_info(self.h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def _buffers(self):
        if self._views is None:
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
        self.exchange_by_torch()
        capi.check(self.lib.g4s_spmv_dist_finish(self.h, host._ptr(y_local), st))
        return y_local

    def exchange_by_torch(self):
        """The transport half of one product when the process group is not RCCL: between g4s_spmv_dist_begin and _finish, carry the library's
        send buffer to the peers and fill its receive buffer through torch.distributed."""
        send, scut, recv, rcut = self._buffers()
        on_dev = _gloo_on_device(send, self.group)
        if self.allgather:                                         # every rank's slot (pad entries) to everybody: one all_gather over the views
            pad = scut[-1]
            if self.world == 1:
                return
            if on_dev:
                torch.cuda.synchronize()
                mine = send[:pad].cpu()
                slots = [torch.empty(pad, dtype=torch.float64) for _ in range(self.world)]
                dist.all_gather(slots, mine, group=self.group)
                recv[:pad * self.world].copy_(torch.cat(slots))
            else:
                dist.all_gather_into_tensor(recv[:pad * self.world], send[:pad].clone(), group=self.group)
            return
        ops = []
        for k in range(self.world):
            if k == self.rank:
                continue
            if scut[k + 1] > scut[k]:
                ops.append(dist.P2POp(dist.isend, send[scut[k]:scut[k + 1]], k, group=self.group))
            if rcut[k + 1] > rcut[k]:
                ops.append(dist.P2POp(dist.irecv, recv[rcut[k]:rcut[k + 1]], k, group=self.group))
        if ops:
            if on_dev:
                torch.cuda.synchronize()
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def allreduce_sum(self, t):
        """Sum over the ranks, in place: the library's RCCL communicator when there is one, torch.distributed otherwise."""
        if self.comm is not None and self.world > 1:
            self._capi.check(self.lib.g4s_comm_allreduce_sum_f64(self.comm, self._host._ptr(t), t.numel(), self._host._stream()))
        else:
            _all_reduce_sum(t, self.group)

    def conj_grad(self, BI_local, F_local, zero_resid_local, acc, steps):
        """Jacobi-CG on the partitioned operator in one C call (g4s_conj_grad_dist: product and dot products on the library's RCCL
        communicator). Needs the RCCL wiring (backend nccl, or loopback). Returns (d0_local, iterations, residual)."""
        capi, host = self._capi, self._host
        if self.comm is None:
            raise RuntimeError("DistSpMV.conj_grad needs the library's RCCL communicator (backend nccl); over gloo use dist_conj_grad(D, …)")
        d0 = torch.empty(self.n_local, dtype=torch.float64, device=F_local.device)
        zr = zero_resid_local if zero_resid_local is not None and zero_resid_local.numel() else None
        cycles, res = C.c_int32(0), C.c_double(0.0)
        capi.check(self.lib.g4s_conj_grad_dist(self.h, self.comm, self.n_local, host._ptr(BI_local), host._ptr(zr) if zr is not None else None,
                                               int(zr.numel()) if zr is not None else 0, host._ptr(F_local), host._ptr(d0), float(acc), int(steps),
                                               C.byref(cycles), C.byref(res), host._stream()))
        return d0, cycles.value, res.value

    def close(self):
        poisoned = False
        if getattr(self, "h", None):
            i = self._capi.DistInfo()
            poisoned = self.lib.g4s_spmv_dist_get_info(self.h, C.byref(i)) == 0 and bool(i.reserved & 4)
            DistSpMV._by_handle.pop(self.h.value, None)
            self.lib.g4s_spmv_dist_destroy(self.h)
            self.h = None
        if getattr(self, "comm", None) is not None:
            if self.comm:                                          # (a poisoned handle has aborted its communicator already: the library remembers that and g4s_comm_destroy is then a no-op — a C caller need not know)
                self.lib.g4s_comm_destroy(self.comm)
            self.comm = None
