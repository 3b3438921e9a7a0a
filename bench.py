#!/usr/bin/env python3
"""bench.py — the headline metric of BASELINE.json: fp64 CSR SpMV GEdges/s (+ achieved HBM GB/s against the roofline).

A step = one pass of the hot path over the whole matrix: y = A·x on BASELINE configs[1], the 10M×10M R-MAT power-law
matrix with ≈1e8 nonzeros (synthetic, generated in HBM by the kernels of g4s_amd/csrc/synth.hip). With N>1 ranks the rows
are split by equal work (the rule of mm/inc/BIN.h:101-122), every rank owns its slab of x and y, and a step is
{exchange x over RCCL/xGMI, local SpMV} — total work is fixed, so scaling is "strong". Inputs are resident in HBM before
the timed region. Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rmat|banded|lap5|lap7] [--exchange auto|dist|allgather]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (≈6.3 TB/s achievable on a float4 copy)

WORKLOADS = {
    # name: (description, builder kwargs)
    "rmat": "fp64 CSR SpMV, 10M x 10M R-MAT (0.57,0.19,0.19,0.05) power-law, 1e8 edge draws deduplicated (BASELINE configs[1])",
    "banded": "fp64 CSR SpMV, 10M x 10M banded, half-bandwidth 5 (north_star: 'power-law and banded')",
    "lap5": "fp64 CSR SpMV, 1M x 1M 5-point Laplacian (BASELINE configs[0])",
    "lap7": "fp64 CSR SpMV, 7-point 3-D Laplacian 431^3 = 80M rows (BASELINE configs[3])",
}


def build_matrix(name, host, small):
    if name == "rmat":
        n, scale, edges = (10_000_000, 24, 100_000_000) if not small else (200_000, 18, 2_000_000)
        return host.rmat_csr(n, scale, edges, 20240521)
    if name == "banded":
        return host.banded_csr(10_000_000 if not small else 200_000, 5, 20240521)
    if name == "lap5":
        s = 1000 if not small else 300
        return host.laplacian_csr(5, s, s)
    if name == "lap7":
        s = 431 if not small else 60
        return host.laplacian_csr(7, s, s, s)
    raise SystemExit(f"unknown workload {name}")


def host_threads():
    """Threads for the CPU baseline: the affinity mask, clipped by the cgroup CPU quota and by the GPU box's CPU share
    (16 cores per GPU slot; oversubscribing the 256 visible cores measured SLOWER than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("G4S_CPU_THREADS", "16"))))


def cpu_baseline(A, x, budget_s=18.0):
    """The oracle's SpMV (oracle/g4s_oracle.c, 'port') timed on this box's host cores on the same matrix: one thread (BASELINE configs[0]),
    14 threads (the reference's hard-coded value, mm/src/mkl_spgemm.cpp:61) and the box's CPU share. Reported, not a target. Also returns
    the oracle's y so that the bench line can carry the GPU result's worst relative error."""
    import numpy as np
    from tests import oracle_lib
    o = oracle_lib.load()
    rp, ci, va = A.to_host()
    xh = x.cpu().numpy()
    y = np.zeros(A.rows)
    cores = host_threads()
    counts = sorted({1, min(14, cores), cores})
    out = {}
    for threads in counts:
        o.spmv_mt(rp, ci, va, xh, y, threads)          # warm-up pass (page-in)
        t0, passes = time.perf_counter(), 0
        while True:
            o.spmv_mt(rp, ci, va, xh, y, threads)
            passes += 1
            el = time.perf_counter() - t0
            if el > budget_s / len(counts) or passes >= 400:
                break
        out[threads] = (A.nnz * passes / el / 1e9, passes, el)
    v, passes, el = out[cores]
    _, asum = o.spmv_ld(rp, ci, va, xh)                  # Σ|a_ij·x_j| per row: the scale of the 1e-10 tolerance
    base = {"value": round(v, 4), "unit": "GEdges/s", "cores": cores, "kind": "port",
            "sample": f"whole matrix (nnz={A.nnz}), {passes} passes in {el:.1f}s, OpenMP rows split by equal nnz; "
                      + "; ".join(f"{t} thread(s): {out[t][0]:.4f} GEdges/s ({out[t][1]} passes)" for t in counts),
            "by_threads": {str(t): round(out[t][0], 4) for t in counts},
            "single_thread_value": round(out[1][0], 4)}
    base["reference_library"] = reference_library_baseline(rp, ci, va, xh, A.rows, A.cols, A.nnz, y, asum)
    return base, y, asum


def reference_library_baseline(rp, ci, va, xh, rows, cols, nnz, y_oracle, asum, threads=14):
    """The reference's own sparse CPU path — its mm/ call sequence on oneMKL (mm/inc/mkl_mult.h:40-111: create ×2, mkl_sparse_spmm, convert,
    order, export; 14 threads as mm/src/mkl_spgemm.cpp:61) — computing the same product: x as a cols×1 CSR matrix, so C = A·X holds y on the
    rows of A that hold entries. (Its mv/ benchmark is dense BLAS-2 on dim² doubles and cannot hold this matrix) Run where
    the oneMKL runtime exists (it ships in this image; no headers, no reference source compiled — oracle/mkl_ref.py); reported, not a target."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
    try:
        import mkl_ref
        if not mkl_ref.available():
            return {"skipped": "libmkl_rt.so not found on this box"}
        mkl_ref.load(threading="gnu")
        B = (np.arange(cols + 1, dtype=np.int32), np.zeros(cols, dtype=np.int32), xh)
        tm, runs = {}, []
        for i in range(3):                                        # 1 warm-up + mean of 2, the protocol of tools/bench_spgemm.py --mkl
            crp, cci, cva = mkl_ref.mkl_spgemm((rp, ci, va), B, rows, cols, 1, timings=tm, threads=threads)
            if i:
                runs.append(dict(tm))
        mean = {k: sum(r[k] for r in runs) / len(runs) for k in runs[0]}
        y = np.zeros(rows)
        y[np.diff(crp) > 0] = cva
        return {"value": round(nnz / (mean["total"] * 1e-3) / 1e9, 4), "unit": "GEdges/s", "cores": threads, "kind": "reference",
                "what": "mm/inc/mkl_mult.h:40-111 call sequence on oneMKL " + mkl_ref.version()[35:52].strip() + ", A·x as a cols×1 product",
                "stage_ms": {k: round(v, 1) for k, v in mean.items()},
                "spmm_stage_only": round(nnz / (mean["spmm"] * 1e-3) / 1e9, 4),
                "agrees_with_oracle": bool(np.all(np.abs(y - y_oracle) <= 1e-10 * asum + 1e-300))}
    except Exception as e:                                         # noqa: BLE001 — a baseline, never fatal for the bench line
        return {"error": f"{type(e).__name__}: {e}"}


def config_secondary(kind, world, rank, steps, warmup, small, host, gdist, dist, torch):
    """A secondary workload beside the headline number, same ranks, same timing protocol; reported under `also` / `also_banded`, never as `value`.
    kind "lap7": BASELINE configs[3], the 431^3 7-point Laplacian (80M rows), rows split evenly over the ranks, every rank generating only its slab.
    kind "banded": north_star's banded matrix (10M x 10M, half-bandwidth 5), rows split evenly. N = 1: one g4s_spmv handle; N > 1: the library's
    distributed product (halo planes / halo rows travel, overlapped with the own-column part)."""
    if kind == "lap7":
        s = 431 if not small else 48
        n = s ** 3
        offs = [(n * k) // world for k in range(world + 1)]
        r0, r1 = offs[rank], offs[rank + 1]
        A = host.laplacian_csr(7, s, s, s, r0=r0, r1=r1)
    else:
        n = 10_000_000 if not small else 200_000
        offs = [(n * k) // world for k in range(world + 1)]
        r0, r1 = offs[rank], offs[rank + 1]
        full = host.banded_csr(n, 5, 20240521)
        if world > 1:
            rp, ci, va = gdist.slice_rows(full.rowptr, full.colids, full.values, r0, r1)
            A = host.CSR(rp, ci, va, r1 - r0, n)
            del full
            torch.cuda.empty_cache()
        else:
            A = full
    x_local = host.synth_vector(7, r1 - r0, i0=r0)
    y_local = torch.empty(r1 - r0, dtype=torch.float64, device="cuda")
    nnz_local = A.nnz
    if world > 1:
        D = gdist.DistSpMV(offs, rank, world, A.rowptr, A.colids, A.values, n)
        path = D.info()["own_path"]
        A = None
        torch.cuda.empty_cache()

        def step():
            D(x_local, y_local)
    else:
        D = None
        path = A.info()["spmv_path"]

        def step():
            A.spmv(x_local, y_local)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    nz = torch.tensor([float(nnz_local)], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nz)
    el, nnz = float(el.item()), int(nz.item())
    if D is not None:
        D.close()
    alg = 12 * nnz + 4 * (n + 1) + 16 * n
    extra = {}
    if world == 1:
        inf = A.info()
        extra["plan_bytes"] = inf["plan_bytes"]
        if inf["spmv_path"] == 3:
            # the index-free kernel does not move the CSR arrays the algorithmic model counts: its own bytes are the diagonals + row masks
            # (= plan_bytes) + x + y, and THAT is what crosses the HBM interface (ADVICE r2 / VERDICT r2: print both)
            own = inf["plan_bytes"] + 8 * n + 8 * n
            extra.update({"bytes_model": "csr-algorithmic (12*nnz + 4*(rows+1) + 8*rows + 8*cols); the diagonal kernel itself moves kernel_own_bytes",
                          "kernel_own_bytes": own, "kernel_own_gbs": round(own * steps / el / 1e9, 1), "kernel_own_frac_of_8TBs": round(own * steps / el / 1e9 / 8000.0, 4)})
    return {**extra, "workload": WORKLOADS[kind] + (" [--small size]" if small else ""), "n_gpus": world, "rows": n, "nnz": nnz, "steps": steps, "ms_per_step": round(el / steps * 1e3, 5),
            "value": round(nnz * steps / el / 1e9, 3), "unit": "GEdges/s", "spmv_path": {0: "stream", 1: "blocked", 3: "diagonal (index-free)", 4: "block-row"}.get(path, str(path)),
            "hbm_gbs_algorithmic_whole_job": round(alg * steps / el / 1e9, 1),
            # the fraction of n_gpus x 8 TB/s on the bytes the kernel MOVES: the CSR-algorithmic bytes on the CSR paths; on the index-free diagonal path those
            # bytes are not moved (its own fraction is kernel_own_frac_of_8TBs above) and the figure is labelled for what it is: a CSR-equivalent rate
            ("csr_equivalent_frac_of_n_gpus_x_8TBs" if path == 3 else "frac_of_n_gpus_x_8TBs"): round(alg * steps / el / 1e9 / (8000.0 * world), 4)}


def config2_spgemm(small, host, capi, torch, runs=10, mkl_threads=14):
    """BASELINE configs[2] beside the headline: C = A·A on the R-MAT scale-21 matrix (edge factor 3: the largest whose nnz(C) fits the reference's
    int32 crpt), ONE g4s_spgemm_csr_i32_f64 call per run, in the reference's protocol — 1 warm-up + mean of 10 runs, GFLOPS = 2·flop/t
    (mm/src/mkl_spgemm.cpp:60-85). Roofline on SURVEY §8d's byte model. Next to it the reference's own call sequence (mm/inc/mkl_mult.h:40-111)
    on oneMKL on this box's host, same matrix, one run (≈ 20 s), where the runtime exists. Reported under `also_spgemm`, never as `value`."""
    import numpy as np
    scale, ef = (21, 3.0) if not small else (15, 3.0)
    n = 1 << scale
    A = host.rmat_csr(n, scale, int(ef * n), 20240522)
    A.values.abs_()
    flop = host.get_flop(A, A)
    times, stages, nnzc = [], [], 0
    for i in range(runs + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = host.HashSpGEMM(A, A)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        nnzc = c.nnz
        if i:
            times.append(dt)
            stages.append(c.timings)
        if i == runs:
            # size-independent properties of the result the timed call produced (the parity tests compare whole arrays at this size)
            checksum = float(c.values.sum().item())
            # every term is positive: Σ C = Σ_k (column sums of A)_k · (row sums of A)_k
            col_sum = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, A.colids.long(), A.values)
            row_sum = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, torch.repeat_interleave(torch.arange(n, device="cuda"), torch.diff(A.rowptr).long()), A.values)
            want = float((col_sum * row_sum).sum().item())
        del c
    mean = sum(times) / len(times)
    model = 12 * A.nnz + 4 * n + 12 * flop + 12 * nnzc + 8 * n
    out = {"metric": "fp64 SpGEMM A*A GFLOPS (2*flop/t)", "value": round(2 * flop / (mean * 1e-3) / 1e9, 2), "unit": "GFLOPS", "ms_per_call": round(mean, 3), "min_ms": round(min(times), 3),
           "runs": runs, "warmup": 1, "protocol": "mm/src/mkl_spgemm.cpp:60-85 (1 warm-up + mean of 10)", "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"R-MAT scale {scale}, edge factor {ef}, C = A*A, one g4s_spgemm_csr_i32_f64 call (device pointers, sorted output)" + (" [--small]" if small else ""),
                      "rows": n, "nnz_A": A.nnz, "flop": flop, "nnz_C": nnzc, "compression": round(flop / max(nnzc, 1), 3)},
           "stage_ms": {k: round(sum(st[k] for st in stages) / len(stages), 3) for k in stages[0]},
           "roofline": {"bound": "hbm", "model": "12*nnz(A) + 4*rows + 12*flop + 12*nnz(C) + 8*rows (SURVEY 8d)", "model_bytes": model,
                        "achieved": round(model / (mean * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(model / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "compulsory_bytes": 12 * (2 * A.nnz + nnzc)},
           "self_check": {"sum_of_C": checksum, "expected_from_A": want, "rel_err": abs(checksum - want) / want, "ok": bool(abs(checksum - want) <= 1e-9 * want)}}
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mkl_ref
        if not mkl_ref.available():
            out["cpu_baseline"] = {"skipped": "libmkl_rt.so not found on this box"}
        else:
            mkl_ref.load(threading="gnu")
            rp, ci, va = A.to_host()
            tm = {}
            crp, cci, cva = mkl_ref.mkl_spgemm((rp, ci, va), (rp, ci, va), n, n, n, timings=tm, threads=mkl_threads)
            out["cpu_baseline"] = {"value": round(2 * flop / (tm["total"] * 1e-3) / 1e9, 3), "unit": "GFLOPS", "cores": mkl_threads, "kind": "reference",
                                   "sample": f"the same matrix, ONE run without warm-up of the call sequence mm/inc/mkl_mult.h:40-111 on oneMKL {mkl_ref.version()[35:52].strip()}; stage ms: "
                                             + ", ".join(f"{k} {v:.0f}" for k, v in tm.items()), "nnz_C_equal_to_gpu": bool(len(cci) == nnzc)}
            del crp, cci, cva
    except Exception as e:                                         # noqa: BLE001 — a baseline, never fatal
        out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as fresh child processes (this process has not
    touched the GPU yet and never will), relay rank 0's JSON line, return the children's exit code."""
    import subprocess
    # --standalone: the launcher binds a free port itself and keeps it (a port picked here and handed over could be taken in between)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)               # (0.16 s of products: one 5 ms hiccup of the box moves the line by 3 %, not by 13 % as with 100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="rmat", choices=sorted(WORKLOADS))
    ap.add_argument("--exchange", default="auto", choices=["auto", "dist", "allgather"],
                    help="auto = dist: the C-ABI multi-GPU product (g4s_spmv_dist_*) with the packed exchange (RCCL send/recv of the referenced x entries, overlapped with "
                         "the own-column product); allgather: the same product with ONE ncclAllGather of the whole vector (also the fallback when the packed wiring fails)")
    ap.add_argument("--small", action="store_true", help="reduced sizes for plumbing checks (not a valid benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary configs[3] (7-point Laplacian 431^3) measurement reported under `also`")
    ap.add_argument("--spgemm-runs", type=int, default=10, help="timed runs of the configs[2] SpGEMM call under `also_spgemm` (after 1 warm-up)")
    ap.add_argument("--no-nt", action="store_true", help="plain loads for the matrix stream (A/B against nontemporal)")
    ap.add_argument("--path", default="auto", choices=["auto", "stream", "blocked"],
                    help="SpMV path: auto = the library picks per matrix (propagation-blocked without gather locality, index-free diagonal form for "
                         "stencil / banded matrices, row-streaming CSR kernel otherwise); stream / blocked force one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + G4S_BENCH_SAME_DEVICE=1 rehearses the N>1 path with all ranks on cuda:0 (plumbing check only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))              # before anything touches the GPU

    import torch
    import torch.distributed as dist
    from g4s_amd import capi, host
    from g4s_amd import dist as gdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    if os.environ.get("G4S_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    # the library's device code is loaded now, not inside the first g4s_csr_create (g4s_warm_up, include/g4s.h: HIP loads a translation unit's code object at its
    # first launch — 12 to 25 ms that are the process's, not the plan's; plan.build_ms below is then the build itself)
    capi.check(capi.load().g4s_warm_up())
    # ---- inputs, resident in HBM
    A_full = build_matrix(args.workload, host, args.small)
    n_rows, n_cols, nnz_total = A_full.rows, A_full.cols, A_full.nnz
    # (one part needs no partition pass: g4s_row_partition copies the row pointers to the host — 40 MB through the runtime's staging buffers — and the first create
    # behind that copy measured 16 ms slower every other run, 36 against 20 ms on one box)
    offs = gdist.row_partition(A_full.rowptr, world) if world > 1 else [0, n_rows]
    r0, r1 = offs[rank], offs[rank + 1]
    flags = (capi.SPMV_NO_NT if args.no_nt else 0) | {"auto": 0, "stream": capi.SPMV_STREAM, "blocked": capi.SPMV_BLOCKED}[args.path]
    mode = args.exchange if args.exchange != "auto" else "dist"
    D = None
    plan_ms = plan_ms_repeated = None
    if world > 1:
        rp, ci, va = gdist.slice_rows(A_full.rowptr, A_full.colids, A_full.values, r0, r1)
        del A_full
        torch.cuda.empty_cache()
        x_local = host.synth_vector(7, r1 - r0, i0=r0)
        y_local = torch.empty(r1 - r0, dtype=torch.float64, device="cuda")
        # the library's multi-GPU product (g4s_spmv_dist_*, csrc/dist.hip): own / remote column split; exchange "dist" = packed ncclSend/ncclRecv of the
        # referenced x entries on the library's own RCCL communicator, own-column product overlapped with it; "allgather" = ONE in-place ncclAllGather of
        # the padded slabs (north_star's "RCCL all-gather of the dense vector"). DistSpMV agrees after every set-up phase whether it succeeded on ALL
        # ranks and raises everywhere at the same point if not; the ranks then fall back — together — to the all-gather exchange, which needs no wiring.
        err = ""
        try:
            D = gdist.DistSpMV(offs, rank, world, rp, ci, va, n_cols, spmv_flags=flags, exchange="packed" if mode == "dist" else "allgather")
        except Exception as e:                                  # noqa: BLE001 — reported below, decided collectively
            D, err = None, f"{type(e).__name__}: {e}"
        okf = torch.tensor([1 if D is not None else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0:
            if err:
                print(f"[bench rank {rank}] g4s_spmv_dist set-up failed ({err}); every rank falls back to --exchange allgather", file=sys.stderr, flush=True)
            if D is not None:
                D.close()
            if mode == "allgather":
                raise SystemExit("the all-gather exchange could not be set up either")
            mode = "allgather"
            D = gdist.DistSpMV(offs, rank, world, rp, ci, va, n_cols, spmv_flags=flags, exchange="allgather")
        dinfo = D.info()
        recv_bytes = dinfo["recv_bytes"]
        merged = bool(dinfo["reserved"] & 1)
        info = {"spmv_path": dinfo["rem_path"] if merged else dinfo["own_path"], "dist_form": "merged (one product on the gathered x)" if merged else "own + remote columns",
                "algorithmic_bytes": 12 * int(rp[-1].item()) + 4 * (r1 - r0 + 1) + 8 * (r1 - r0) + 8 * (r1 - r0 + dinfo["n_ref"]),
                "rows": r1 - r0, "nnz": int(rp[-1].item()), "plan_bytes": None}

        def product():
            D(x_local, y_local)
        step = product
    else:
        A = host.CSR(A_full.rowptr, A_full.colids, A_full.values, n_rows, n_cols, spmv_flags=flags)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        A.handle                                                   # g4s_csr_create: the plan is built here, once per matrix
        torch.cuda.synchronize()
        plan_ms = (time.perf_counter() - tp0) * 1e3
        # the same create once more on a second handle: the library's block cache serves it (the first create of a process that did not call g4s_warm_up also
        # pays for loading the plan kernels: 25 to 44 ms by box for a build that takes 18.6)
        A2 = host.CSR(A_full.rowptr, A_full.colids, A_full.values, n_rows, n_cols, spmv_flags=flags)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        A2.handle
        torch.cuda.synchronize()
        plan_ms_repeated = (time.perf_counter() - tp0) * 1e3
        del A2
        x_full = host.synth_vector(7, n_cols)
        x_local = x_full
        y_local = torch.empty(n_rows, dtype=torch.float64, device="cuda")
        info = A.info()
        recv_bytes = 0

        def product():
            A.spmv(x_full, y_local)
        step = product

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- roofline of the SpMV launch: HIP events on the launch stream around back-to-back launches. One launch of the hot path is
    # spmv_csr_adaptive_kernel (+ the few-µs long-row fixup) on the streaming path, or pb_prepare_kernel + pb_producer_kernel + pb_consumer_kernel on
    # the blocked path (their durations add up to kernel_ms; profiles/ holds the rocprofv3 split). At N=1 the timed region above
    # IS that; at N>1 it is re-measured without the exchange, outside the timed region.
    if world == 1:
        kernel_ms = ev0.elapsed_time(ev1) / args.steps
    else:
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(args.steps):
            product()
        k1.record()
        torch.cuda.synchronize()
        kernel_ms = k0.elapsed_time(k1) / args.steps
    achieved = info["algorithmic_bytes"] / (kernel_ms * 1e-3) / 1e9

    # HBM bytes per launch come from PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, tools/prof_pmc.sh), which cannot run inside this process:
    # the number is read from the stored profile of the same workload and SpMV path, and the line says so (traffic_source).
    # A stored number is only as good as the kernels it was taken on: the profile records the SHA-256 of the SpMV kernel sources (tools/kernel_hash.py:
    # spmv.hip, spmv_pb.hip, spmv_bcsr.hip and their headers, common.hpp) and a number whose hash differs from the sources of the loaded build is refused.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == args.workload and tj.get("n_gpus") == world and not args.small and tj.get("spmv_path", info["spmv_path"]) == info["spmv_path"]:
                # the hash the LOADED library was built from (g4s_build_info), not the tree's: a stale build, or an A/B variant loaded through G4S_LIB, is not
                # what the stored profile was taken on
                binfo = dict(kv.split("=", 1) for kv in capi.load().g4s_build_info().decode().split(";") if "=" in kv)
                now = binfo.get("spmv_kernel_sources_sha256", "unknown") + ("" if not binfo.get("variant") else "+variant:" + binfo["variant"])
                if tj.get("kernel_sources_sha256") == now:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = f"stored PMC profile profiles/{tj.get('source', 'traffic_latest.json')} of the kernel sources the loaded library was built from (sha256 {now[:12]}…; not measured by this run)"
                else:
                    traffic_source = (f"REFUSED: profiles/traffic_latest.json was taken on kernel sources {str(tj.get('kernel_sources_sha256'))[:12]}…, this build is {now[:12]}… "
                                      "— re-take it with tools/r04_profile.sh")
        except Exception as e:                                      # noqa: BLE001
            traffic, traffic_source = None, f"stored profile unreadable: {type(e).__name__}"

    result = {
        "metric": "fp64 SpMV GEdges/s",
        "value": round(nnz_total * args.steps / elapsed / 1e9, 4),
        "unit": "GEdges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload] + (" [--small: NOT the benchmark size]" if args.small else ""),
                   "rows": n_rows, "cols": n_cols, "nnz": nnz_total, "index": "int32",
                   "partition": f"1-D rows by equal nnz+rows over {world} rank(s)",
                   "exchange": ("none (single GPU)" if world == 1 else
                                (f"g4s_spmv_dist_* ({info.get('dist_form')}; " + ("packed ncclSend/ncclRecv of the referenced x entries" if mode == "dist" else "one in-place ncclAllGather of the padded slabs")
                                 + f"{'' if merged else ', overlapped with the own-column product'}), {recv_bytes} B received by rank 0 per step")),
                   "matrix_loads": "plain" if args.no_nt else "nontemporal",
                   "spmv_path": {0: "stream", 1: "blocked", 3: "diagonal (index-free)", 4: "block-row"}[info["spmv_path"]],
                   "reproducible": ("yes: fixed summation order, no atomics" if info["spmv_path"] in (0, 3, 4) else
                                    "no: fp64 sums meet in LDS / global atomics, last bits may differ run to run (inside the 1e-10 tolerance)"),
                   **({"backend": "gloo (rehearsal, not a valid multi-GPU number)"} if args.backend != "nccl" else {})},
        "hbm_gbs_algorithmic_whole_job": round((12 * nnz_total + 4 * (n_rows + 1) + 8 * n_rows + 8 * n_cols) * args.steps / elapsed / 1e9, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": {0: "spmv_csr_adaptive_kernel", 1: "pb_prepare_kernel+pb_producer_kernel+pb_consumer_kernel (propagation-blocked SpMV)",
                                3: "spmv_dia_kernel", 4: "spmv_bcsr_kernel"}[info["spmv_path"]], "kernel_ms": round(kernel_ms, 5),
                     "algorithmic_bytes_per_launch": info["algorithmic_bytes"],
                     "launch_rows": info["rows"], "launch_nnz": info["nnz"],
                     **({"kernel_ms_includes_exchange": True, "note": "N > 1: kernel_ms is one distributed product on rank 0 — pack, ncclSend/ncclRecv and both local products — not a single kernel"} if world > 1 else {})},
        "plan": {"plan_bytes": info.get("plan_bytes"), "build_ms": None if plan_ms is None else round(plan_ms, 2),
                 "build_ms_repeated": None if plan_ms_repeated is None else round(plan_ms_repeated, 2),
                 "build_in_products": None if plan_ms is None else round(plan_ms / (elapsed / args.steps * 1e3), 1)},
    }
    if info["spmv_path"] == 3 and world == 1 and info.get("plan_bytes"):
        # the index-free diagonal kernel does not move the CSR arrays the algorithmic model counts: its own bytes beside them
        own = info["plan_bytes"] + 8 * n_rows + 8 * n_cols
        result["roofline"].update({"bytes_model": "csr-algorithmic; the diagonal kernel itself moves kernel_own_bytes (diagonals + row masks + x + y)",
                                   "kernel_own_bytes": own, "kernel_own_gbs": round(own / (kernel_ms * 1e-3) / 1e9, 1),
                                   "kernel_own_frac": round(own / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, y_cpu, asum = cpu_baseline(A, x_full)
        result["cpu_baseline"] = base
        # the bench checks itself: the y the timed launches produced against the oracle's, relative to Σ|a_ij·x_j| of the row
        import numpy as np
        err = np.abs(y_local.cpu().numpy() - y_cpu) / np.maximum(asum, 1e-300)
        result["max_rel_err"] = float(err.max())
        result["parity"] = {"against": "oracle/g4s_oracle.c CSR SpMV on the same matrix and x", "tolerance": 1e-10, "ok": bool(err.max() <= 1e-10)}
    elif rank == 0:
        result["cpu_baseline"] = None
    if not args.no_also and args.workload == "rmat":
        # the multi-GPU config of BASELINE.json (configs[3]) in the same run, same ranks — R-MAT has no column locality and its strong
        # scaling is exchange-bound by construction; the stencil shows what the row partition does when only halos travel
        torch.cuda.empty_cache()
        try:                                                       # a failure here (raised on every rank alike, see DistSpMV) must not cost the headline line
            also = config_secondary("lap7", world, rank, max(10, args.steps // 2), min(args.warmup, 5), args.small, host, gdist, dist, torch)
        except Exception as e:                                     # noqa: BLE001
            also = {"error": f"{type(e).__name__}: {e}"}
        result["also"] = also
        torch.cuda.empty_cache()
        try:                                                       # north_star: "GEdges/s on synthetic power-law AND banded matrices"
            also_b = config_secondary("banded", world, rank, max(10, args.steps // 2), min(args.warmup, 5), args.small, host, gdist, dist, torch)
        except Exception as e:                                     # noqa: BLE001
            also_b = {"error": f"{type(e).__name__}: {e}"}
        result["also_banded"] = also_b
    if not args.no_also and args.workload == "rmat" and world == 1:
        # BASELINE configs[2] in the driver's own line ("driver-timed SpGEMM"); guarded like `also`
        try:
            del A, x_full, y_local
        except Exception:                                          # noqa: BLE001
            pass
        torch.cuda.empty_cache()
        try:
            result["also_spgemm"] = config2_spgemm(args.small, host, capi, torch, runs=args.spgemm_runs)
        except Exception as e:                                     # noqa: BLE001
            result["also_spgemm"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        if D is not None:
            D.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
