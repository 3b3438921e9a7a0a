#!/usr/bin/env python3
"""Does a hipGraph help the Uzawa outer iteration? Measured, on the Cookbook2-sized problem (32x32x8 elements, neq 29 403, 8 192 pressure unknowns).

One outer iteration of solve_Ahat_p_fhat_CG (citcoms/lib/Stokes_flow_Incomp.c:296-405) is: z = BPI∘r, <r,z>, s = z + δ·s, grad(s), a velocity
solve of N CG iterations (direction, K·p, p·Ap, update: 4 launches each), div(u1), <s, div u1>, the P / V / r updates, div(V), four norms — about
20 + 4·N launches of 3–11 µs and, in g4s_stokes_uzawa_cg, two host read-backs (the CG's state, the iteration's nine scalars). Here the SAME
launches are issued through the C-ABI's capture-safe entry points (g4s_elem_op_grad_p / _div_u, the g4s_cg_* step API on the assembled K through
g4s_spmv; the outer loop's axpys and dots by torch element-wise kernels of the same sizes) with NO host read inside, N fixed to what the eager
solver needed, and timed three ways: eager launches, the same sequence replayed from a hipGraph, and the production call per outer iteration.
usage: python tools/uzawa_graph_probe.py"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from g4s_amd import capi, host  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.helpers import assemble_csr, stokes_problem  # noqa: E402

lib, o = capi.load(), oracle_lib.load()
pr = stokes_problem(32, 32, 8, 1)
ien, idmap, nno, neq, nel = pr["ien"], pr["id"], pr["nno"], pr["neq"], len(pr["ien"])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Kd, gd, nmd, ard, bcd, Fd = dev(pr["K"]), dev(pr["g"]), dev(pr["nmass"]), dev(pr["area"]), dev(pr["bc"]), dev(pr["F"])
h = C.c_void_p()
capi.check(lib.g4s_elem_op_create(C.byref(h), nel, 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq, Kd.data_ptr()))
BId, BPId = torch.empty(neq, dtype=torch.float64, device="cuda"), torch.empty(nel, dtype=torch.float64, device="cuda")
capi.check(lib.g4s_elem_op_inverse_diagonal(h, BId.data_ptr(), None))
capi.check(lib.g4s_elem_op_pressure_preconditioner(h, gd.data_ptr(), BId.data_ptr(), BPId.data_ptr(), None))
A = host.CSR.from_host(*assemble_csr(ien, idmap, pr["K"], neq), neq, neq)
A.handle
v_res = float(np.linalg.norm(pr["F"]))
imp = 1e-4
prm, res = capi.StokesParams(imp, 1.0, v_res, 250, 100, 0, 0), capi.StokesResult()


def production():
    V, P = torch.zeros(neq, dtype=torch.float64, device="cuda"), torch.zeros(nel, dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_stokes_uzawa_cg(h, A.handle, gd.data_ptr(), BId.data_ptr(), BPId.data_ptr(), nmd.data_ptr(), ard.data_ptr(), pr["volume"], bcd.data_ptr(),
                                       len(pr["bc"]), Fd.data_ptr(), V.data_ptr(), P.data_ptr(), C.byref(prm), C.byref(res), None, 0, None))


production()
torch.cuda.synchronize()
ts = []
for _ in range(9):
    t0 = time.perf_counter()
    production()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
prod_ms = sorted(ts)[4]
outer, inner = res.outer_iterations, res.inner_iterations
N = int(round(inner / (outer + 1)))                               # CG iterations per velocity solve (the initial residual solve included)

# ---- one outer iteration, no host reads, N fixed
ws = C.c_void_p()
capi.check(lib.g4s_cg_ws_create(C.byref(ws), neq))
f64 = lambda n: torch.zeros(n, dtype=torch.float64, device="cuda")
r1, z1, s1, s2, Fp, Pv, r2 = (torch.rand(nel, dtype=torch.float64, device="cuda") for _ in range(7))
tmp, u1, V = f64(neq), f64(neq), torch.rand(neq, dtype=torch.float64, device="cuda")
nm3 = nmd.repeat_interleave(3)
acc = imp * v_res
p_ptr, Ap_ptr = C.c_void_p(), C.c_void_p()


def outer_iteration():
    st = host._stream()
    torch.mul(BPId, r1, out=z1)
    r1z1 = torch.dot(r1, z1)
    torch.add(z1, s1, alpha=0.37, out=s2)
    capi.check(lib.g4s_elem_op_grad_p(h, gd.data_ptr(), s2.data_ptr(), tmp.data_ptr(), bcd.data_ptr(), len(pr["bc"]), st))
    capi.check(lib.g4s_cg_begin(ws, tmp.data_ptr(), BId.data_ptr(), u1.data_ptr(), bcd.data_ptr(), len(pr["bc"]), st))
    for _ in range(N):
        capi.check(lib.g4s_cg_direction(ws, 250, acc, st))
        capi.check(lib.g4s_cg_buffers(ws, C.byref(p_ptr), C.byref(Ap_ptr), None))
        capi.check(lib.g4s_spmv(A.handle, p_ptr, Ap_ptr, 1.0, 0.0, st))
        capi.check(lib.g4s_cg_reduce_pAp(ws, st))
        capi.check(lib.g4s_cg_update(ws, BId.data_ptr(), u1.data_ptr(), st))
    capi.check(lib.g4s_elem_op_div_u(h, gd.data_ptr(), u1.data_ptr(), Fp.data_ptr(), st))
    alpha = r1z1 / torch.dot(s2, Fp)
    torch.addcmul(r1, Fp, -alpha.expand_as(Fp), out=r2)
    Pv.addcmul_(s2, alpha.expand_as(s2))
    V.addcmul_(u1, -alpha.expand_as(u1))
    capi.check(lib.g4s_elem_op_div_u(h, gd.data_ptr(), V.data_ptr(), z1.data_ptr(), st))
    n1 = torch.dot(V * V, nm3); n2 = torch.dot(u1 * u1, nm3); n3 = torch.dot(Pv * Pv, ard); n4 = torch.dot(s2 * s2, ard); n5 = torch.dot(z1 * z1, 1.0 / ard)
    return n1 + n2 + n3 + n4 + n5


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3 / reps)
    return sorted(out)[2]


eager_ms = timed(outer_iteration)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    outer_iteration()                                              # warm the allocator on the capture stream
    with torch.cuda.graph(g, stream=side):
        outer_iteration()
torch.cuda.current_stream().wait_stream(side)
graph_ms = timed(g.replay)
print(json.dumps({"workload": f"Cookbook2-sized Stokes problem: neq {neq}, pressure unknowns {nel}, accuracy {imp}, assembled K through g4s_spmv",
                  "production_solve_ms_median_of_9": round(prod_ms, 3), "outer_iterations": outer, "inner_cg_iterations": inner,
                  "production_ms_per_outer_iteration": round(prod_ms / (outer + 1), 4), "cg_iterations_per_velocity_solve": N,
                  "launches_per_outer_iteration": 4 * N + 30,
                  "one_outer_iteration_no_host_reads_eager_ms": round(eager_ms, 4), "the_same_from_a_hipgraph_ms": round(graph_ms, 4),
                  "graph_over_eager": round(graph_ms / eager_ms, 3),
                  "note": "eager and graph issue identical kernels; the production number also holds the two host read-backs per outer iteration and the initial residual solve"}))
lib.g4s_cg_ws_destroy(ws)
lib.g4s_elem_op_destroy(h)
