#!/usr/bin/env python3
"""BASELINE configs[4] shape: one incompressible Stokes solve (solve_Ahat_p_fhat_CG, citcoms/lib/Stokes_flow_Incomp.c:188-452) on the
Cookbook2-sized mesh (32×32×8 hexahedra, neq 29403, 8192 pressure unknowns) with synthetic seeded operators — device-resident
g4s_stokes_uzawa_cg vs the oracle's restatement on one host thread. Usage: python tools/bench_stokes.py [ez] [imp] [csr]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from g4s_amd import capi  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.helpers import assemble_csr, stokes_problem  # noqa: E402
from g4s_amd import host  # noqa: E402

ez = int(sys.argv[1]) if len(sys.argv) > 1 else 8
imp = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4          # control.accuracy of Cookbook2 (Instructions.c:658)
lib, o = capi.load(), oracle_lib.load()
pr = stokes_problem(32, 32, ez, 1)
ien, idmap, nno, neq, nel = pr["ien"], pr["id"], pr["nno"], pr["neq"], len(pr["ien"])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
Kd, gd, nmd, ard, bcd, Fd = dev(pr["K"]), dev(pr["g"]), dev(pr["nmass"]), dev(pr["area"]), dev(pr["bc"]), dev(pr["F"])
h = C.c_void_p()
capi.check(lib.g4s_elem_op_create(C.byref(h), nel, 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq, Kd.data_ptr()))
BId = torch.empty(neq, dtype=torch.float64, device="cuda")
BPId = torch.empty(nel, dtype=torch.float64, device="cuda")
capi.check(lib.g4s_elem_op_inverse_diagonal(h, BId.data_ptr(), None))
capi.check(lib.g4s_elem_op_pressure_preconditioner(h, gd.data_ptr(), BId.data_ptr(), BPId.data_ptr(), None))
v_res = float(np.linalg.norm(pr["F"]))
use_csr = len(sys.argv) > 3 and sys.argv[3] == "csr"          # velocity operator: element-by-element (default) or the assembled CSR matrix
Acsr = host.CSR.from_host(*assemble_csr(ien, idmap, pr["K"], neq), neq, neq) if use_csr else None
KCSR = Acsr.handle if use_csr else None
prm = capi.StokesParams(imp, 1.0, v_res, 250, 100, 0, 0)
res = capi.StokesResult()


def solve():
    Vd, Pd = torch.zeros(neq, dtype=torch.float64, device="cuda"), torch.zeros(nel, dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_stokes_uzawa_cg(h, KCSR, gd.data_ptr(), BId.data_ptr(), BPId.data_ptr(), nmd.data_ptr(), ard.data_ptr(), pr["volume"], bcd.data_ptr(),
                                       len(pr["bc"]), Fd.data_ptr(), Vd.data_ptr(), Pd.data_ptr(), C.byref(prm), C.byref(res), None, 0, None))
    return Vd, Pd


solve()
torch.cuda.synchronize()
times = []
for _ in range(9):                                   # median: the loop is host-launch-bound and a shared box adds 2–8× outliers
    t0 = time.perf_counter()
    Vd, Pd = solve()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
gpu_ms = sorted(times)[len(times) // 2]
t0 = time.perf_counter()
Vo, Po, cnt, inc, hist, inner = o.solve_Ahat_p_fhat_CG(ien, idmap, nno, neq, pr["K"], pr["g"], BId.cpu().numpy(), BPId.cpu().numpy(), pr["nmass"], pr["area"],
                                                       pr["volume"], pr["bc"], pr["F"], np.zeros(neq), np.zeros(nel), imp, 1.0, v_res, 250, 100)
cpu_ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"workload": f"Stokes solve, 32x32x{ez} elements: neq {neq}, pressure unknowns {nel}, accuracy {imp}, stiffness operator: " + ("assembled CSR through g4s_spmv" if use_csr else "element-by-element"),
                  "outer_iterations": res.outer_iterations, "outer_iterations_oracle": cnt, "inner_cg_iterations": res.inner_iterations,
                  "inner_cg_iterations_oracle": inner, "gpu_ms_median_of_9": round(gpu_ms, 3), "gpu_ms_min": round(min(times), 3), "gpu_us_per_inner_iteration": round(gpu_ms * 1e3 / max(res.inner_iterations, 1), 2),
                  "incompressibility": res.incompressibility, "cpu_oracle_ms_1thread": round(cpu_ms, 1), "speedup_vs_1thread": round(cpu_ms / gpu_ms, 1),
                  "max_rel_diff_V": float(np.max(np.abs(Vd.cpu().numpy() - Vo)) / np.max(np.abs(Vo))),
                  "max_rel_diff_P": float(np.max(np.abs(Pd.cpu().numpy() - Po)) / np.max(np.abs(Po)))}))
lib.g4s_elem_op_destroy(h)
