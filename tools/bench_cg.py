#!/usr/bin/env python3
"""BASELINE configs[4] shape: CitcomS Cookbook2-sized regional Stokes operator (32×32×8 hexahedra, nno 9801, neq 29403, nel 8192,
synthetic SPD 24×24 element blocks), Jacobi-CG to accuracy 1e-4·|F| with at most 250 iterations (Instructions.c:658,674) —
device-resident g4s_conj_grad vs the oracle's restatement of conj_grad on one host thread. Reports time per CG iteration and per mat-vec."""
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
from tests.helpers import hex_mesh, hex_node_map, spd_blocks  # noqa: E402

ez = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib, o = capi.load(), oracle_lib.load()
ien, idmap, nno, neq = hex_mesh(32, 32, ez)
K = spd_blocks(len(ien), 24, 1)
rng = np.random.default_rng(1)
bc = np.array(sorted(set(idmap[rng.choice(nno, nno // 9, replace=False)].ravel().tolist())), np.int32)
F = rng.uniform(-1, 1, neq)
F[bc] = 0.0
acc = 1e-4 * np.linalg.norm(F)
Kd = torch.from_numpy(K).cuda()
h = C.c_void_p()
capi.check(lib.g4s_elem_op_create(C.byref(h), len(ien), 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq, Kd.data_ptr()))
BId = torch.empty(neq, dtype=torch.float64, device="cuda")
capi.check(lib.g4s_elem_op_inverse_diagonal(h, BId.data_ptr(), None))
Fd, bcd = torch.from_numpy(F).cuda(), torch.from_numpy(bc).cuda()
d0 = torch.empty(neq, dtype=torch.float64, device="cuda")
cyc, res = C.c_int32(250), C.c_double()
for _ in range(2):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad(h, None, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad(h, None, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / reps * 1e3
# mat-vec alone
u = torch.from_numpy(F).cuda()
Au = torch.empty_like(u)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(20):
    lib.g4s_elem_op_apply(h, u.data_ptr(), Au.data_ptr(), st)
e0.record()
for _ in range(200):
    lib.g4s_elem_op_apply(h, u.data_ptr(), Au.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
mv_us = e0.elapsed_time(e1) / 200 * 1e3
BI = BId.cpu().numpy()
t0 = time.perf_counter()
d_or, cyc_or, res_or, _ = o.conj_grad_elem(ien, idmap, K, neq, BI, bc, F, acc, 250)
cpu_ms = (time.perf_counter() - t0) * 1e3
alg = len(ien) * 576 * 8 + 16 * neq
print(json.dumps({"workload": f"Cookbook2-sized element operator 32x32x{ez}: nel {len(ien)}, neq {neq}", "cg_iterations": cyc.value, "cg_iterations_oracle": cyc_or,
                  "gpu_cg_ms": round(gpu_ms, 3), "gpu_us_per_iteration": round(gpu_ms * 1e3 / cyc.value, 2), "gpu_matvec_us": round(mv_us, 2),
                  "matvec_algorithmic_bytes": alg, "matvec_GBps": round(alg / (mv_us * 1e-6) / 1e9, 1), "matvec_frac_of_8TBps": round(alg / (mv_us * 1e-6) / 8e12, 4),
                  "cpu_oracle_cg_ms_1thread": round(cpu_ms, 1), "max_rel_diff": float(np.max(np.abs(d0.cpu().numpy() - d_or)) / np.max(np.abs(d_or)))}))
# steady state: the same operator driven to 1e-13·|F| (tens of iterations) — per-iteration time without the per-solve fixed cost
acc2 = 1e-13 * np.linalg.norm(F)
for _ in range(2):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad(h, None, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc2, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad(h, None, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc2, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
long_ms = (time.perf_counter() - t0) / reps * 1e3
d_or2, cyc_or2, _, _ = o.conj_grad_elem(ien, idmap, K, neq, BI, bc, F, acc2, 250)
print(json.dumps({"workload": "same operator, accuracy 1e-13*|F|", "cg_iterations": cyc.value, "cg_iterations_oracle": cyc_or2, "gpu_cg_ms": round(long_ms, 3),
                  "gpu_us_per_iteration": round(long_ms * 1e3 / cyc.value, 2), "max_rel_diff": float(np.max(np.abs(d0.cpu().numpy() - d_or2)) / np.max(np.abs(d_or2)))}))
# the same operator in CitcomS's node-assembled form (Node_map / Eqn_k → per-node 3×3 neighbour blocks)
nm, max_eqn = hex_node_map(32, 32, ez, idmap)
bcw = np.ones((nno, 3))
bcw.reshape(-1)[np.searchsorted(idmap.ravel(), bc)] = 0.0
ks = o.construct_node_ks(ien, idmap, nno, neq, nm, K, bcw)
hn = C.c_void_p()
capi.check(lib.g4s_node_op_create(C.byref(hn), nno, neq, max_eqn, np.ascontiguousarray(nm).ctypes.data, np.ascontiguousarray(idmap).ctypes.data,
                                  ks[0].ctypes.data, ks[1].ctypes.data, ks[2].ctypes.data))
for _ in range(20):
    lib.g4s_node_op_apply(hn, u.data_ptr(), Au.data_ptr(), None, 0, st)
e0.record()
for _ in range(200):
    lib.g4s_node_op_apply(hn, u.data_ptr(), Au.data_ptr(), None, 0, st)
e1.record()
torch.cuda.synchronize()
nmv_us = e0.elapsed_time(e1) / 200 * 1e3
for _ in range(2):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad_node(hn, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc2, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    cyc.value = 250
    capi.check(lib.g4s_conj_grad_node(hn, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc2, C.byref(cyc), C.byref(res), None))
torch.cuda.synchronize()
node_ms = (time.perf_counter() - t0) / reps * 1e3
nalg = nno * 27 * (72 + 4) + 16 * neq
print(json.dumps({"workload": "same operator, node-assembled form, accuracy 1e-13*|F|", "cg_iterations": cyc.value, "gpu_cg_ms": round(node_ms, 3),
                  "gpu_us_per_iteration": round(node_ms * 1e3 / cyc.value, 2), "gpu_matvec_us": round(nmv_us, 2), "matvec_algorithmic_bytes": nalg,
                  "matvec_GBps": round(nalg / (nmv_us * 1e-6) / 1e9, 1), "max_rel_diff_vs_elem_oracle": float(np.max(np.abs(d0.cpu().numpy() - d_or2)) / np.max(np.abs(d_or2)))}))
lib.g4s_node_op_destroy(hn)
lib.g4s_elem_op_destroy(h)
