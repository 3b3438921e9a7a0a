#!/usr/bin/env python3
"""The Cookbook2-sized stiffness operator in its three forms — element-by-element (g4s_elem_op_apply), node-assembled blocks
(g4s_node_op_apply) and assembled CSR (g4s_spmv) — one mat-vec each, same vector. Usage: python tools/bench_fe_ops.py [ez]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from g4s_amd import capi, host
from tests import oracle_lib
from tests.helpers import assemble_csr, hex_mesh, hex_node_map, spd_blocks
ez = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib, o = capi.load(), oracle_lib.load()
ien, idmap, nno, neq = hex_mesh(32, 32, ez)
K = spd_blocks(len(ien), 24, 1)
Kd = torch.from_numpy(K).cuda()
h = C.c_void_p()
capi.check(lib.g4s_elem_op_create(C.byref(h), len(ien), 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq, Kd.data_ptr()))
nm, max_eqn = hex_node_map(32, 32, ez, idmap)
ks = o.construct_node_ks(ien, idmap, nno, neq, nm, K, np.ones((nno, 3)))
hn = C.c_void_p()
capi.check(lib.g4s_node_op_create(C.byref(hn), nno, neq, max_eqn, np.ascontiguousarray(nm).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, ks[0].ctypes.data, ks[1].ctypes.data, ks[2].ctypes.data))
rp, ci, va = assemble_csr(ien, idmap, K, neq)
A = host.CSR.from_host(rp, ci, va, neq, neq)
S = host.CSR.from_host(rp, ci, va, neq, neq, spmv_flags=capi.SPMV_STREAM)     # the same matrix on the row-streaming CSR kernel
S.handle
u = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, neq)).cuda()
outs = [torch.empty_like(u) for _ in range(4)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
calls = {"element-by-element": lambda y: lib.g4s_elem_op_apply(h, u.data_ptr(), y.data_ptr(), st),
         "node-assembled blocks": lambda y: lib.g4s_node_op_apply(hn, u.data_ptr(), y.data_ptr(), None, 0, st),
         "assembled CSR (g4s_spmv)": lambda y: lib.g4s_spmv(A.handle, u.data_ptr(), y.data_ptr(), 1.0, 0.0, st),
         "assembled CSR, G4S_SPMV_STREAM": lambda y: lib.g4s_spmv(S.handle, u.data_ptr(), y.data_ptr(), 1.0, 0.0, st)}
bytes_ = {"element-by-element": len(ien) * 576 * 8 + 16 * neq, "node-assembled blocks": nno * 27 * 76 + 16 * neq, "assembled CSR (g4s_spmv)": 12 * len(ci) + 4 * (neq + 1) + 16 * neq,
          "assembled CSR, G4S_SPMV_STREAM": 12 * len(ci) + 4 * (neq + 1) + 16 * neq}
res = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (name, f), y in zip(calls.items(), outs):
    for _ in range(20):
        f(y)
    e0.record()
    for _ in range(200):
        f(y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    res[name] = {"us": round(us, 2), "bytes": bytes_[name], "GBps": round(bytes_[name] / us / 1e3, 1)}
ref = outs[0].cpu().numpy()
for name, y in zip(calls, outs):
    res[name]["max_rel_diff_vs_elements"] = float(np.max(np.abs(y.cpu().numpy() - ref)) / np.max(np.abs(ref)))
print(json.dumps({"mesh": f"32x32x{ez}", "neq": neq, "nnz_assembled": int(len(ci)), "g4s_spmv_path": A.info()["spmv_path"], "matvec": res}))
