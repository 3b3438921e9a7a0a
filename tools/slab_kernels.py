#!/usr/bin/env python3
"""One rank's merged-form product of the 8-way partition of configs[1], 30 times: for rocprofv3 kernel stats (tools/prof_any.sh). Usage: python tools/slab_kernels.py [rank=4] [ranks=8]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from g4s_amd import capi, dist as gdist, host
r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = capi.load()
A = bench.build_matrix("rmat", host, False)
n = A.rows
x = host.synth_vector(7, n)
offs = gdist.row_partition(A.rowptr, W)
r0, r1 = offs[r], offs[r + 1]
rp, ci, va = gdist.slice_rows(A.rowptr, A.colids, A.values, r0, r1)
h = C.c_void_p()
o = (C.c_int64 * (W + 1))(*offs)
torch.cuda.synchronize()
capi.check(lib.g4s_spmv_dist_create(C.byref(h), r, W, o, n, host._ptr(rp), host._ptr(ci), host._ptr(va), capi.DEVICE_POINTERS))
for k in range(W):
    if k != r:
        capi.check(lib.g4s_spmv_dist_set_give(h, k, 0, None, 0))
xl, yl = x[r0:r1].contiguous(), torch.empty(r1 - r0, dtype=torch.float64, device="cuda")
for _ in range(30):
    capi.check(lib.g4s_spmv_dist_begin(h, host._ptr(xl), host._ptr(yl), host._stream()))
    capi.check(lib.g4s_spmv_dist_finish(h, host._ptr(yl), host._stream()))
torch.cuda.synchronize()
info = capi.DistInfo()
capi.check(lib.g4s_spmv_dist_get_info(h, C.byref(info)))
print(f"rank {r}: rows {r1 - r0} nnz {info.nnz_own + info.nnz_rem} n_ref {info.n_ref}")
lib.g4s_spmv_dist_destroy(h)
