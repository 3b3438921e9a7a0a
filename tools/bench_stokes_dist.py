#!/usr/bin/env python3
"""g4s_stokes_uzawa_cg_dist on the Cookbook2-sized mesh (32×32×8) as ONE rank whose operators send half of their slab through RCCL to themselves
(loopback) and whose sums go through ncclAllReduce: the host-side cost of the partitioned loop — speculative (one host wait per outer iteration) against
G4S_STOKES_SYNC=1 (the solve is waited for before the rest of the iteration is enqueued). Usage: python tools/bench_stokes_dist.py [ez] [imp]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from g4s_amd import capi, dist as gdist  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.helpers import assemble_csr, stokes_problem  # noqa: E402
from tests.test_stokes_dist_gpu import div_grad_csr  # noqa: E402

ez = int(sys.argv[1]) if len(sys.argv) > 1 else 8
imp = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
o = oracle_lib.load()
pr = stokes_problem(32, 32, ez, 1)
ien, idmap, nno, neq, nel = pr["ien"], pr["id"], pr["nno"], pr["neq"], len(pr["ien"])
BI = o.element_inverse_diagonal(ien, idmap, pr["K"], neq)
BPI = o.build_diagonal_of_Ahat(ien, idmap, pr["g"], BI)
Kc = assemble_csr(ien, idmap, pr["K"], neq)
Dc, Dtc = div_grad_csr(ien, idmap, pr["g"], neq)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
K = gdist.DistSpMV([0, neq], 0, 1, *[dev(a) for a in Kc], neq, loopback=True)
D = gdist.DistSpMV([0, nel], 0, 1, *[dev(a) for a in Dc], neq, col_offsets=[0, neq])
Dt = gdist.DistSpMV([0, neq], 0, 1, *[dev(a) for a in Dtc], nel, col_offsets=[0, nel])
tr = gdist.rccl_transport(K.comm)
v_res = float(np.linalg.norm(pr["F"]))
prm = capi.StokesParams(imp, 1.0, v_res, 250, 100, 0, 0)
args = [dev(BI), dev(BPI), dev(np.repeat(pr["nmass"], 3)), dev(pr["area"])]
bc, F = dev(pr["bc"]), dev(pr["F"])


def solve():
    V, P = torch.zeros(neq, dtype=torch.float64, device="cuda"), torch.zeros(nel, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _ = gdist.stokes_uzawa_dist(K, D, Dt, tr, *args, pr["volume"], bc, F, V, P, prm)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, res


out = {}
for name, env in (("speculative", None), ("sync", "1"), ("speculative_again", None)):
    if env: os.environ["G4S_STOKES_SYNC"] = env
    else: os.environ.pop("G4S_STOKES_SYNC", None)
    solve()
    ts = []
    for _ in range(5):
        t, res = solve()
        ts.append(t)
    out[name] = {"ms": round(1e3 * min(ts), 3), "outer": res.outer_iterations, "inner": res.inner_iterations,
                 "ms_per_outer": round(1e3 * min(ts) / max(1, res.outer_iterations), 4)}
print(json.dumps({"mesh": [32, 32, ez], "neq": neq, "nel": nel, "imp": imp, "transport": "RCCL, one rank, loopback", **out}))
for h in (K, D, Dt):
    h.close()
