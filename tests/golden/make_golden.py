"""Generates the committed fixtures under tests/golden/. Run from the repo root: python tests/golden/make_golden.py

Provenance of each file (inputs + expected outputs only; no reference source is stored):
  graphprocess_dense.npz  xx, w random (seed 42); result = the REFERENCE's GraphProcess (deepmd/source/op/graph.h:21-32,
                          compiled in place into oracle/_ref by oracle/Makefile) driving the OptMatmul gather. Needs /root/reference.
  spgemm_rmat8.npz        A = R-MAT scale 8 (n=256, 2048 draws, seed 1); C = A·A and y = A·x from the ORACLE restatement
                          (oracle/g4s_oracle.c). The reference has no buildable SpGEMM/SpMV here ("parity unpinned",
                          DESIGN.md); these vectors pin the oracle against regressions and are cross-checked with scipy below.
  element_matvec.npz      2×2×2-element hexahedral mesh, seeded SPD 24×24 blocks; Au from the oracle's restatement of the
                          CitcomS gather (Element_calculations.c:453-471), cross-checked against the assembled matrix.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402
from tests.helpers import hex_mesh, spd_blocks, to_scipy  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
o = oracle_lib.load()

ref = oracle_lib.load_ref()
if ref is not None:
    rng = np.random.default_rng(42)
    xx, w = rng.uniform(-1, 1, (29, 13)), rng.uniform(-1, 1, (13, 7))
    res = np.zeros((29, 7))
    ref.ref_graph_process_dense(29, 13, 7, xx, w, res)
    np.savez(os.path.join(HERE, "graphprocess_dense.npz"), xx=xx, w=w, result=res)
else:
    print("oracle/_ref missing: graphprocess_dense.npz not regenerated")

n = 256
arpt, acol, aval = o.rmat_csr(1, 8, n, 2048)
x = o.vector(7, n)
crpt, ccol, cval = o.spgemm((arpt, acol, aval), (arpt, acol, aval), n)
y = o.spmv(arpt, acol, aval, x)
A = to_scipy(arpt, acol, aval, n, n)
assert np.allclose(to_scipy(crpt, ccol, cval, n, n).toarray(), (A @ A).toarray(), rtol=1e-13, atol=1e-13)
assert np.allclose(y, A @ x, rtol=1e-13, atol=1e-13)
np.savez(os.path.join(HERE, "spgemm_rmat8.npz"), n=n, arpt=arpt, acol=acol, aval=aval, x=x, crpt=crpt, ccol=ccol, cval=cval, y=y)

ien, idmap, nno, neq = hex_mesh(2, 2, 2)
K = spd_blocks(len(ien), 24, 3)
u = np.random.default_rng(4).uniform(-1, 1, neq)
Au = o.element_matvec(ien, idmap, K, u, neq)
np.savez(os.path.join(HERE, "element_matvec.npz"), ien=ien, id=idmap, elt_k=K, u=u, Au=Au, neq=neq, nno=nno)
print("golden fixtures written")
