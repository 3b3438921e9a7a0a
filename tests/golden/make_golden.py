"""Generates the committed fixtures under tests/golden/. Run from the repo root: python tests/golden/make_golden.py

Provenance of each file (inputs + expected outputs only; no reference source is stored):
  graphprocess_dense.npz  xx, w random (seed 42); result = the REFERENCE's GraphProcess (deepmd/source/op/graph.h:21-32,
                          compiled in place into oracle/_ref by oracle/Makefile) driving the OptMatmul gather. Needs /root/reference.
  spgemm_rmat8.npz        A = R-MAT scale 8 (n=256, 2048 draws, seed 1); C = A·A and y = A·x from the ORACLE restatement
                          (oracle/g4s_oracle.c). The reference has no buildable SpGEMM/SpMV here ("parity unpinned",
                          DESIGN.md); these vectors pin the oracle against regressions and are cross-checked with scipy below.
  mkl_spgemm.npz          Five SpGEMM cases (4×4 tridiagonal, R-MAT scale 8 and 10 A·A, 60×50·50×70 with empty rows, a power-law matrix with
                          a dense row) and A·x written as an n×1 SpGEMM; C from Intel oneMKL itself, called through ctypes with the reference's
                          call sequence (mm/inc/mkl_mult.h:40-111 → oracle/mkl_ref.py). This is the library the reference's shipped binary
                          (mm/src/mkl_spgemm.cpp) computes with; needs /opt/conda/lib/libmkl_rt.so (2021.4 here; the reference's Makefile names 2022.2).
  mkl_dense.npz           The dense comparison drivers of the reference (mv/mv.c:6-27 dsymv/dtrmv/dspmv/dgemv, mm/src/cblas_dxxmm.c:57-111
                          dsymm/dtrmm/dgemm) at dim 48 and 100 on seeded inputs, results from the same oneMKL.
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

# ---- oneMKL through the reference's call sequence (skipped where the MKL runtime is absent; the committed files then stay as they are)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mkl_ref  # noqa: E402
if mkl_ref.available():
    import scipy.sparse as sp
    from tests.helpers import power_law_csr, random_csr
    out = {"mkl_version": np.array(mkl_ref.version())}
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(4, 4), format="csr")
    T.sort_indices()
    cases = {"tri4": ((T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data.astype(np.float64)),) * 2 + ((4, 4, 4),),
             "rmat8": ((arpt, acol, aval),) * 2 + ((n, n, n),),
             "rmat10": (o.rmat_csr(3, 10, 1024, 8192),) * 2 + ((1024, 1024, 1024),),
             "rect": (random_csr(60, 50, 0.1, 3, empty_rows=(0, 7, 59)), random_csr(50, 70, 0.08, 4, empty_rows=(3,)), (60, 50, 70)),
             "plaw": (power_law_csr(300, 300, 11, 200),) * 2 + ((300, 300, 300),),
             "spmv_rmat8": ((arpt, acol, aval), (np.arange(n + 1, dtype=np.int32), np.zeros(n, np.int32), x), (n, n, 1))}
    for name, (Am, Bm, (M, K, N)) in cases.items():
        c = mkl_ref.mkl_spgemm(Am, Bm, M, K, N)
        co = o.spgemm(Am, Bm, N)
        assert np.array_equal(c[0], co[0]) and np.array_equal(c[1], co[1]), name        # the oracle's index arrays ARE MKL's
        assert np.allclose(c[2], co[2], rtol=1e-12, atol=1e-13), name
        for tag, arrs in (("a", Am), ("b", Bm), ("c", c)):
            for nm, arr in zip(("rpt", "col", "val"), arrs):
                out[f"{name}_{tag}{nm}"] = arr
        out[f"{name}_mkn"] = np.array([M, K, N])
    np.savez_compressed(os.path.join(HERE, "mkl_spgemm.npz"), **out)
    dense = {"mkl_version": np.array(mkl_ref.version())}
    for dim in (48, 100):
        rng = np.random.default_rng(100 + dim)
        Ad, Bd, xd = rng.uniform(-1, 1, dim * dim), rng.uniform(-1, 1, dim * dim), rng.uniform(-1, 1, dim)
        AP = rng.uniform(-1, 1, dim * (dim + 1) // 2)
        dense.update({f"A{dim}": Ad, f"B{dim}": Bd, f"x{dim}": xd, f"AP{dim}": AP,
                      f"dgemm{dim}": mkl_ref.dgemm(Ad, Bd, dim), f"dsymm{dim}": mkl_ref.dsymm(Ad, Bd, dim), f"dtrmm{dim}": mkl_ref.dtrmm(Ad, Bd, dim),
                      f"dgemv{dim}": mkl_ref.dgemv(Ad, xd), f"dsymv{dim}": mkl_ref.dsymv(Ad, xd), f"dtrmv{dim}": mkl_ref.dtrmv(Ad, xd),
                      f"dspmv{dim}": mkl_ref.dspmv(AP, xd)})
    np.savez_compressed(os.path.join(HERE, "mkl_dense.npz"), **dense)
else:
    print("oneMKL runtime not found: mkl_spgemm.npz / mkl_dense.npz not regenerated")

ien, idmap, nno, neq = hex_mesh(2, 2, 2)
K = spd_blocks(len(ien), 24, 3)
u = np.random.default_rng(4).uniform(-1, 1, neq)
Au = o.element_matvec(ien, idmap, K, u, neq)
np.savez(os.path.join(HERE, "element_matvec.npz"), ien=ien, id=idmap, elt_k=K, u=u, Au=Au, neq=neq, nno=nno)
print("golden fixtures written")
