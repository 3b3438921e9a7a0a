"""CPU suite: the oracle against independent implementations (scipy.sparse), closed forms, the reference's own
GraphProcess (oracle/_ref, compiled in place from /root/reference) and the committed golden fixtures."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests import oracle_lib
from tests.helpers import hex_mesh, power_law_csr, random_csr, spd_blocks, to_scipy

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("rows,cols,density,seed", [(1, 1, 1.0, 0), (17, 23, 0.3, 1), (300, 300, 0.05, 2), (1000, 777, 0.01, 3)])
def test_spmv_matches_scipy(oracle, rows, cols, density, seed):
    rp, ci, va = random_csr(rows, cols, density, seed, empty_rows=[0] if rows > 1 else [])
    x = np.random.default_rng(seed).uniform(-1, 1, cols)
    y = oracle.spmv(rp, ci, va, x)
    ref = to_scipy(rp, ci, va, rows, cols) @ x
    yld, asum = oracle.spmv_ld(rp, ci, va, x)
    assert np.all(np.abs(y - ref) <= 1e-13 * (asum + 1e-300) + 1e-300)
    assert np.all(np.abs(y - yld) <= 1e-13 * (asum + 1e-300) + 1e-300)


def test_spmv_alpha_beta_and_beta_zero_ignores_y(oracle):
    rp, ci, va = random_csr(50, 40, 0.2, 5)
    x = np.linspace(-1, 1, 40)
    y0 = np.full(50, np.nan)
    y = oracle.spmv(rp, ci, va, x, y0, alpha=2.0, beta=0.0)     # NaN in y must not propagate when beta == 0
    assert np.all(np.isfinite(y))
    base = oracle.spmv(rp, ci, va, x)
    yin = np.arange(50, dtype=np.float64)
    y2 = oracle.spmv(rp, ci, va, x, yin, alpha=-0.5, beta=3.0)
    assert np.allclose(y2, -0.5 * base + 3.0 * yin, rtol=1e-15, atol=1e-15)


def test_spmv_laplacian_closed_form(oracle):
    # 5-point Laplacian applied to the constant vector: 0 in the interior, boundary deficit elsewhere
    nx, ny = 13, 9
    rp, ci, va = oracle.laplacian5(nx, ny)
    assert rp[-1] == 5 * nx * ny - 2 * nx - 2 * ny
    y = oracle.spmv(rp, ci, va, np.ones(nx * ny)).reshape(ny, nx)
    assert np.all(y[1:-1, 1:-1] == 0.0)
    assert y[0, 0] == 2.0 and y[0, 1] == 1.0
    rp7, ci7, va7 = oracle.laplacian7(4, 5, 6)
    y7 = oracle.spmv(rp7, ci7, va7, np.ones(120)).reshape(6, 5, 4)
    assert np.all(y7[1:-1, 1:-1, 1:-1] == 0.0) and y7[0, 0, 0] == 3.0


def test_spmv_mt_matches_single_thread(oracle):
    rp, ci, va = power_law_csr(4000, 4000, 11, 600)
    x = np.random.default_rng(0).uniform(-1, 1, 4000)
    y1 = oracle.spmv(rp, ci, va, x)
    y4 = np.zeros(4000)
    used = oracle.spmv_mt(rp, ci, va, x, y4, 4)
    assert used >= 1 and np.array_equal(y1, y4)   # same per-row order => bit-identical


# ------------------------------------------------------------------ SpGEMM pieces
def test_flop_bin_rows_offset(oracle):
    arp, aci, ava = random_csr(60, 50, 0.1, 7, empty_rows=[3, 4])
    brp, bci, bva = random_csr(50, 70, 0.1, 8, empty_rows=[0])
    tot, rf = oracle.flop(arp, aci, brp, per_row=True)
    blen = np.diff(brp)
    expect = np.array([blen[aci[arp[i]:arp[i + 1]]].sum() for i in range(60)])
    assert np.array_equal(rf, expect) and tot == expect.sum()
    b = oracle.bin_id(70, rf)
    for i in range(60):                       # BIN.h:158-177
        nz, bi = min(int(rf[i]), 70), int(b[i])
        if nz == 0:
            assert bi == 0
        else:
            assert (8 << (bi - 1)) >= nz and (bi == 1 or (8 << (bi - 2)) < nz)
    off = oracle.rows_offset(rf, 4)           # BIN.h:101-122
    assert off[0] == 0 and off[-1] == 60 and np.all(np.diff(off) >= 0)
    ps = np.concatenate([[0], np.cumsum(rf)])
    avg = (tot + 3) // 4
    for t in range(3):
        assert off[t + 1] == np.searchsorted(ps, avg * (t + 1), side="left")


def test_bin_id_table_can_be_full(oracle):
    # row_nz >= cols is clipped to cols: the table may be 100 % full
    assert list(oracle.bin_id(8, np.array([0, 1, 8, 9, 1000]))) == [0, 1, 1, 1, 1]
    assert list(oracle.bin_id(100, np.array([8, 9, 16, 17, 64, 65]))) == [1, 2, 2, 3, 4, 5]


@pytest.mark.parametrize("M,K,N,da,db,seed", [(4, 4, 4, 0.6, 0.6, 0), (40, 30, 50, 0.15, 0.2, 1), (200, 200, 200, 0.03, 0.03, 2),
                                              (64, 8, 64, 0.9, 0.9, 3)])
def test_spgemm_matches_scipy(oracle, M, K, N, da, db, seed):
    A = random_csr(M, K, da, seed, empty_rows=[1] if M > 2 else [])
    B = random_csr(K, N, db, seed + 100)
    crpt, ccol, cval = oracle.spgemm(A, B, N, sort_output=True)
    ref = (to_scipy(*A, M, K) @ to_scipy(*B, K, N)).tocsr()
    ref.sort_indices()
    # structural product (scipy drops nothing unless exact cancellation; compare structure via pattern product)
    pat = (to_scipy(A[0], A[1], np.ones_like(A[2]), M, K) @ to_scipy(B[0], B[1], np.ones_like(B[2]), K, N)).tocsr()
    pat.sort_indices()
    assert np.array_equal(crpt, pat.indptr) and np.array_equal(ccol, pat.indices)
    dense = ref.toarray()
    got = to_scipy(crpt, ccol, cval, M, N).toarray()
    assert np.allclose(got, dense, rtol=1e-13, atol=1e-13)


def test_spgemm_tridiagonal_known_answer(oracle):
    # 4×4 tridiag(−1,2,−1) squared: nnz 14, rows [5 −4 1 / −4 6 −4 1 / 1 −4 6 −4 / 1 −4 5], flop 26 (SURVEY.md §8c probe)
    A = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(4, 4), format="csr")
    a = (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))
    crpt, ccol, cval = oracle.spgemm(a, a, 4)
    assert crpt[-1] == 14 and oracle.flop(a[0], a[1], a[0]) == 26
    assert np.array_equal(to_scipy(crpt, ccol, cval, 4, 4).toarray(),
                          np.array([[5, -4, 1, 0], [-4, 6, -4, 1], [1, -4, 6, -4], [0, 1, -4, 5]], float))


def test_spgemm_unsorted_output_is_a_permutation(oracle):
    A = random_csr(30, 30, 0.2, 21)
    c1 = oracle.spgemm(A, A, 30, sort_output=True)
    c0 = oracle.spgemm(A, A, 30, sort_output=False)
    assert np.array_equal(c0[0], c1[0])
    for i in range(30):
        s, e = c1[0][i], c1[0][i + 1]
        o = np.argsort(c0[1][s:e], kind="stable")
        assert np.array_equal(c0[1][s:e][o], c1[1][s:e]) and np.array_equal(c0[2][s:e][o], c1[2][s:e])


def test_spgemm_as_spmv(oracle):
    # A·X with X an n×1 CSR reproduces A·x with the SpGEMM multiply-add (SURVEY.md §8c cross-check 4)
    rp, ci, va = random_csr(80, 60, 0.1, 31)
    x = np.random.default_rng(1).uniform(0.5, 1.5, 60)
    X = (np.arange(61, dtype=np.int32), np.zeros(60, np.int32), x)
    crpt, ccol, cval = oracle.spgemm((rp, ci, va), X, 1)
    y = oracle.spmv(rp, ci, va, x)
    nonempty = np.diff(rp) > 0
    assert np.array_equal(np.diff(crpt) > 0, nonempty) and np.array_equal(cval, y[nonempty])


# ------------------------------------------------------------------ MatrixMarket reader (CSR.h:485-669)
def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_mtx_general_pattern_symmetric_skew(tmp_path, oracle):
    p = _write(tmp_path, "g.mtx", "%%MatrixMarket matrix coordinate real general\n% c\n3 4 4\n3 1 1.5\n1 2 -2\n1 1 4\n2 4 7e-1\n")
    r, c, rp, ci, va = oracle.mtx_read(p)
    assert (r, c) == (3, 4) and list(rp) == [0, 2, 3, 4] and list(ci) == [0, 1, 3, 0] and list(va) == [4, -2, 0.7, 1.5]
    p = _write(tmp_path, "p.mtx", "%%MatrixMarket matrix coordinate pattern general\n2 2 2\n2 2\n1 1\n")
    _, _, rp, ci, va = oracle.mtx_read(p)
    assert list(va) == [1.0, 1.0] and list(ci) == [0, 1]
    # symmetric: 7 stored entries, 3 off-diagonal → nnz 10 (SURVEY.md §8c probe)
    p = _write(tmp_path, "s.mtx", "%%MatrixMarket matrix coordinate real symmetric\n4 4 7\n1 1 2\n2 1 -1\n2 2 2\n3 2 -1\n3 3 2\n4 3 -1\n4 4 2\n")
    r, c, rp, ci, va = oracle.mtx_read(p)
    assert rp[-1] == 10
    assert np.array_equal(to_scipy(rp, ci, va, 4, 4).toarray(), sp.diags([-1, 2, -1], [-1, 0, 1], shape=(4, 4)).toarray())
    p = _write(tmp_path, "k.mtx", "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 5\n3 2 -3\n")
    _, _, rp, ci, va = oracle.mtx_read(p)
    d = to_scipy(rp, ci, va, 3, 3).toarray()
    assert np.array_equal(d, -d.T) and d[1, 0] == 5 and d[0, 1] == -5
    p = _write(tmp_path, "c.mtx", "%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 3.0 9.0\n")
    assert list(oracle.mtx_read(p)[4]) == [3.0]           # real part kept (CSR.h:544-554)


def test_mtx_duplicates_kept_and_errors(tmp_path, oracle):
    p = _write(tmp_path, "d.mtx", "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n1 1 2\n2 2 3\n")
    _, _, rp, ci, va = oracle.mtx_read(p)
    assert list(rp) == [0, 2, 3] and list(ci) == [0, 0, 1] and sorted(va[:2]) == [1, 2]   # not merged
    for bad in ["%%MatrixMarket matrix array real general\n1 1\n1\n", "%%MatrixMarket matrix coordinate real hermitian\n1 1 1\n1 1 1\n",
                "%%MatrixMarket vector coordinate real general\n1 1 1\n1 1 1\n", "garbage\n"]:
        with pytest.raises(ValueError):
            oracle.mtx_read(_write(tmp_path, "bad.mtx", bad))


# ------------------------------------------------------------------ graph interface
def test_graph_process_pinned_to_reference_build(oracle):
    """oracle_spmm_dense / oracle_dense_rows_times_matrix against the REFERENCE's GraphProcess (graph.h:21-32) built in
    place into oracle/_ref. Skipped only where neither /root/reference nor a prebuilt oracle/_ref exists."""
    ref = oracle_lib.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libref_graph.so not built (no /root/reference here)")
    rng = np.random.default_rng(3)
    M, N, K = 37, 19, 11
    xx, w = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, K))
    got = oracle.dense_rows_times_matrix(xx, w)
    want = np.zeros((M, K))
    ref.ref_graph_process_dense(M, N, K, xx, w, want)
    assert np.array_equal(got, want)                       # same k order, multiply then add: bit-identical
    assert np.allclose(got, xx @ w, rtol=1e-14, atol=1e-14)

    # driver loop order: a race-free gather (writes only result[vi*deg+nb]) and an apply that reads what gather wrote
    deg = 5
    rows = (C.POINTER(C.c_double) * M)(*[xx[i].ctypes.data_as(C.POINTER(C.c_double)) for i in range(M)])
    states = np.ascontiguousarray(w.ravel())

    @oracle_lib.FUN_GATHER
    def gather(vi, nb, ew, st, res):
        res[vi * (deg + 1) + nb] = ew[vi][nb] * st[nb] + vi

    @oracle_lib.FUN_APPLY
    def apply(vi, ew, st, res):
        res[vi * (deg + 1) + deg] = sum(res[vi * (deg + 1) + j] for j in range(deg))

    r_or, r_ref = np.zeros(M * (deg + 1)), np.zeros(M * (deg + 1))
    oracle.lib.oracle_spmm_dense(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, r_or.ctypes.data, gather, apply, None, 1)
    ref.ref_graph_process_cb(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, r_ref.ctypes.data, gather, apply)
    assert np.array_equal(r_or, r_ref)


def test_element_matvec_equals_assembled_matrix(oracle):
    # element-by-element K·u (Element_calculations.c:453-471) == assembled sparse matrix times u
    ien, idmap, nno, neq = hex_mesh(3, 2, 2)
    K = spd_blocks(len(ien), 24, 5)
    u = np.random.default_rng(2).uniform(-1, 1, neq)
    Au = oracle.element_matvec(ien, idmap, K, u, neq)
    Au1 = oracle.element_matvec(ien, idmap, K, u, neq, base=1)
    assert np.array_equal(Au, Au1)
    A = np.zeros((neq, neq))
    for e in range(len(ien)):
        eq = idmap[ien[e]].ravel()
        A[np.ix_(eq, eq)] += K[e].reshape(24, 24)
    assert np.allclose(Au, A @ u, rtol=1e-12, atol=1e-12)
    assert np.allclose(A, A.T)


def test_sym_quadratic_form_equals_double_loop(oracle):
    # gather1/apply1 (RedlichKwongMFTP.cpp:960-970) == Σ_ij x_i x_j a_ij (calculateAB :1036-1059), b = Σ x_i b_i
    rng = np.random.default_rng(9)
    m = 23
    a, x, b = rng.uniform(0, 1, m * m), rng.uniform(0, 1, m), rng.uniform(0, 1, m)
    r = oracle.sym_quadratic_form(m, 1, a, x, b)
    full = sum(x[i] * x[j] * a[i + m * j] for i in range(m) for j in range(m))
    assert abs(r[0] - full) <= 1e-13 * abs(full) and abs(r[1] - x @ b) <= 1e-13 * abs(x @ b)
    a2 = rng.uniform(0, 1, 2 * m * m)
    r2 = oracle.sym_quadratic_form(m, 2, a2, x)
    f0 = sum(x[i] * x[j] * a2[2 * (i + m * j)] for i in range(m) for j in range(m))
    f1 = sum(x[i] * x[j] * a2[2 * (i + m * j) + 1] for i in range(m) for j in range(m))
    assert abs(r2[0] - f0) <= 1e-13 * abs(f0) and abs(r2[1] - f1) <= 1e-13 * abs(f1)


# ------------------------------------------------------------------ generators
def test_generators_are_deterministic_and_in_range(oracle):
    k1 = oracle.rmat_keys(20240521, 10, 1000, 0, 5000)
    k2 = np.concatenate([oracle.rmat_keys(20240521, 10, 1000, 0, 1234), oracle.rmat_keys(20240521, 10, 1000, 1234, 5000 - 1234)])
    assert np.array_equal(k1, k2)                              # counter-based: chunking does not matter
    assert k1.min() >= 0 and (k1 // 1000).max() < 1000 and (k1 % 1000).max() < 1000
    rows = k1 // 1000
    assert (rows < 500).mean() > 0.7                           # a+b = 0.76 of the mass in the upper half
    rp, ci, va = oracle.rmat_csr(20240521, 10, 1000, 5000)
    assert rp[-1] == len(np.unique(k1)) and np.all(np.abs(va) <= 1.0)
    for r in range(1000):
        assert np.all(np.diff(ci[rp[r]:rp[r + 1]]) > 0)        # sorted, duplicate-free
    rpb, cib, vab = oracle.banded(50, 3, 1)
    assert rpb[-1] == 50 * 7 - 3 * 4 and cib[0] == 0 and cib[rpb[1] - 1] == 3


# ------------------------------------------------------------------ golden fixtures
def test_golden_fixtures(oracle):
    g = np.load(os.path.join(GOLD, "spgemm_rmat8.npz"))
    crpt, ccol, cval = oracle.spgemm((g["arpt"], g["acol"], g["aval"]), (g["arpt"], g["acol"], g["aval"]), int(g["n"]))
    assert np.array_equal(crpt, g["crpt"]) and np.array_equal(ccol, g["ccol"]) and np.array_equal(cval, g["cval"])
    assert np.array_equal(oracle.spmv(g["arpt"], g["acol"], g["aval"], g["x"]), g["y"])
    d = np.load(os.path.join(GOLD, "graphprocess_dense.npz"))           # produced by the reference's GraphProcess
    assert np.array_equal(oracle.dense_rows_times_matrix(d["xx"], d["w"]), d["result"])
    e = np.load(os.path.join(GOLD, "element_matvec.npz"))
    assert np.array_equal(oracle.element_matvec(e["ien"], e["id"], e["elt_k"], e["u"], int(e["neq"])), e["Au"])


# ------------------------------------------------------------------ CG around the element mat-vec (conj_grad, General_matrix_functions.c:307-424)
def test_conj_grad_solves_the_system(oracle):
    ien, idmap, nno, neq = hex_mesh(4, 3, 3)
    K = spd_blocks(len(ien), 24, 6)
    BI = oracle.element_inverse_diagonal(ien, idmap, K, neq)
    A = np.zeros((neq, neq))
    for e in range(len(ien)):
        eq = idmap[ien[e]].ravel()
        A[np.ix_(eq, eq)] += K[e].reshape(24, 24)
    assert np.allclose(BI, 1.0 / np.diag(A), rtol=1e-14)
    bc = np.array(sorted(set(idmap[np.arange(0, nno, 7)].ravel().tolist())), np.int32)      # some boundary equations
    F = np.random.default_rng(1).uniform(-1, 1, neq)
    F[bc] = 0.0
    d0, cycles, res, hist = oracle.conj_grad_elem(ien, idmap, K, neq, BI, bc, F, 1e-8 * np.linalg.norm(F), 250)
    assert 1 <= cycles < 250 and res <= 1e-8 * np.linalg.norm(F) and np.all(d0[bc] == 0.0)
    free = np.setdiff1d(np.arange(neq), bc)
    sol = np.linalg.solve(A[np.ix_(free, free)], F[free])
    assert np.allclose(d0[free], sol, rtol=1e-6, atol=1e-9)
    assert np.all(np.diff(np.log(hist + 1e-300))[-3:] < 0)        # converging at the end


def test_dense_grad_matches_numpy(oracle):
    """_opt_matmul_grad.py:6-12: dxx = grad·wᵀ, dw = xxᵀ·grad."""
    rng = np.random.default_rng(3)
    for (M, N, K) in [(1, 1, 1), (7, 3, 5), (50, 20, 30)]:
        xx, w, g = rng.integers(-4, 5, (M, N)).astype(float), rng.integers(-4, 5, (N, K)).astype(float), rng.integers(-4, 5, (M, K)).astype(float)
        dxx, dw = oracle.dense_rows_times_matrix_grad(xx, w, g)
        assert np.array_equal(dxx, g @ w.T) and np.array_equal(dw, xx.T @ g)


def test_uzawa_operators_and_iteration(oracle):
    """assemble_div_u and assemble_grad_p are transposes of each other (before the boundary strip); build_diagonal_of_Ahat is the
    diagonal of G·diag(BI)·Gᵀ; solve_Ahat_p_fhat_CG drives div(V) to the requested accuracy and solves the momentum equation."""
    from tests.helpers import stokes_problem
    pr = stokes_problem(4, 3, 2, 5)
    ien, idmap, nno, neq, nel = pr["ien"], pr["id"], pr["nno"], pr["neq"], len(pr["ien"])
    rng = np.random.default_rng(0)
    U, P = rng.uniform(-1, 1, neq), rng.uniform(-1, 1, nel)
    none = np.zeros(0, np.int32)
    assert abs(oracle.assemble_div_u(ien, idmap, pr["g"], U) @ P - U @ oracle.assemble_grad_p(ien, idmap, pr["g"], neq, none, P)) < 1e-10
    # dense G: row e holds g[e] at the equations of the element's nodes
    G = np.zeros((nel, neq))
    for e in range(nel):
        for a in range(8):
            for d in range(3):
                G[e, idmap[ien[e, a], d]] += pr["g"][e, 3 * a + d]
    assert np.allclose(oracle.assemble_div_u(ien, idmap, pr["g"], U), G @ U, rtol=1e-12)
    BI = oracle.element_inverse_diagonal(ien, idmap, pr["K"], neq)
    BPI = oracle.build_diagonal_of_Ahat(ien, idmap, pr["g"], BI)
    assert np.allclose(1.0 / BPI, np.einsum("ej,j,ej->e", G, BI, G), rtol=1e-12)
    v_res = float(np.linalg.norm(pr["F"]))
    V, Pn, cnt, inc, hist, inner = oracle.solve_Ahat_p_fhat_CG(ien, idmap, nno, neq, pr["K"], pr["g"], BI, BPI, pr["nmass"], pr["area"], pr["volume"],
                                                              pr["bc"], pr["F"], np.zeros(neq), np.zeros(nel), 1e-7, 1.0, v_res, 500, 60)
    assert 2 <= cnt < 60 and inc <= 1e-7 and hist.shape == (cnt + 1, 5) and inner > cnt
    assert np.all(np.diff(hist[1:, 4]) < 0) or hist[-1, 4] < hist[1, 4]
    mom = pr["F"] - oracle.assemble_grad_p(ien, idmap, pr["g"], neq, pr["bc"], Pn) - oracle.element_matvec(ien, idmap, pr["K"], V, neq)
    mom[pr["bc"]] = 0.0
    assert np.linalg.norm(mom) <= 50 * 1e-7 * v_res


def test_node_assembled_operator_equals_element_by_element(oracle):
    """two formulations of one operator. n_assemble_del2_u on Node_map/Eqn_k (built by construct_node_ks from the
    same element matrices) agrees with the element-by-element gather to 1e-12 — with the boundary dofs weighted out of Eqn_k, for
    vectors that vanish on the boundary and with the boundary rows stripped, as CitcomS uses both."""
    from tests.helpers import hex_mesh, hex_node_map, spd_blocks
    for (ex, ey, ez, seed) in [(1, 1, 1, 0), (3, 2, 2, 1), (5, 4, 3, 2)]:
        ien, idmap, nno, neq = hex_mesh(ex, ey, ez)
        K = spd_blocks(len(ien), 24, seed)
        nm, max_eqn = hex_node_map(ex, ey, ez, idmap)
        rng = np.random.default_rng(seed)
        bc_nodes = rng.choice(nno, max(1, nno // 7), replace=False)
        bcw = np.ones((nno, 3))
        bcw[bc_nodes, rng.integers(0, 3, len(bc_nodes))] = 0.0           # one constrained direction per boundary node (VBX / VBY / VBZ)
        bc = np.array(sorted(idmap[bcw == 0.0].tolist()), np.int32)
        k1, k2, k3 = oracle.construct_node_ks(ien, idmap, nno, neq, nm, K, bcw)
        u = rng.uniform(-1, 1, neq)
        u[bc] = 0.0
        got = oracle.n_assemble_del2_u(nno, neq, nm, idmap, k1, k2, k3, u, bc)
        want = oracle.element_matvec(ien, idmap, K, u, neq)
        want[bc] = 0.0
        assert np.allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max()), (ex, ey, ez)
        # no boundary at all: the stored half reproduces the full symmetric operator for any vector
        k1, k2, k3 = oracle.construct_node_ks(ien, idmap, nno, neq, nm, K, np.ones((nno, 3)))
        v = rng.uniform(-1, 1, neq)
        assert np.allclose(oracle.n_assemble_del2_u(nno, neq, nm, idmap, k1, k2, k3, v, np.zeros(0, np.int32)), oracle.element_matvec(ien, idmap, K, v, neq),
                           rtol=0, atol=1e-12 * np.abs(want).max())


def test_outer_and_heap_spgemm_agree_with_the_hash_oracle(oracle):
    """a11: the reference's other two SpGEMM algorithms (OuterSpGEMM, mm/inc/outer_mult.h:271-542; HeapSpGEMM, mm/inc/heap_mult.h:47-223),
    restated independently, against the hash restatement: crpt and ccol bit for bit from all three; the outer-product sums run in the
    same (ascending inner index) order as the hash loop, so its values are bit-identical too; the heap's equal-key order depends on the
    heap shape, so its values agree to rounding. And all three against oneMKL's own results (tests/golden/mkl_spgemm.npz)."""
    import os
    from tests.helpers import power_law_csr, random_csr
    cases = [(random_csr(1, 1, 1.0, 0), random_csr(1, 1, 1.0, 1), 1, 1, 1),
             (random_csr(40, 30, 0.15, 2, empty_rows=[1]), random_csr(30, 50, 0.2, 3, empty_rows=[0]), 40, 30, 50),
             (random_csr(300, 300, 0.03, 4), random_csr(300, 300, 0.03, 5), 300, 300, 300),
             (power_law_csr(500, 400, 6, 300), power_law_csr(400, 700, 7, 350), 500, 400, 700),
             (random_csr(200, 8, 0.9, 8), random_csr(8, 200, 0.9, 9), 200, 8, 200)]
    for A, B, M, K, N in cases:
        h = oracle.spgemm(A, B, N, sort_output=True)
        for nblockers in (1, 4, 64):
            o = oracle.spgemm_outer(A, B, K, N, nblockers)
            assert np.array_equal(o[0], h[0]) and np.array_equal(o[1], h[1]) and np.array_equal(o[2], h[2])
        p = oracle.spgemm_heap(A, B, N)
        assert np.array_equal(p[0], h[0]) and np.array_equal(p[1], h[1])
        _, _, scale = oracle.spgemm((A[0], A[1], np.abs(A[2])), (B[0], B[1], np.abs(B[2])), N)
        assert np.all(np.abs(p[2] - h[2]) <= 1e-14 * scale + 1e-300)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mkl_spgemm.npz"))
    for name in ("tri4", "rmat8", "rect", "plaw"):
        A = tuple(g[f"{name}_a{k}"] for k in ("rpt", "col", "val"))
        B = tuple(g[f"{name}_b{k}"] for k in ("rpt", "col", "val"))
        M, K, N = (int(v) for v in g[f"{name}_mkn"])
        for got in (oracle.spgemm_outer(A, B, K, N), oracle.spgemm_heap(A, B, N)):
            assert np.array_equal(got[0], g[f"{name}_crpt"]) and np.array_equal(got[1], g[f"{name}_ccol"])
            assert np.allclose(got[2], g[f"{name}_cval"], rtol=1e-12, atol=1e-13)
