"""GPU parity of the gather/apply graph interface (B3): spmm_dense with registered patterns, and the device-resident forms,
against the oracle (which is itself pinned to the reference's GraphProcess build, tests/test_oracle_cpu.py) and against the
golden fixture produced by the reference's GraphProcess. Tolerances: fp64, 1e-10·Σ|terms| (summation order differs)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import oracle_lib
from tests.helpers import hex_mesh, spd_blocks

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rows(arr2d, base=0):
    rows = (C.POINTER(C.c_double) * (arr2d.shape[0] + base))()
    for e in range(arr2d.shape[0]):
        rows[e + base] = arr2d[e].ctypes.data_as(C.POINTER(C.c_double))
    return rows


def test_dense_rows_times_matrix_device(oracle):
    from g4s_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(0)
    # M >= 4096 with N, K <= 128 takes the LDS-resident persistent kernel, everything else the panel kernel
    for (M, N, K) in [(1, 1, 1), (29, 13, 7), (64, 4, 16), (100, 100, 100), (1000, 50, 25), (333, 70, 130), (5000, 1, 100),
                      (4096, 100, 100), (10007, 37, 50), (5000, 128, 128), (7001, 3, 17), (4100, 129, 20), (4100, 20, 129)]:
        xx, w = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, K))
        # asymmetric integer data catches a transposed fragment map exactly
        xi = rng.integers(-3, 4, (M, N)).astype(np.float64)
        wi = rng.integers(-3, 4, (N, K)).astype(np.float64)
        for a, b, exact in ((xi, wi, True), (xx, w, False)):
            ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            rd = torch.full((M, K), float("nan"), dtype=torch.float64, device="cuda")
            capi.check(lib.g4s_dense_rows_times_matrix(M, N, K, ad.data_ptr(), bd.data_ptr(), rd.data_ptr(), None))
            got = rd.cpu().numpy()
            want = oracle.dense_rows_times_matrix(a, b)
            if exact:
                assert np.array_equal(got, want), (M, N, K)
            else:
                scale = np.abs(a) @ np.abs(b)
                assert np.all(np.abs(got - want) <= 1e-10 * scale + 1e-300), (M, N, K)


def test_dense_rows_times_matrix_grad_device(oracle):
    """dxx = grad·wᵀ and dw = xxᵀ·grad (_opt_matmul_grad.py:6-12) against the oracle's triple loops; integer data must be exact."""
    from g4s_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(5)
    for (M, N, K) in [(1, 1, 1), (3, 2, 5), (29, 13, 7), (64, 16, 16), (100, 100, 100), (1000, 50, 25), (333, 70, 130), (4096, 100, 100),
                      (10007, 37, 50), (5000, 128, 128), (4100, 129, 20), (4100, 20, 129), (70000, 100, 100)]:
        for exact in (True, False):
            if exact:
                xx, w, g = (rng.integers(-3, 4, sh).astype(np.float64) for sh in ((M, N), (N, K), (M, K)))
            else:
                xx, w, g = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, K)), rng.uniform(-1, 1, (M, K))
            xd, wd, gd = torch.from_numpy(xx).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(g).cuda()
            dxx = torch.full((M, N), float("nan"), dtype=torch.float64, device="cuda")
            dw = torch.full((N, K), float("nan"), dtype=torch.float64, device="cuda")
            capi.check(lib.g4s_dense_rows_times_matrix_grad(M, N, K, xd.data_ptr(), wd.data_ptr(), gd.data_ptr(), dxx.data_ptr(), dw.data_ptr(), None))
            want_dxx, want_dw = oracle.dense_rows_times_matrix_grad(xx, w, g)
            got_dxx, got_dw = dxx.cpu().numpy(), dw.cpu().numpy()
            if exact:
                assert np.array_equal(got_dxx, want_dxx) and np.array_equal(got_dw, want_dw), (M, N, K)
            else:
                assert np.all(np.abs(got_dxx - want_dxx) <= 1e-10 * (np.abs(g) @ np.abs(w).T) + 1e-300), (M, N, K)
                assert np.all(np.abs(got_dw - want_dw) <= 1e-10 * (np.abs(xx).T @ np.abs(g)) + 1e-300), (M, N, K)
    # either output may be skipped; the reduction over rows is reproducible
    M, N, K = 20000, 100, 100
    xd, wd, gd = (torch.from_numpy(rng.uniform(-1, 1, sh)).cuda() for sh in ((M, N), (N, K), (M, K)))
    a, b = torch.zeros((N, K), dtype=torch.float64, device="cuda"), torch.zeros((N, K), dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_dense_rows_times_matrix_grad(M, N, K, xd.data_ptr(), wd.data_ptr(), gd.data_ptr(), None, a.data_ptr(), None))
    capi.check(lib.g4s_dense_rows_times_matrix_grad(M, N, K, xd.data_ptr(), wd.data_ptr(), gd.data_ptr(), None, b.data_ptr(), None))
    assert torch.equal(a, b)


def test_dense_golden_from_reference_graphprocess():
    from g4s_amd import capi
    lib = capi.load()
    d = np.load(os.path.join(GOLD, "graphprocess_dense.npz"))      # result produced by the reference's GraphProcess
    xx, w = d["xx"], d["w"]
    M, N = xx.shape
    K = w.shape[1]
    rd = torch.empty((M, K), dtype=torch.float64, device="cuda")
    xd, wd = torch.from_numpy(xx).cuda(), torch.from_numpy(w).cuda()          # keep the device copies alive across the call
    capi.check(lib.g4s_dense_rows_times_matrix(M, N, K, xd.data_ptr(), wd.data_ptr(), rd.data_ptr(), None))
    assert np.all(np.abs(rd.cpu().numpy() - d["result"]) <= 1e-10 * (np.abs(xx) @ np.abs(w)))


def test_scoped_pattern_keeps_the_reference_call_line(tmp_path):
    """Round 5 (missing 2): g4s::ScopedPattern in front of the UNTOUCHED four-argument GraphProcess(graph, result, gather, apply) call
    (deepmd/source/op/opt_matmul.cc:51, graph.h:21-32) lands it on the fp64 MFMA kernel. examples/graph_process_host.cpp holds that call line once; "scoped_pattern"
    wraps it in the scope. Checked against the fixture the REFERENCE's GraphProcess produced (tests/golden/graphprocess_dense.npz) and, bit for bit, against the
    five-argument device form (g4s_dense_rows_times_matrix) — the scope must reach the same kernel, not the host loop."""
    import subprocess
    from g4s_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "g4s_amd", "lib")
    exe = str(tmp_path / "graph_process_host")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "graph_process_host.cpp"),
                           "-L" + libdir, "-lg4s_hip", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe])
    d = np.load(os.path.join(GOLD, "graphprocess_dense.npz"))
    xx, w = np.ascontiguousarray(d["xx"]), np.ascontiguousarray(d["w"])
    M, N = xx.shape
    K = w.shape[1]
    payload = f"{M} {N} {K}\n".encode() + xx.tobytes() + w.tobytes()
    out = subprocess.run([exe, "scoped_pattern"], input=payload, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    got = np.frombuffer(out.stdout, dtype=np.float64).reshape(M, K)
    assert np.all(np.abs(got - d["result"]) <= 1e-10 * (np.abs(xx) @ np.abs(w)))
    lib = capi.load()
    rd = torch.empty((M, K), dtype=torch.float64, device="cuda")
    xd, wd = torch.from_numpy(xx).cuda(), torch.from_numpy(w).cuda()
    capi.check(lib.g4s_dense_rows_times_matrix(M, N, K, xd.data_ptr(), wd.data_ptr(), rd.data_ptr(), None))
    assert np.array_equal(got, rd.cpu().numpy()), "the scoped call did not run the pattern's kernel"
    host_loop = subprocess.run([exe, "serial"], input=payload, capture_output=True, timeout=300)   # without the scope: the host loop, the reference's own sums
    assert host_loop.returncode == 0 and np.array_equal(np.frombuffer(host_loop.stdout, dtype=np.float64).reshape(M, K), d["result"])


def test_element_op_device(oracle):
    from g4s_amd import capi
    lib = capi.load()
    for (ex, ey, ez, seed) in [(1, 1, 1, 0), (2, 2, 2, 3), (8, 5, 3, 4), (16, 16, 4, 5)]:
        ien, idmap, nno, neq = hex_mesh(ex, ey, ez)
        K = spd_blocks(len(ien), 24, seed)
        u = np.random.default_rng(seed).uniform(-1, 1, neq)
        want = oracle.element_matvec(ien, idmap, K, u, neq)
        scale = oracle.element_matvec(ien, idmap, np.abs(K), np.abs(u), neq)
        Kd, ud = torch.from_numpy(K).cuda(), torch.from_numpy(u).cuda()
        Aud = torch.full((neq,), float("nan"), dtype=torch.float64, device="cuda")
        h = C.c_void_p()
        capi.check(lib.g4s_elem_op_create(C.byref(h), len(ien), 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data,
                                          nno, neq, Kd.data_ptr()))
        capi.check(lib.g4s_elem_op_apply(h, ud.data_ptr(), Aud.data_ptr(), None))
        got = Aud.cpu().numpy()
        capi.check(lib.g4s_elem_op_apply(h, ud.data_ptr(), Aud.data_ptr(), None))
        assert np.array_equal(got, Aud.cpu().numpy())                 # reproducible: no atomics
        lib.g4s_elem_op_destroy(h)
        assert np.all(np.abs(got - want) <= 1e-10 * scale + 1e-300)


def test_element_golden_fixture():
    from g4s_amd import capi
    lib = capi.load()
    e = np.load(os.path.join(GOLD, "element_matvec.npz"))
    neq, nno = int(e["neq"]), int(e["nno"])
    Kd, ud = torch.from_numpy(e["elt_k"]).cuda(), torch.from_numpy(e["u"]).cuda()
    Aud = torch.empty(neq, dtype=torch.float64, device="cuda")
    h = C.c_void_p()
    capi.check(lib.g4s_elem_op_create(C.byref(h), len(e["ien"]), 8, 3, np.ascontiguousarray(e["ien"]).ctypes.data,
                                      np.ascontiguousarray(e["id"]).ctypes.data, nno, neq, Kd.data_ptr()))
    capi.check(lib.g4s_elem_op_apply(h, ud.data_ptr(), Aud.data_ptr(), None))
    lib.g4s_elem_op_destroy(h)
    assert np.allclose(Aud.cpu().numpy(), e["Au"], rtol=1e-12, atol=1e-12 * np.abs(e["Au"]).max())


def test_element_op_rejects_bad_maps():
    from g4s_amd import capi
    lib = capi.load()
    ien, idmap, nno, neq = hex_mesh(1, 1, 1)
    h = C.c_void_p()
    bad = idmap.copy()
    bad[0, 0] = neq                                          # out of range
    assert lib.g4s_elem_op_create(C.byref(h), 1, 8, 3, ien.ctypes.data, bad.ctypes.data, nno, neq, None) == capi.ERR_INVALID
    bad = idmap.copy()
    bad[1, 0] = bad[0, 0]                                    # two owners of one equation
    assert lib.g4s_elem_op_create(C.byref(h), 1, 8, 3, ien.ctypes.data, bad.ctypes.data, nno, neq, None) == capi.ERR_INVALID


def test_spmm_dense_with_registered_callbacks(oracle):
    """The reference call shape end to end: host row pointers, host callbacks (which the device never calls) used as the
    registration key, results compared with the oracle running those same callbacks' arithmetic."""
    from g4s_amd import capi
    lib = capi.load()

    # ---- CitcomS element pattern, 1-based edgeWeight slot as in Drive_solvers.c:52-55
    ien, idmap, nno, neq = hex_mesh(4, 3, 2)
    nel = len(ien)
    K = spd_blocks(nel, 24, 8)
    u = np.random.default_rng(8).uniform(-1, 1, neq + 1)       # CitcomS vectors carry one padding slot
    u[neq] = 0.0

    @capi.FUN_GATHER
    def gather_elem(e, a, ew, st, res):                         # never called by the device path
        raise RuntimeError("host callback must not run")

    @capi.FUN_APPLY
    def apply_elem(e, ew, st, res):
        raise RuntimeError("host callback must not run")

    desc = capi.PatternDesc(kind=capi.PATTERN_ELEMENT_BLOCK_MATVEC, num_elems=nel, nodes_per_elem=8, dof=3,
                            ien=np.ascontiguousarray(ien).ctypes.data, id=np.ascontiguousarray(idmap).ctypes.data, nno=nno, neq=neq,
                            edge_weight_base=1, static_weights=1)
    capi.check(lib.g4s_register_pattern(gather_elem, apply_elem, C.byref(desc)))
    rows = _rows(K, base=1)
    Au = np.zeros(neq + 1)                                      # caller zeroes Au (Element_calculations.c:495-496)
    tm = C.c_double(-1.0)
    lib.spmm_dense(nel, 8, C.cast(rows, C.c_void_p), u.ctypes.data, Au.ctypes.data, Au.ctypes.data, gather_elem, apply_elem, C.byref(tm), 1)
    want = oracle.element_matvec(ien, idmap, K, u[:neq], neq, base=1)
    scale = oracle.element_matvec(ien, idmap, np.abs(K), np.abs(u[:neq]), neq)
    assert np.all(np.abs(Au[:neq] - want) <= 1e-10 * scale + 1e-300) and Au[neq] == 0.0 and tm.value >= 0.0
    # second call (cached weights) accumulates into result as the callbacks do (+=)
    lib.spmm_dense(nel, 8, C.cast(rows, C.c_void_p), u.ctypes.data, Au.ctypes.data, Au.ctypes.data, gather_elem, apply_elem, C.byref(tm), 1)
    assert np.all(np.abs(Au[:neq] - 2 * want) <= 2e-10 * scale + 1e-300)

    # ---- DeePMD dense pattern (graph.numNodes = M, degree = K, edgeWeight = row pointers into xx, states = w)
    M, N, Kc = 77, 25, 50
    rng = np.random.default_rng(5)
    xx, w = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, Kc))

    @capi.FUN_GATHER
    def gather_dense(e, a, ew, st, res):
        raise RuntimeError("host callback must not run")

    @capi.FUN_APPLY
    def apply_dense(e, ew, st, res):
        raise RuntimeError("host callback must not run")

    desc2 = capi.PatternDesc(kind=capi.PATTERN_DENSE_ROW_TIMES_MATRIX, inner=N)
    capi.check(lib.g4s_register_pattern(gather_dense, apply_dense, C.byref(desc2)))
    res = np.full((M, Kc), np.nan)
    lib.spmm_dense(M, Kc, C.cast(_rows(xx), C.c_void_p), w.ctypes.data, None, res.ctypes.data, gather_dense, apply_dense, None, 8)
    assert np.all(np.abs(res - oracle.dense_rows_times_matrix(xx, w)) <= 1e-10 * (np.abs(xx) @ np.abs(w)))

    # ---- Cantera quadratic forms (numbers = 1 and 2)
    m = 53
    a1, x, b = rng.uniform(0, 1, m * m), rng.uniform(0, 1, m), rng.uniform(0, 1, m)

    @capi.FUN_GATHER
    def gather_q(i, j, ew, st, res):
        raise RuntimeError("host callback must not run")

    @capi.FUN_APPLY
    def apply_q(i, ew, st, res):
        raise RuntimeError("host callback must not run")

    desc3 = capi.PatternDesc(kind=capi.PATTERN_SYM_QUADRATIC_FORM, numbers=1)
    capi.check(lib.g4s_register_pattern(gather_q, apply_q, C.byref(desc3)))
    out = np.array([0.5, 0.25])                                 # accumulated into, as c[0], c[1] are in the reference
    lib.spmm_dense(m, m, C.cast(_rows(a1.reshape(1, -1)), C.c_void_p), x.ctypes.data, b.ctypes.data, out.ctypes.data, gather_q, apply_q, None, 3)
    want = oracle.sym_quadratic_form(m, 1, a1, x, b)
    assert abs(out[0] - 0.5 - want[0]) <= 1e-12 * abs(want[0]) and abs(out[1] - 0.25 - want[1]) <= 1e-12 * abs(want[1])
    a2 = rng.uniform(0, 1, 2 * m * m)
    out2 = np.zeros(2)
    capi.check(lib.g4s_sym_quadratic_form(m, 2, a2.ctypes.data, x.ctypes.data, None, out2.ctypes.data))
    want2 = oracle.sym_quadratic_form(m, 2, a2, x)
    assert np.all(np.abs(out2 - want2) <= 1e-12 * np.abs(want2))

    # ---- an unregistered pair gets the reference's host loop (tests/test_host_callbacks_cpu.py pins it to the reference build); with the
    # REFUSE policy it is turned away before any callback runs
    @capi.FUN_GATHER
    def gather_unknown(e, a, ew, st, res):
        raise RuntimeError("host callback must not run under the REFUSE policy")

    capi.check(lib.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_REFUSE))
    try:
        st = lib.g4s_spmm_dense(M, Kc, C.cast(_rows(xx), C.c_void_p), w.ctypes.data, None, res.ctypes.data, gather_unknown, apply_dense, None, 1)
        assert st == capi.ERR_UNSUPPORTED and b"refused" in lib.g4s_last_error()
    finally:
        capi.check(lib.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_SERIAL))
    for g, a in ((gather_elem, apply_elem), (gather_dense, apply_dense), (gather_q, apply_q)):
        capi.check(lib.g4s_unregister_pattern(g, a))


def test_cookbook2_sized_element_op(oracle):
    """BASELINE config 5 sizes (CitcomS Cookbook2: 32×32×8 elements, nno 9801, neq 29403, nel 8192) with synthetic SPD blocks:
    element-by-element device result == assembled CSR matrix times u through the SpMV kernel (two formulations, one operator)."""
    from g4s_amd import capi, host
    import scipy.sparse as sp
    lib = capi.load()
    ien, idmap, nno, neq = hex_mesh(32, 32, 8)
    assert (nno, neq, len(ien)) == (9801, 29403, 8192)
    K = spd_blocks(8192, 24, 1)
    u = np.random.default_rng(1).uniform(-1, 1, neq)
    eq = idmap[ien].reshape(8192, 24)
    rows = np.repeat(eq, 24, axis=1).ravel()
    cols = np.tile(eq, (1, 24)).ravel()
    A = sp.coo_matrix((K.ravel(), (rows, cols)), shape=(neq, neq)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    Acsr = host.CSR.from_host(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, neq, neq)
    ud = torch.from_numpy(u).cuda()
    y_csr = Acsr.spmv(ud).cpu().numpy()
    Kd = torch.from_numpy(K).cuda()
    Aud = torch.empty(neq, dtype=torch.float64, device="cuda")
    h = C.c_void_p()
    capi.check(lib.g4s_elem_op_create(C.byref(h), 8192, 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq,
                                      Kd.data_ptr()))
    capi.check(lib.g4s_elem_op_apply(h, ud.data_ptr(), Aud.data_ptr(), None))
    lib.g4s_elem_op_destroy(h)
    scale = np.abs(A) @ np.abs(u)
    assert np.all(np.abs(Aud.cpu().numpy() - y_csr) <= 1e-10 * scale)
    want = oracle.element_matvec(ien, idmap, K, u, neq)
    assert np.all(np.abs(Aud.cpu().numpy() - want) <= 1e-10 * scale)
