"""The general case of the graph interface (B3): a (gather, apply) pair that is NOT one of the registered patterns runs the reference's driver
loop on the host, in the product (g4s_spmm_dense / spmm_dense, g4s_amd/csrc/graph.hip). Compared bit for bit with the REFERENCE's own
GraphProcess (deepmd/source/op/graph.h:21-32) compiled in place into oracle/_ref — the one place a reference-built pin exists. No GPU needed:
host callbacks never touch the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# C callbacks for the threaded policy (Python callbacks would serialise on the GIL): the OptMatmul gather shape of
# deepmd/source/op/opt_matmul.cc:52-58 — a race-free gather, every (vertex, neighbour) writes its own result slot — written here, not copied.
CB_SRC = r"""
int g_inner = 0, g_degree = 0;
void cb_gather(int e, int a, const double **ew, const double *st, double *res)
{
    double s = 0.0;
    for (int k = 0; k < g_inner; ++k) s += ew[e][k] * st[k * g_degree + a];
    res[e * (g_degree + 1) + a] = s;
}
void cb_apply(int e, const double **ew, const double *st, double *res)
{
    double s = 0.0;                      /* reads what this vertex's gathers wrote: apply must come after all of them */
    for (int a = 0; a < g_degree; ++a) s += res[e * (g_degree + 1) + a];
    res[e * (g_degree + 1) + g_degree] = s + ew[e][0] * st[0];
}
"""


@pytest.fixture(scope="module")
def ref():
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libref_graph.so not built (no /root/reference here and no prebuilt copy)")
    return lib


@pytest.fixture(scope="module")
def g4s():
    from g4s_amd import capi
    return capi.load()


@pytest.fixture(scope="module")
def cb(tmp_path_factory):
    d = tmp_path_factory.mktemp("cb")
    src, so = d / "cb.c", d / "libcb.so"
    src.write_text(CB_SRC)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", str(so), str(src)])
    return C.CDLL(str(so))


def _rows(xx):
    return (C.POINTER(C.c_double) * len(xx))(*[xx[i].ctypes.data_as(C.POINTER(C.c_double)) for i in range(len(xx))])


def test_unregistered_python_callbacks_run_in_reference_order(g4s, ref):
    """One thread, ascending vertices, gather 0 … degree−1 then apply: a gather that depends on what earlier VERTICES left in result (a running
    value carried through result[0]) and an apply that reads its vertex's gathers — any other order changes the bits."""
    from g4s_amd import capi
    rng = np.random.default_rng(5)
    M, deg = 23, 6
    xx = rng.uniform(-1, 1, (M, deg))
    states = rng.uniform(-1, 1, deg)
    rows = _rows(xx)
    calls = []

    @oracle_lib.FUN_GATHER
    def gather(vi, nb, ew, st, res):
        calls.append((vi, nb))
        res[1 + vi * (deg + 1) + nb] = ew[vi][nb] * st[nb] + res[0]
        res[0] = res[0] * 0.5 + ew[vi][nb]                       # order-dependent carry

    @oracle_lib.FUN_APPLY
    def apply(vi, ew, st, res):
        calls.append((vi, -1))
        res[1 + vi * (deg + 1) + deg] = sum(res[1 + vi * (deg + 1) + j] for j in range(deg)) - res[0]

    assert g4s.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_SERIAL) == 0
    got, want = np.zeros(1 + M * (deg + 1)), np.zeros(1 + M * (deg + 1))
    secs = C.c_double(-1.0)
    capi.check(g4s.g4s_spmm_dense(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, got.ctypes.data, gather, apply, C.byref(secs), 8))
    mine = list(calls)
    assert mine == [(v, n) for v in range(M) for n in list(range(deg)) + [-1]]
    assert secs.value >= 0.0
    # the reference's loop is an OpenMP parallel for over 8 threads (graph.h:23-24): with an order-dependent gather it is only deterministic on
    # one thread, which is what OMP_NUM_THREADS cannot force (omp_set_num_threads(8) is hard-coded) — so compare on a carry-free copy below
    # and keep the order assertion above for the sequencing
    del calls[:]

    @oracle_lib.FUN_GATHER
    def gather2(vi, nb, ew, st, res):
        res[1 + vi * (deg + 1) + nb] = ew[vi][nb] * st[nb] + vi

    @oracle_lib.FUN_APPLY
    def apply2(vi, ew, st, res):
        res[1 + vi * (deg + 1) + deg] = sum(res[1 + vi * (deg + 1) + j] for j in range(deg))

    got[:] = 0.0
    capi.check(g4s.g4s_spmm_dense(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, got.ctypes.data, gather2, apply2, None, 8))
    ref.ref_graph_process_cb(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, want.ctypes.data, gather2, apply2)
    assert np.array_equal(got, want)
    # the void reference symbol takes the same path
    got[:] = 0.0
    g4s.spmm_dense(M, deg, C.cast(rows, C.c_void_p), states.ctypes.data, None, got.ctypes.data, gather2, apply2, None, 8)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("policy,threads", [("serial", 8), ("parallel", 8), ("parallel", 3), ("parallel", 1)])
def test_unregistered_c_callbacks_equal_the_reference_build(g4s, ref, cb, policy, threads):
    """Race-free C callbacks (the OptMatmul gather shape): the product's host loop — serial, and with threadNum threads when the caller has declared
    the gather race-free — against the reference's GraphProcess on its 8 OpenMP threads, bit for bit."""
    from g4s_amd import capi
    rng = np.random.default_rng(11)
    M, N, K = 301, 17, 9
    xx, w = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, K))
    rows = _rows(xx)
    C.c_int.in_dll(cb, "g_inner").value = N
    C.c_int.in_dll(cb, "g_degree").value = K
    gather = C.cast(cb.cb_gather, oracle_lib.FUN_GATHER)
    apply = C.cast(cb.cb_apply, oracle_lib.FUN_APPLY)
    got, want = np.zeros(M * (K + 1)), np.zeros(M * (K + 1))
    assert g4s.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_PARALLEL if policy == "parallel" else capi.HOST_CALLBACKS_SERIAL) == 0
    try:
        capi.check(g4s.g4s_spmm_dense(M, K, C.cast(rows, C.c_void_p), w.ctypes.data, None, got.ctypes.data, gather, apply, None, threads))
    finally:
        g4s.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_SERIAL)
    ref.ref_graph_process_cb(M, K, C.cast(rows, C.c_void_p), w.ctypes.data, None, want.ctypes.data, gather, apply)
    assert np.array_equal(got, want)
    assert np.allclose(got.reshape(M, K + 1)[:, :K], xx @ w, rtol=1e-13, atol=1e-13)


def test_refusal_stays_reachable(g4s):
    from g4s_amd import capi

    @oracle_lib.FUN_GATHER
    def gather(vi, nb, ew, st, res):
        raise AssertionError("must not be called")

    @oracle_lib.FUN_APPLY
    def apply(vi, ew, st, res):
        raise AssertionError("must not be called")

    res = np.zeros(4)
    assert g4s.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_REFUSE) == 0
    try:
        st = g4s.g4s_spmm_dense(2, 2, None, None, None, res.ctypes.data, gather, apply, None, 1)
        assert st == capi.ERR_UNSUPPORTED and b"refused" in g4s.g4s_last_error()
    finally:
        g4s.g4s_set_host_callback_policy(capi.HOST_CALLBACKS_SERIAL)
    assert g4s.g4s_set_host_callback_policy(7) == capi.ERR_INVALID


def test_four_argument_graphprocess_is_the_reference_spelling(tmp_path, ref):
    """include/g4s/graph.hpp's global GraphProcess(graph, result, gather, apply) — the signature of deepmd/source/op/graph.h:21, std::function callbacks, no
    descriptor — in a C++ program that builds the OptMatmul graph the way deepmd/source/op/opt_matmul.cc:43-61 does: against the reference's own GraphProcess
    (oracle/_ref) bit for bit, serial and with the race-free declaration; the refusal policy surfaces as an exception."""
    import subprocess
    lib = os.path.join(ROOT, "g4s_amd", "lib")
    exe = str(tmp_path / "graph_process_host")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "graph_process_host.cpp"),
                           "-L" + lib, "-lg4s_hip", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe])
    rng = np.random.default_rng(23)
    M, N, K = 257, 19, 7
    xx, w = rng.uniform(-1, 1, (M, N)), rng.uniform(-1, 1, (N, K))
    want = np.zeros(M * K)
    ref.ref_graph_process_dense(M, N, K, xx.ravel(), w.ravel(), want)
    payload = f"{M} {N} {K}\n".encode() + xx.tobytes() + w.tobytes()
    for mode in ("serial", "parallel", "scoped_race_free"):        # scoped_race_free: g4s::ScopedRaceFree around the untouched call instead of the process-wide policy
        out = subprocess.run([exe, mode], input=payload, capture_output=True, timeout=120)
        assert out.returncode == 0, out.stderr.decode()
        got = np.frombuffer(out.stdout, dtype=np.float64)
        assert np.array_equal(got, want), mode
    out = subprocess.run([exe, "refuse"], input=payload, capture_output=True, timeout=120)
    assert out.returncode == 3 and b"refused" in out.stderr
