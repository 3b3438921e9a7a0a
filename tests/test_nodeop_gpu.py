"""GPU parity of the node-assembled operator (g4s_node_op_*) against the oracle's restatement of n_assemble_del2_u
(citcoms/lib/Element_calculations.c:516-577) on Node_map / Eqn_k arrays built by the oracle's construct_node_ks
(Construct_arrays.c:335-470). The device form adds the same products in another order: |diff| <= 1e-10 · Σ|terms|."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import hex_mesh, hex_node_map, spd_blocks

pytestmark = pytest.mark.gpu


def _problem(oracle, ex, ey, ez, seed):
    ien, idmap, nno, neq = hex_mesh(ex, ey, ez)
    K = spd_blocks(len(ien), 24, seed)
    nm, max_eqn = hex_node_map(ex, ey, ez, idmap)
    rng = np.random.default_rng(seed)
    bc_nodes = rng.choice(nno, max(1, nno // 9), replace=False)
    bcw = np.ones((nno, 3))
    bcw[bc_nodes] = 0.0
    bc = np.array(sorted(idmap[bc_nodes].ravel().tolist()), np.int32)
    k1, k2, k3 = oracle.construct_node_ks(ien, idmap, nno, neq, nm, K, bcw)
    return ien, idmap, nno, neq, K, nm, max_eqn, bc, (k1, k2, k3), rng


def _create(lib, capi, nno, neq, nm, max_eqn, idmap, ks):
    h = C.c_void_p()
    capi.check(lib.g4s_node_op_create(C.byref(h), nno, neq, max_eqn, np.ascontiguousarray(nm).ctypes.data, np.ascontiguousarray(idmap).ctypes.data,
                                      ks[0].ctypes.data, ks[1].ctypes.data, ks[2].ctypes.data))
    return h


@pytest.mark.parametrize("ex,ey,ez,seed", [(1, 1, 1, 0), (3, 2, 2, 1), (8, 8, 4, 2), (32, 32, 8, 3)])
def test_node_op_apply(oracle, ex, ey, ez, seed):
    from g4s_amd import capi
    lib = capi.load()
    ien, idmap, nno, neq, K, nm, max_eqn, bc, ks, rng = _problem(oracle, ex, ey, ez, seed)
    h = _create(lib, capi, nno, neq, nm, max_eqn, idmap, ks)
    bcd = torch.from_numpy(bc).cuda()
    for trial in range(2):
        u = rng.uniform(-1, 1, neq)
        ud = torch.from_numpy(u).cuda()
        Au = torch.full((neq,), float("nan"), dtype=torch.float64, device="cuda")
        capi.check(lib.g4s_node_op_apply(h, ud.data_ptr(), Au.data_ptr(), bcd.data_ptr() if trial == 0 else None, len(bc) if trial == 0 else 0, None))
        want = oracle.n_assemble_del2_u(nno, neq, nm, idmap, *ks, u, bc if trial == 0 else np.zeros(0, np.int32))
        scale = oracle.n_assemble_del2_u(nno, neq, nm, idmap, *(np.abs(k) for k in ks), np.abs(u), np.zeros(0, np.int32))
        got = Au.cpu().numpy()
        assert np.all(np.abs(got - want) <= 1e-10 * scale + 1e-300), (ex, ey, ez, trial)
        if trial == 0:
            assert np.all(got[bc] == 0.0)
    lib.g4s_node_op_destroy(h)


def test_conj_grad_on_node_op_matches_element_operator(oracle):
    """The same CG (conj_grad, General_matrix_functions.c:307-424) on the two formulations of one operator: same iteration count,
    same solution. The right-hand side vanishes on the boundary, as in CitcomS."""
    from g4s_amd import capi
    lib = capi.load()
    ien, idmap, nno, neq, K, nm, max_eqn, bc, ks, rng = _problem(oracle, 16, 16, 8, 5)
    F = rng.uniform(-1, 1, neq)
    F[bc] = 0.0
    BI = oracle.element_inverse_diagonal(ien, idmap, K, neq)
    acc = 1e-8 * np.linalg.norm(F)
    d_or, cyc_or, res_or, _ = oracle.conj_grad_elem(ien, idmap, K, neq, BI, bc, F, acc, 250)
    h = _create(lib, capi, nno, neq, nm, max_eqn, idmap, ks)
    BId, Fd, bcd = torch.from_numpy(BI).cuda(), torch.from_numpy(F).cuda(), torch.from_numpy(bc).cuda()
    d0 = torch.full((neq,), float("nan"), dtype=torch.float64, device="cuda")
    cyc, res = C.c_int32(250), C.c_double()
    capi.check(lib.g4s_conj_grad_node(h, neq, BId.data_ptr(), bcd.data_ptr(), len(bc), Fd.data_ptr(), d0.data_ptr(), acc, C.byref(cyc), C.byref(res), None))
    lib.g4s_node_op_destroy(h)
    assert abs(cyc.value - cyc_or) <= 1 and res.value <= acc
    got = d0.cpu().numpy()
    assert np.all(got[bc] == 0.0)
    assert np.allclose(got, d_or, rtol=1e-6, atol=1e-7 * np.abs(d_or).max())


def test_node_op_rejects_malformed_maps(oracle):
    from g4s_amd import capi
    lib = capi.load()
    ien, idmap, nno, neq, K, nm, max_eqn, bc, ks, rng = _problem(oracle, 2, 2, 1, 7)
    h = C.c_void_p()
    bad = nm.copy()
    bad[3, 0:3] = idmap[2]                                          # slot group 0 must be the node itself
    assert lib.g4s_node_op_create(C.byref(h), nno, neq, max_eqn, bad.ctypes.data, np.ascontiguousarray(idmap).ctypes.data, ks[0].ctypes.data, ks[1].ctypes.data,
                                  ks[2].ctypes.data) == capi.ERR_INVALID
    bad = nm.copy()
    bad[5, 4] = bad[5, 3]                                           # a slot group that is not one node's three equations
    assert lib.g4s_node_op_create(C.byref(h), nno, neq, max_eqn, bad.ctypes.data, np.ascontiguousarray(idmap).ctypes.data, ks[0].ctypes.data, ks[1].ctypes.data,
                                  ks[2].ctypes.data) == capi.ERR_UNSUPPORTED
    bad = nm.copy()
    bad[1, 7] = neq + 5
    assert lib.g4s_node_op_create(C.byref(h), nno, neq, max_eqn, bad.ctypes.data, np.ascontiguousarray(idmap).ctypes.data, ks[0].ctypes.data, ks[1].ctypes.data,
                                  ks[2].ctypes.data) in (capi.ERR_INVALID, capi.ERR_UNSUPPORTED)
    assert lib.g4s_node_op_create(C.byref(h), nno, neq, 41, nm.ctypes.data, np.ascontiguousarray(idmap).ctypes.data, ks[0].ctypes.data, ks[1].ctypes.data,
                                  ks[2].ctypes.data) == capi.ERR_INVALID
