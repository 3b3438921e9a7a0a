"""g4s_csr_update_values / g4s_spmv_dist_update_values: new values into an existing plan (round 4). A time-stepping caller rebuilds its operator with the same
pattern again and again (CitcomS: citcoms/lib/Drive_solvers.c:88,134 → construct_stiffness_B_matrix, Construct_arrays.c:740); the plan — row blocks, the blocked
path's regrouping, the diagonal / block-row layouts — depends on the pattern only. After an update a handle must behave like one created from the new values:
bit for bit on the three reproducible paths, within the tolerance on the blocked path (its sums are LDS atomics: last bits move from run to run anyway)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import assemble_csr, hex_mesh, power_law_csr, random_csr, spd_blocks, stokes_problem

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _cases(oracle):
    from g4s_amd import capi
    rng = np.random.default_rng(5)
    out = []
    rp, ci, va = random_csr(30000, 30000, 0.0006, 3)
    out.append(("stream", rp, ci, va, 30000, 30000, capi.SPMV_STREAM, 0, True))
    rp, ci, va = power_law_csr(60000, 60000, 17, 9000)
    out.append(("blocked+map", rp, ci, va, 60000, 60000, capi.SPMV_BLOCKED | capi.SPMV_UPDATABLE, 1, False))
    out.append(("blocked, no map (re-plan)", rp, ci, va, 60000, 60000, capi.SPMV_BLOCKED, 1, False))
    rp, ci, va = oracle.laplacian7(40, 31, 22)
    out.append(("diagonal", rp, ci, rng.uniform(-1, 1, len(ci)), 40 * 31 * 22, 40 * 31 * 22, 0, 3, True))
    ien, idmap, nno, neq = hex_mesh(10, 9, 7)
    rp, ci, va = assemble_csr(ien, idmap, spd_blocks(len(ien), 24, 9), neq)
    out.append(("block-row", rp, ci, va, neq, neq, 0, 4, True))
    return out


def test_update_values_every_path(oracle):
    from g4s_amd import capi, host
    rng = np.random.default_rng(11)
    for name, rp, ci, va, rows, cols, flags, path, exact in _cases(oracle):
        A = host.CSR.from_host(rp, ci, va, rows, cols, spmv_flags=flags)
        assert A.info()["spmv_path"] == path, (name, A.info())
        x = rng.uniform(-1, 1, cols)
        xd = torch.from_numpy(x).cuda()
        A.spmv(xd)                                                  # a product with the old values first
        for rep in range(2):
            vnew = rng.uniform(-1, 1, len(ci)) + rep
            A.update_values(torch.from_numpy(vnew).cuda())
            y = A.spmv(xd).cpu().numpy()
            B = host.CSR.from_host(rp, ci, vnew, rows, cols, spmv_flags=flags)
            yb = B.spmv(xd).cpu().numpy()
            want = oracle.spmv(rp, ci, vnew, x)
            _, asum = oracle.spmv_ld(rp, ci, vnew, x)
            assert np.all(np.abs(y - want) <= TOL * asum + 1e-300), name
            if exact:
                assert np.array_equal(y, yb), f"{name}: an updated handle differs from a fresh one"
            else:
                assert np.all(np.abs(y - yb) <= TOL * asum + 1e-300), name
            assert A.info()["spmv_path"] == path
            B.close()
        A.close()


def test_update_values_owned_copy_and_in_place(oracle):
    """A handle created from HOST arrays owns device copies: new values arrive as a host array (copied in) or a device array (copied in). A handle that BORROWS
    device arrays takes values == NULL as "rewritten in place"."""
    from g4s_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(2)
    rp, ci, va = oracle.laplacian7(24, 19, 13)
    n = 24 * 19 * 13
    va = rng.uniform(-1, 1, len(ci))
    h = C.c_void_p()
    capi.check(lib.g4s_csr_create(C.byref(h), n, n, len(ci), rp.ctypes.data, ci.ctypes.data, va.ctypes.data, capi.HOST_POINTERS))
    x = rng.uniform(-1, 1, n)
    xd, yd = torch.from_numpy(x).cuda(), torch.empty(n, dtype=torch.float64, device="cuda")
    v1 = rng.uniform(-1, 1, len(ci))
    capi.check(lib.g4s_csr_update_values(h, v1.ctypes.data, capi.HOST_POINTERS, None))
    capi.check(lib.g4s_spmv(h, xd.data_ptr(), yd.data_ptr(), 1.0, 0.0, None))
    assert np.array_equal(yd.cpu().numpy(), oracle.spmv(rp, ci, v1, x))
    v2 = torch.from_numpy(rng.uniform(-1, 1, len(ci))).cuda()
    capi.check(lib.g4s_csr_update_values(h, v2.data_ptr(), capi.DEVICE_POINTERS, None))
    capi.check(lib.g4s_spmv(h, xd.data_ptr(), yd.data_ptr(), 1.0, 0.0, None))
    assert np.array_equal(yd.cpu().numpy(), oracle.spmv(rp, ci, v2.cpu().numpy(), x))
    lib.g4s_csr_destroy(h)
    # borrowed arrays rewritten in place
    rpd, cid, vad = torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(va).cuda()
    torch.cuda.synchronize()
    capi.check(lib.g4s_csr_create(C.byref(h), n, n, len(ci), rpd.data_ptr(), cid.data_ptr(), vad.data_ptr(), capi.DEVICE_POINTERS))
    vad.copy_(v2)
    torch.cuda.synchronize()
    capi.check(lib.g4s_csr_update_values(h, None, capi.DEVICE_POINTERS, None))
    capi.check(lib.g4s_spmv(h, xd.data_ptr(), yd.data_ptr(), 1.0, 0.0, None))
    assert np.array_equal(yd.cpu().numpy(), oracle.spmv(rp, ci, v2.cpu().numpy(), x))
    # a host array for a handle that borrows: refused
    assert lib.g4s_csr_update_values(h, v1.ctypes.data, capi.HOST_POINTERS, None) == capi.ERR_INVALID
    lib.g4s_csr_destroy(h)


def test_update_values_partitioned_handle(oracle):
    """The row-partitioned operator (one rank in loopback: own + remote parts, and the merged form through a forced merge is covered by the two-part path
    of the same code): new values for the slab through g4s_spmv_dist_update_values, product equal to the oracle's with the new values."""
    import os
    from g4s_amd import capi, dist as gdist
    n = 40000
    rp, ci, va = power_law_csr(n, n, 41, 5000)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, n)
    d = [torch.from_numpy(a).cuda() for a in (rp, ci, va)]
    D = gdist.DistSpMV([0, n], 0, 1, *d, n, loopback=True, spmv_flags=capi.SPMV_UPDATABLE)
    xl = torch.from_numpy(x).cuda()
    D(xl)
    for rep in range(2):
        vnew = rng.uniform(-1, 1, len(ci))
        vd = torch.from_numpy(vnew).cuda()
        capi.check(D.lib.g4s_spmv_dist_update_values(D.h, vd.data_ptr(), capi.DEVICE_POINTERS, None))
        y = D(xl).cpu().numpy()
        want = oracle.spmv(rp, ci, vnew, x)
        _, asum = oracle.spmv_ld(rp, ci, vnew, x)
        assert np.all(np.abs(y - want) <= TOL * asum + 1e-300)
    D.close()
    E = gdist.DistSpMV([0, n], 0, 1, *d, n, loopback=True)          # without the flag: refused, not a crash
    assert E.lib.g4s_spmv_dist_update_values(E.h, d[2].data_ptr(), capi.DEVICE_POINTERS, None) == capi.ERR_INVALID
    E.close()


def test_stokes_two_solves_with_a_changed_stiffness_matrix(oracle):
    """The caller pattern the entry point exists for: the assembled K_csr of a Stokes solve gets new values (a new viscosity field: citcoms/lib/Drive_solvers.c:134
    inside the viscosity iteration) and the SAME handle drives the next solve — outer iteration count and solution equal to the oracle's solve with the new
    element matrices."""
    from g4s_amd import capi, host
    lib = capi.load()
    pr = stokes_problem(6, 6, 4, 1)
    ien, idmap, nno, neq, nel = pr["ien"], pr["id"], pr["nno"], pr["neq"], len(pr["ien"])
    imp, scale, vlow, steps = 1e-6, 1.0, 500, 40
    v_res = float(np.linalg.norm(pr["F"]))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    K2 = pr["K"] * np.random.default_rng(4).uniform(0.5, 2.0, nel)[:, None]       # every element matrix scaled by its own "viscosity"
    rp, ci, va1 = assemble_csr(ien, idmap, pr["K"], neq)
    _, _, va2 = assemble_csr(ien, idmap, K2, neq)
    Acsr = host.CSR.from_host(rp, ci, va1, neq, neq)
    assert Acsr.info()["spmv_path"] == 4                            # block-row path: the plan keeps its own copy of the values
    gd, nmd, ard, bcd, Fd = dev(pr["g"]), dev(pr["nmass"]), dev(pr["area"]), dev(pr["bc"]), dev(pr["F"])
    for Ke, va in ((pr["K"], None), (K2, va2)):
        if va is not None:
            Acsr.update_values(dev(va))
        BI = oracle.element_inverse_diagonal(ien, idmap, Ke, neq)
        BPI = oracle.build_diagonal_of_Ahat(ien, idmap, pr["g"], BI)
        Vo, Po, cnt_o, inc_o, hist_o, inner_o = oracle.solve_Ahat_p_fhat_CG(ien, idmap, nno, neq, Ke, pr["g"], BI, BPI, pr["nmass"], pr["area"], pr["volume"],
                                                                           pr["bc"], pr["F"], np.zeros(neq), np.zeros(nel), imp, scale, v_res, vlow, steps, 0, 0)
        Kd = dev(Ke)
        h = C.c_void_p()
        capi.check(lib.g4s_elem_op_create(C.byref(h), nel, 8, 3, np.ascontiguousarray(ien).ctypes.data, np.ascontiguousarray(idmap).ctypes.data, nno, neq, Kd.data_ptr()))
        Vd, Pd, BId, BPId = dev(np.zeros(neq)), dev(np.zeros(nel)), dev(BI), dev(BPI)    # (named: a temporary's memory would be reused before the solve reads it)
        prm = capi.StokesParams(imp, scale, v_res, vlow, steps, 0, 0)
        res = capi.StokesResult()
        capi.check(lib.g4s_stokes_uzawa_cg(h, Acsr.handle, gd.data_ptr(), BId.data_ptr(), BPId.data_ptr(), nmd.data_ptr(), ard.data_ptr(), pr["volume"], bcd.data_ptr(),
                                           len(pr["bc"]), Fd.data_ptr(), Vd.data_ptr(), Pd.data_ptr(), C.byref(prm), C.byref(res), None, 0, None))
        lib.g4s_elem_op_destroy(h)
        assert res.outer_iterations == cnt_o
        assert np.allclose(Vd.cpu().numpy(), Vo, rtol=0, atol=1e-8 * np.abs(Vo).max())
        assert np.allclose(Pd.cpu().numpy(), Po, rtol=0, atol=1e-7 * np.abs(Po).max())
    Acsr.close()
