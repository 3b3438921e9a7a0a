"""GPU parity of the synthetic-input generators against the oracle's (integer outputs bit-exact, values bit-exact:
both sides evaluate the same counter-based hash)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_rmat_matches_oracle(oracle):
    from g4s_amd import host
    n, scale, edges, seed = 3000, 12, 40000, 20240521
    A = host.rmat_csr(n, scale, edges, seed, chunk=7777)      # odd chunking: counter-based → same stream
    rp, ci, va = oracle.rmat_csr(seed, scale, n, edges)
    grp, gci, gva = A.to_host()
    assert np.array_equal(grp, rp) and np.array_equal(gci, ci) and np.array_equal(gva, va)


def test_vector_laplacian_banded_match_oracle(oracle):
    from g4s_amd import host
    assert np.array_equal(host.synth_vector(7, 1000, i0=5).cpu().numpy(), oracle.vector(7, 1000, i0=5))
    for (nx, ny) in [(1, 1), (7, 5), (64, 33)]:
        A = host.laplacian_csr(5, nx, ny)
        for got, want in zip(A.to_host(), oracle.laplacian5(nx, ny)):
            assert np.array_equal(got, want)
    A = host.laplacian_csr(7, 6, 5, 4)
    for got, want in zip(A.to_host(), oracle.laplacian7(6, 5, 4)):
        assert np.array_equal(got, want)
    A = host.laplacian_csr(7, 6, 5, 4, r0=31, r1=97)          # a slab of rows, global columns
    for got, want in zip(A.to_host(), oracle.laplacian7(6, 5, 4, 31, 97)):
        assert np.array_equal(got, want)
    for (n, hb) in [(50, 3), (9, 8), (1000, 5), (5, 0)]:
        A = host.banded_csr(n, hb, 99)
        for got, want in zip(A.to_host(), oracle.banded(n, hb, 99)):
            assert np.array_equal(got, want)


def test_device_allocator_round_trip():
    """g4s_dev_alloc / g4s_dev_free: blocks come from the library's caching allocator — a freed block is handed out again for a
    request it fits within 25 %, a much smaller request gets its own block, a plain hipMalloc pointer (torch's) is not ours to cache,
    g4s_shutdown drops what is cached, and the memory is usable from torch while it is owned."""
    import ctypes as C
    from g4s_amd import capi, host
    lib = capi.load()
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = 64 << 20
    capi.check(lib.g4s_dev_alloc(C.byref(a), n))
    t = host.view_f64(a, n // 8)
    t.fill_(3.0)
    assert float(t.sum().item()) == 3.0 * (n // 8)
    del t
    capi.check(lib.g4s_dev_free(a))
    capi.check(lib.g4s_dev_alloc(C.byref(b), n - 4096))            # fits the cached block
    assert b.value == a.value
    capi.check(lib.g4s_dev_alloc(C.byref(c), 1 << 20))             # far smaller: a block of its own
    assert c.value not in (0, b.value)
    capi.check(lib.g4s_dev_free(b))
    capi.check(lib.g4s_dev_free(c))
    capi.check(lib.g4s_dev_free(None))
    capi.check(lib.g4s_shutdown())
    capi.check(lib.g4s_dev_alloc(C.byref(a), 4096))                # the library keeps working after a shutdown
    capi.check(lib.g4s_dev_free(a))
