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


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100_000, 3_000_001])
def test_own_exclusive_scan_equals_cumsum(n):
    """csrc/prims.hpp: the library's exclusive prefix sum (three kernels, DPP/shuffle wave scans) that replaced hipcub::DeviceScan on the SpGEMM call
    path and in the SpMV plan — int32 and int64, tile boundaries, more tiles than one offsets pass."""
    import ctypes as C
    from g4s_amd import capi, host
    lib = capi.load()
    g = torch.Generator(device="cuda").manual_seed(n)
    a = torch.randint(0, 1000, (n,), dtype=torch.int32, device="cuda", generator=g)
    out = torch.empty_like(a)
    capi.check(lib.g4s_prim_exclusive_scan_i32(a.data_ptr(), out.data_ptr(), n, host._stream()))
    want = torch.cumsum(a.long(), 0) - a.long()
    assert torch.equal(out.long(), want)
    b = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda", generator=g)
    out64 = torch.empty_like(b)
    capi.check(lib.g4s_prim_exclusive_scan_i64(b.data_ptr(), out64.data_ptr(), n, host._stream()))
    assert torch.equal(out64, torch.cumsum(b, 0) - b)


@pytest.mark.parametrize("n,bits", [(1, 31), (255, 4), (2049, 9), (70_000, 21), (1_000_003, 31), (300_000, 1)])
def test_own_radix_sort_is_stable_and_descending(n, bits):
    """csrc/prims.hpp: key-value radix sort, descending keys, ties in input order (what orders a row class longest-first for the persistent SpGEMM
    kernels; replaced hipcub::DeviceRadixSort there). Checked against torch's stable sort; inputs must be left untouched."""
    from g4s_amd import capi, host
    lib = capi.load()
    g = torch.Generator(device="cuda").manual_seed(n + bits)
    keys = torch.randint(0, min(1 << bits, (1 << 31) - 1), (n,), dtype=torch.int32, device="cuda", generator=g)
    if n > 1000:
        keys[::7] = keys[0]                                          # many ties
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    k0, v0 = keys.clone(), vals.clone()
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    capi.check(lib.g4s_prim_sort_pairs_desc_i32(keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, bits, host._stream()))
    order = torch.sort(keys.long(), descending=True, stable=True).indices
    assert torch.equal(ko, keys[order]) and torch.equal(vo, vals[order])
    assert torch.equal(keys, k0) and torch.equal(vals, v0)
