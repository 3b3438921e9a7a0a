"""GPU parity: HIP SpGEMM (through the C-ABI) against the oracle's restatement of HashSpGEMM (mm/inc/hash_mult.h).

crpt and ccol must be bit-exact (integer work). cval: |c_gpu − c_oracle| ≤ 1e-10 · Σ|a_ij·b_jk| (fp64 atomics add the
products in a different order than the reference's (j outer, k inner) loop)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import power_law_csr, random_csr, to_scipy

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _abs(A):
    return (A[0], A[1], np.abs(A[2]))


def _check(oracle, A, B, M, K, N, two_phase=False):
    from g4s_amd import host
    a = host.CSR.from_host(*A, M, K)
    b = host.CSR.from_host(*B, K, N)
    c = host.HashSpGEMM(a, b, two_phase=two_phase)
    crpt, ccol, cval = c.to_host()
    orpt, ocol, oval = oracle.spgemm(A, B, N, sort_output=True)
    assert np.array_equal(crpt, orpt), "row pointer differs"
    assert np.array_equal(ccol, ocol), "column ids differ"
    _, _, scale = oracle.spgemm(_abs(A), _abs(B), N, sort_output=True)
    assert np.all(np.abs(cval - oval) <= TOL * scale + 1e-300), f"max err {np.max(np.abs(cval - oval) / (scale + 1e-300))}"
    assert host.get_flop(a, b) == oracle.flop(A[0], A[1], B[0])
    return c


@pytest.mark.parametrize("M,K,N,da,db,seed", [(1, 1, 1, 1.0, 1.0, 0), (4, 4, 4, 0.6, 0.6, 1), (40, 30, 50, 0.15, 0.2, 2),
                                              (300, 300, 300, 0.03, 0.03, 3), (64, 8, 64, 0.9, 0.9, 4), (2000, 1500, 1800, 0.01, 0.01, 5)])
def test_spgemm_random(oracle, M, K, N, da, db, seed):
    A = random_csr(M, K, da, seed, empty_rows=[1] if M > 2 else [])
    B = random_csr(K, N, db, seed + 100, empty_rows=[0] if K > 2 else [])
    _check(oracle, A, B, M, K, N)


@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_short_rows_only_product(oracle, two_phase):
    """Every row at most 512 products (the reference's own examples: can_24, patents_main): no column map, no window splits, no column scratch, the numeric phase of
    the one-call form takes the symbolic phase's row lists; rows of at most 32 products go through the 64-product form of the wave kernel, a row of more than 64
    A-entries through the table kernel behind it. Bit-identical values: these rows are summed in the reference's order."""
    from g4s_amd import host
    rng = np.random.default_rng(71)
    M = K = N = 3000
    lens = rng.integers(0, 5, M)
    lens[[5, 900]] = [20, 100]                                      # 20 × ≤ 5 entries: ≤ 512 products; 100 A-entries: past the wave kernel's 64
    arp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aci = np.concatenate([np.sort(rng.choice(K, n, replace=False)) for n in lens]).astype(np.int32)
    A = (arp, aci, rng.uniform(-1, 1, aci.size))
    blens = rng.integers(0, 5, K)
    brp = np.concatenate([[0], np.cumsum(blens)]).astype(np.int32)
    bci = np.concatenate([np.sort(rng.choice(N, n, replace=False)) for n in blens]).astype(np.int32)
    B = (brp, bci, rng.uniform(-1, 1, bci.size))
    c = _check(oracle, A, B, M, K, N, two_phase=two_phase)
    _, _, oval = oracle.spgemm(A, B, N, sort_output=True)
    assert np.array_equal(c.to_host()[2], oval)


def test_spgemm_golden_fixture(oracle):
    import os
    from g4s_amd import host
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "spgemm_rmat8.npz"))
    n = int(g["n"])
    a = host.CSR.from_host(g["arpt"], g["acol"], g["aval"], n, n)
    c = host.HashSpGEMM(a, a)
    crpt, ccol, cval = c.to_host()
    assert np.array_equal(crpt, g["crpt"]) and np.array_equal(ccol, g["ccol"])
    assert np.allclose(cval, g["cval"], rtol=1e-12, atol=1e-12)


@pytest.fixture(params=["windows", "tables"])
def mid_row_kernels(request, monkeypatch):
    """Mid-size rows go to the bitmap-window kernels while B has <= 4 M columns and to the LDS key tables beyond; "tables" forces
    the latter at test sizes (the library reads the variable on every call)."""
    if request.param == "tables":
        monkeypatch.setenv("G4S_SPGEMM_WINDOW_MAX_N", "0")
    return request.param


def test_spgemm_all_row_classes(oracle, mid_row_kernels):
    """Rows that land in every class: empty, tiny, small, medium, large (optimistic table), overflow → windows, window class."""
    rng = np.random.default_rng(7)
    K, N = 3000, 60000
    # B: 3000 rows × 60000 cols, 100 entries per row (row 0 empty)
    bl = np.full(K, 100)
    bl[0] = 0
    brp = np.concatenate([[0], np.cumsum(bl)]).astype(np.int32)
    bci = np.concatenate([np.sort(rng.choice(N, l, replace=False)) for l in bl]).astype(np.int32)
    bva = rng.uniform(-1, 1, brp[-1])
    # A rows with 0, 1 (→ empty B row), 1, 4, 30, 120, 1000, 2900 entries → flop 0,0,100,400,3000,12000,1e5,2.9e5; plus a 3000-entry row (flop 3e5.. <393216) and duplicates
    lens = [0, 1, 1, 4, 30, 120, 1000, 2900, 2999] + [2] * 50
    rows = []
    for i, l in enumerate(lens):
        if i == 1:
            rows.append(np.array([0]))
        else:
            rows.append(np.sort(rng.choice(np.arange(1, K), l, replace=False)))
    arp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    aci = np.concatenate(rows).astype(np.int32)
    ava = rng.uniform(-1, 1, arp[-1])
    M = len(lens)
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), M, K, N)
    nz = np.diff(c.to_host()[0])
    assert nz[0] == 0 and nz[1] == 0 and nz[6] > 24576 and nz[5] > 4096      # exercised: overflow→hub (symbolic), hub (numeric)


def test_spgemm_flop_hub_path(oracle):
    # flop > 2 M for one row → straight to the HBM bitmap-rank path in symbolic; small N so that distinct ≪ flop
    rng = np.random.default_rng(11)
    K, N = 2000, 9000
    brp = (np.arange(K + 1) * 1100).astype(np.int32)
    bci = np.concatenate([np.sort(rng.choice(N, 1100, replace=False)) for _ in range(K)]).astype(np.int32)
    bva = rng.uniform(0, 1, brp[-1])
    arp = np.array([0, 1950, 1953, 1953], np.int32)
    aci = np.concatenate([np.sort(rng.choice(K, 1950, replace=False)), [3, 7, 9]]).astype(np.int32)
    ava = rng.uniform(0, 1, arp[-1])
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), 3, K, N)
    assert np.diff(c.to_host()[0])[0] > 8000


def test_spgemm_power_law_square(oracle, mid_row_kernels):
    rp, ci, va = power_law_csr(6000, 6000, 23, 1500)
    _check(oracle, (rp, ci, va), (rp, ci, va), 6000, 6000, 6000)


@pytest.fixture(params=["colmap", "plain"])
def column_map(request, monkeypatch):
    """The window kernels renumber B's non-empty columns when at least an eighth of them are empty; "plain" keeps B's ids."""
    if request.param == "plain":
        monkeypatch.setenv("G4S_SPGEMM_NO_COLMAP", "1")
    return request.param


def _three_window_case(oracle, two_phase):
    """B with 2.6 M columns: without the column map the window kernels make three passes per row; rows of 600–9000 products, columns
    bunched at the window seams (2^20, 2^21) and at both ends."""
    rng = np.random.default_rng(29)
    K, N = 400, 2_600_000
    seams = np.array([0, 1 << 20, 1 << 21, N - 64])
    brows = []
    for k in range(K):
        spread = rng.choice(N, 40, replace=False)
        near = (seams[rng.integers(0, 4, 24)] + rng.integers(-40, 64, 24)).clip(0, N - 1)
        brows.append(np.unique(np.concatenate([spread, near])))
    brp = np.concatenate([[0], np.cumsum([len(r) for r in brows])]).astype(np.int32)
    bci = np.concatenate(brows).astype(np.int32)
    bva = rng.uniform(-1, 1, brp[-1])
    lens = [10, 20, 40, 70, 150, 399, 0, 3]
    rows = [np.sort(rng.choice(K, l, replace=False)) for l in lens]
    arp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aci = np.concatenate(rows).astype(np.int32)
    ava = rng.uniform(-1, 1, arp[-1])
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), len(lens), K, N, two_phase=two_phase)
    nz = np.diff(c.to_host()[0])
    assert nz[5] > 8192 and 1024 < nz[2] <= 4096                  # one row needs two value chunks; mid rows use the window kernel


@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_three_windows(oracle, column_map, two_phase):
    """Both column numberings, and both call forms: the one-call form carries the sorted columns (window ids) from the symbolic phase to the
    numeric one; g4s_spgemm_symbolic + g4s_spgemm_numeric makes the numeric kernel mark and emit them itself and translate in place."""
    _three_window_case(oracle, two_phase)


@pytest.mark.parametrize("shape", ["256", "512", "1024"])
@pytest.mark.parametrize("static_rows", [False, True])
def test_spgemm_workgroup_shapes(oracle, monkeypatch, shape, static_rows):
    """Every row class through every workgroup shape of the window kernels (the defaults pick one per class), with the rows handed out
    through the counter and strided."""
    for v in ("SYM_MED", "SYM_LARGE", "SYM_WINDOW", "NUM_MED", "NUM_LARGE", "NUM_M2", "NUM_M3"):
        monkeypatch.setenv("G4S_SPGEMM_T_" + v, shape)
    monkeypatch.setenv("G4S_SPGEMM_M3_CUT", "20000")
    if static_rows:
        monkeypatch.setenv("G4S_SPGEMM_STATIC_ROWS", "1")
    rp, ci, va = power_law_csr(6000, 6000, 23, 1500)
    _check(oracle, (rp, ci, va), (rp, ci, va), 6000, 6000, 6000)
    _three_window_case(oracle, False)


@pytest.mark.parametrize("walk", ["units", "entry_pass"])
@pytest.mark.parametrize("short_rows", ["wave", "tables"])
@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_walk_forms(oracle, monkeypatch, walk, short_rows, two_phase):
    """Round 4: the window kernels walk their products from precomputed unit lists (default) or with the per-chunk entry pass they fall back to when a unit
    table would outgrow its cap (G4S_SPGEMM_NO_UNITS); rows of at most 512 products are merged by one wavefront (default) or go through the hash-table
    kernels that also take what the wavefront kernel hands back (G4S_SPGEMM_NO_WAVE_ROWS). Every combination, both call forms, every row class."""
    if walk == "entry_pass":
        monkeypatch.setenv("G4S_SPGEMM_NO_UNITS", "1")
    if short_rows == "tables":
        monkeypatch.setenv("G4S_SPGEMM_NO_WAVE_ROWS", "1")
    rp, ci, va = power_law_csr(6000, 6000, 23, 1500)
    _check(oracle, (rp, ci, va), (rp, ci, va), 6000, 6000, 6000, two_phase=two_phase)
    _three_window_case(oracle, two_phase)


def _rank_cut_case(stride):
    """B over 800 000 (× stride) columns = three 344 064-column segments of the rank kernel: rows 0 … 799 are blocks of 1 000 consecutive columns, the rest are
    pieces that put a row's outputs exactly on the kernel's cuts. A's rows (all past 8 192 products, so that the symbolic phase writes cuts for them):
      0  exactly 8 192 outputs (one full chunk, no count cut)      1  8 193 outputs (a count cut whose chunk holds one output)
      2  a block straddling the first segment boundary             3  segment 1 empty (two segment starts at the same output)
      4  output 8 192 is the first column of segment 1 (count cut = segment start)
      5  20 equal B rows: 40 000 products for 2 000 outputs (five rounds of products in one chunk)      6  the same on top of 20 000 distinct columns
      7  all three segments, the last one ending at the last column"""
    seg = 344_064
    blocks = [np.arange(1000 * k, 1000 * (k + 1)) for k in range(800)]
    extra = {
        800: np.arange(8000, 8192), 801: np.arange(8192, 8193), 802: np.arange(335_808, 336_000), 803: np.arange(seg, 345_000),
        **{810 + i: np.arange(1000, 3000) for i in range(20)},
    }
    K = 830
    brows = [blocks[k] if k < 800 else extra.get(k, np.zeros(0, dtype=np.int64)) for k in range(K)]
    brp = np.concatenate([[0], np.cumsum([len(r) for r in brows])]).astype(np.int32)
    bci = (np.concatenate(brows) * stride).astype(np.int32)
    rng = np.random.default_rng(57)
    bva = rng.uniform(0.5, 1, brp[-1])
    dup = list(range(810, 830))
    arows = [
        list(range(0, 8)) + [800] + dup[:1],                       # 8 192 outputs … plus one repeated B row so that the row passes 8 192 products
        list(range(0, 8)) + [800, 801] + dup[:1],
        list(range(340, 352)),
        list(range(0, 10)) + list(range(700, 710)),
        [802] + list(range(336, 344)) + [803] + list(range(345, 351)),
        dup,
        list(range(0, 20)) + dup,
        list(range(330, 346)) + list(range(680, 700)) + list(range(790, 800)) + dup[:3],
        [5], [], [700, 810],
    ]
    arp = np.concatenate([[0], np.cumsum([len(r) for r in arows])]).astype(np.int32)
    aci = np.concatenate([np.sort(np.array(r, dtype=np.int32)) for r in arows]).astype(np.int32)
    ava = rng.uniform(0.5, 1, arp[-1])
    return (arp, aci, ava), (brp, bci, bva), len(arows), K, 800_000 * stride


@pytest.mark.parametrize("two_phase", [False, True])
@pytest.mark.parametrize("path", ["rank", "columns"])
@pytest.mark.parametrize("stride", [1, 2])
def test_spgemm_rank_kernel_cuts(oracle, monkeypatch, stride, path, two_phase):
    """Round 5 (spgemm_rank.hpp): the rows past 8 192 products carry cuts instead of columns and take the rank kernel. Outputs placed exactly on its cuts, with B's own
    column ids (stride 1: every column holds an entry, no column map) and with the odd columns empty (stride 2: the map renumbers them to the same compact ids);
    "columns" is the round-4 path on the same input (G4S_SPGEMM_NO_RANK)."""
    if path == "columns":
        monkeypatch.setenv("G4S_SPGEMM_NO_RANK", "1")
    A, B, M, K, N = _rank_cut_case(stride)
    c = _check(oracle, A, B, M, K, N, two_phase=two_phase)
    nz = np.diff(c.to_host()[0])
    assert list(nz[:6]) == [8192, 8193, 12000, 20000, 8192 + 936 + 6000, 2000]


@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_crowded_chunk_streams_its_units_in_batches(oracle, two_phase):
    """One chunk / one window with thousands of 64-entry units: 100 equal B rows of 2 000 entries give 200 000 products for 2 000 outputs — 3 125 units, 195 per
    wavefront, so the rank kernel and the symbolic window kernel stream the units past their register round in three batches of 64 descriptors (stream_unit_groups);
    a second row repeats it on top of 30 000 distinct columns (two chunks, the crowded one not the first)."""
    N, K = 400_000, 160
    hot = np.arange(0, 4000, 2)
    brows = [hot] * 100 + [np.arange(10_000 + 1000 * k, 10_000 + 1000 * (k + 1)) for k in range(30)] + [np.zeros(0, dtype=np.int64)] * 30
    brp = np.concatenate([[0], np.cumsum([len(r) for r in brows])]).astype(np.int32)
    bci = np.concatenate(brows).astype(np.int32)
    rng = np.random.default_rng(63)
    bva = rng.uniform(-1, 1, brp[-1])
    arows = [list(range(100)), list(range(130)), [3, 7, 150], [], list(range(100, 130))]
    arp = np.concatenate([[0], np.cumsum([len(r) for r in arows])]).astype(np.int32)
    aci = np.concatenate([np.array(r, dtype=np.int32) for r in arows]).astype(np.int32)
    ava = rng.uniform(-1, 1, arp[-1])
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), len(arows), K, N, two_phase=two_phase)
    assert list(np.diff(c.to_host()[0])) == [2000, 32000, 2000, 0, 30000]


def test_spgemm_two_call_form_carries_its_columns(oracle, monkeypatch):
    """g4s_spgemm_symbolic leaves the sorted columns, the column map and the window splits of ITS product for the g4s_spgemm_numeric call that follows with the same
    arrays (round 4). Checked: the carried and the uncarried (G4S_SPGEMM_NO_CARRY) numeric calls give the same C; a symbolic call of another product in between
    drops the state (the first product's numeric call then runs without it); a one-call product in between drops it too; a numeric call with other arrays
    does not take it."""
    import ctypes as C
    from g4s_amd import capi, host
    lib = capi.load()
    rpX, ciX, vaX = power_law_csr(6000, 6000, 23, 1500)
    rpY, ciY, vaY = power_law_csr(5000, 5000, 19, 1700)
    X, Y = host.CSR.from_host(rpX, ciX, vaX, 6000, 6000), host.CSR.from_host(rpY, ciY, vaY, 5000, 5000)

    def symbolic(a):
        crpt = torch.empty(a.rows + 1, dtype=torch.int32, device="cuda")
        cnnz = C.c_int64(0)
        capi.check(lib.g4s_spgemm_symbolic(a.rows, a.cols, a.cols, host._ptr(a.rowptr), host._ptr(a.colids), host._ptr(a.rowptr), host._ptr(a.colids), host._ptr(crpt), C.byref(cnnz), None))
        return crpt, cnnz.value

    def numeric(a, crpt, cnnz):
        ccol = torch.empty(cnnz, dtype=torch.int32, device="cuda")
        cval = torch.empty(cnnz, dtype=torch.float64, device="cuda")
        capi.check(lib.g4s_spgemm_numeric(a.rows, a.cols, a.cols, host._ptr(a.rowptr), host._ptr(a.colids), host._ptr(a.values), host._ptr(a.rowptr), host._ptr(a.colids),
                                          host._ptr(a.values), host._ptr(crpt), host._ptr(ccol), host._ptr(cval), capi.DEVICE_POINTERS | capi.SORT_OUTPUT, None))
        return ccol.cpu().numpy(), cval.cpu().numpy()

    def expect(A, n):
        orpt, ocol, oval = oracle.spgemm(A, A, n, sort_output=True)
        _, _, scale = oracle.spgemm(_abs(A), _abs(A), n, sort_output=True)
        return orpt, ocol, oval, scale

    wantX, wantY = expect((rpX, ciX, vaX), 6000), expect((rpY, ciY, vaY), 5000)

    def same(got, crpt, want):
        assert np.array_equal(crpt.cpu().numpy(), want[0]) and np.array_equal(got[0], want[1])
        assert np.all(np.abs(got[1] - want[2]) <= TOL * want[3] + 1e-300)

    cX, nX = symbolic(X)
    same(numeric(X, cX, nX), cX, wantX)                            # carried
    X.values.mul_(2.0)                                             # new values, same pattern: the state serves further numeric calls
    got = numeric(X, cX, nX)
    assert np.array_equal(got[0], wantX[1]) and np.all(np.abs(got[1] - 4.0 * wantX[2]) <= 4.0 * TOL * wantX[3] + 1e-300)
    X.values.mul_(0.5)
    monkeypatch.setenv("G4S_SPGEMM_NO_CARRY", "1")
    cX2, nX2 = symbolic(X)
    same(numeric(X, cX2, nX2), cX2, wantX)                         # not carried
    monkeypatch.delenv("G4S_SPGEMM_NO_CARRY")
    cX, nX = symbolic(X)
    cY, nY = symbolic(Y)                                           # drops X's state
    same(numeric(X, cX, nX), cX, wantX)                            # X without it (Y's stays: other arrays)
    same(numeric(Y, cY, nY), cY, wantY)                            # Y carried
    cX, nX = symbolic(X)
    host.HashSpGEMM(Y, Y)                                          # a one-call product in between drops it
    same(numeric(X, cX, nX), cX, wantX)
    cX, nX = symbolic(X)
    Xb = host.CSR.from_host(rpX, ciX, vaX, 6000, 6000)             # the same matrix in OTHER arrays: the key does not match
    cXb, nXb = torch.empty_like(cX), nX
    cXb.copy_(cX)
    same(numeric(Xb, cXb, nXb), cXb, wantX)
    # the SAME buffers refilled with ANOTHER pattern (equal shape and entry counts: every row's columns shifted by one, cyclically) and the caller's own
    # crpt — pointers, M, K, N all match the carried state, the index arrays do not: the state must not be used (it would apply X's cuts and columns to Z's crpt)
    cX, nX = symbolic(X)
    ciZ = ciX.copy()
    for r in range(6000):
        ciZ[rpX[r]:rpX[r + 1]] = np.sort((ciX[rpX[r]:rpX[r + 1]].astype(np.int64) + 1) % 6000)
    wantZ = expect((rpX, ciZ, vaX), 6000)
    X.colids.copy_(torch.from_numpy(ciZ))                          # in place: X.colids keeps its address
    cZ = cX                                                        # … and so does crpt
    cZ.copy_(torch.from_numpy(wantZ[0]))
    same(numeric(X, cZ, int(wantZ[0][-1])), cZ, wantZ)
    capi.check(lib.g4s_trim())                                     # releases what the last symbolic call still holds


def test_spgemm_short_rows_by_one_wavefront(oracle):
    """spgemm_small_wave_kernel on the cases its merge has to get right: a row whose runs share columns (sums in run order), duplicate columns INSIDE a B row
    (an unmerged CSR, CSR.h:485-669 keeps duplicates), empty B rows among the entries, a row of exactly 512 products, a row of 70 entries (more than the 64 lanes:
    handed to the table kernel), a row of 600 products whose output is shorter than 512 (numeric classes are cut by output length: handed back too). The sums of
    these rows are added in the reference's (j outer, k inner) order: bit for bit equal to the oracle's, not only within the tolerance."""
    from g4s_amd import host
    rng = np.random.default_rng(77)
    K, N = 400, 5000
    blen = rng.integers(1, 9, K)
    blen[:10] = 0                                                  # B rows 0-9 empty
    blen[10] = 512; blen[11] = 300; blen[12] = 300; blen[13:16] = 200
    brp = np.concatenate([[0], np.cumsum(blen)]).astype(np.int32)
    bci = np.concatenate([np.sort(rng.choice(N, l, replace=False)) if l else np.zeros(0, np.int64) for l in blen]).astype(np.int32)
    # duplicates inside B row 20 (a repeated column, adjacent after sorting)
    bci[brp[20]:brp[21]] = np.sort(np.resize(bci[brp[20]:brp[21]][:max(1, blen[20] // 2)], blen[20]))
    bci[brp[12]:brp[13]] = bci[brp[11]:brp[12]]                    # rows 11 and 12 hold the same columns: every product of row "11+12" is a duplicate
    bva = rng.integers(-8, 9, brp[-1]).astype(np.float64) + rng.uniform(-1, 1, brp[-1])
    rows = [np.array([10]),                                        # exactly 512 products
            np.array([11, 12]),                                    # 600 products, 300 outputs
            np.array([0, 3, 20, 25]),                              # empty B rows among the entries, duplicates inside a run
            np.sort(rng.choice(np.arange(16, K), 70, replace=False)),   # 70 entries
            np.array([13, 14, 15]),                                # 600 products, overlapping columns by chance only
            np.array([5]),                                         # only an empty B row
            np.arange(20, 60)]                                     # 40 short runs
    arp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    aci = np.concatenate(rows).astype(np.int32)
    ava = rng.uniform(-2, 2, arp[-1])
    M = len(rows)
    for two_phase in (False, True):
        c = _check(oracle, (arp, aci, ava), (brp, bci, bva), M, K, N, two_phase=two_phase)
        crpt, ccol, cval = c.to_host()
        orpt, ocol, oval = oracle.spgemm((arp, aci, ava), (brp, bci, bva), N, sort_output=True)
        flop = np.array([sum(blen[c_] for c_ in r) for r in rows])
        for i in range(M):
            if flop[i] <= 512 and len(rows[i]) <= 64:
                assert np.array_equal(cval[crpt[i]:crpt[i + 1]], oval[orpt[i]:orpt[i + 1]]), f"row {i} is not bit-identical"
        assert crpt[2] - crpt[1] == 300 and crpt[6] == crpt[5]


@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_row_of_300k_outputs(oracle, column_map, two_phase):
    """Output rows of ≈ 150 K and ≈ 142 K entries (19 value chunks; past the 131 072 a list item of the emit step once packed) next to
    ordinary rows: the numeric big-row kernel takes them, the symbolic window kernel counts them (flop 1.65 M < the 2 M of the hub class)."""
    rng = np.random.default_rng(41)
    K, N = 2000, 300_000
    brp = (np.arange(K + 1) * 1100).astype(np.int32)
    bci = np.concatenate([2 * np.sort(rng.choice(N // 2, 1100, replace=False)) for _ in range(K)]).astype(np.int32)   # odd columns stay empty
    bva = rng.uniform(0.5, 1, brp[-1])
    lens = [1500, 3, 40, 0, 400]
    rows = [np.sort(rng.choice(K, l, replace=False)) for l in lens]
    arp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aci = np.concatenate(rows).astype(np.int32)
    ava = rng.uniform(0.5, 1, arp[-1])
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), len(lens), K, N, two_phase=two_phase)
    nz = np.diff(c.to_host()[0])
    assert nz[0] > 131_072 and nz[4] > 131_072


def test_spgemm_row_past_a_million_outputs(oracle):
    """An output row of 1.1 M entries (B rows with disjoint column blocks): past the big-row kernel's class (1 M), so the numeric phase takes
    the HBM-bitmap hub path for it, next to rows of the window and table classes."""
    rng = np.random.default_rng(43)
    K, per, N = 1200, 1000, 1_200_000
    brp = (np.arange(K + 1) * per).astype(np.int32)
    bci = np.arange(K * per, dtype=np.int32)                      # row k holds columns [1000 k, 1000 (k + 1))
    bva = rng.uniform(0.5, 1, K * per)
    lens = [1100, 3, 0, 40]
    rows = [np.sort(rng.choice(K, l, replace=False)) for l in lens]
    arp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aci = np.concatenate(rows).astype(np.int32)
    ava = rng.uniform(0.5, 1, arp[-1])
    c = _check(oracle, (arp, aci, ava), (brp, bci, bva), len(lens), K, N)
    nz = np.diff(c.to_host()[0])
    assert nz[0] == 1_100_000 and nz[3] == 40_000


def test_spgemm_raw_pointer_host_call(oracle, g4s):
    """The mkl(...)-shaped entry point (mm/inc/mkl_mult.h:40-43): host arrays in, callee-allocated host arrays out, 7 stage timings."""
    from g4s_amd import capi
    A = random_csr(500, 400, 0.02, 31)
    B = random_csr(400, 450, 0.02, 32)
    crpt, ccol, cval = C.c_void_p(), C.c_void_p(), C.c_void_p()
    cnnz = C.c_int64()
    t = capi.Timings()
    p = lambda a: a.ctypes.data
    capi.check(g4s.g4s_spgemm_csr_i32_f64(p(A[0]), p(A[1]), p(A[2]), p(B[0]), p(B[1]), p(B[2]), C.byref(crpt), C.byref(ccol), C.byref(cval),
                                          500, 400, 450, C.byref(cnnz), C.byref(t), capi.SORT_OUTPUT))
    orpt, ocol, oval = oracle.spgemm(A, B, 450)
    n = cnnz.value
    assert n == orpt[-1]
    got_rpt = np.ctypeslib.as_array(C.cast(crpt, C.POINTER(C.c_int32)), (501,)).copy()
    got_col = np.ctypeslib.as_array(C.cast(ccol, C.POINTER(C.c_int32)), (n,)).copy()
    got_val = np.ctypeslib.as_array(C.cast(cval, C.POINTER(C.c_double)), (n,)).copy()
    for ptr in (crpt, ccol, cval):
        g4s.g4s_free(ptr)                                   # allocator pairing (utility.h:126-153)
    assert np.array_equal(got_rpt, orpt) and np.array_equal(got_col, ocol) and np.allclose(got_val, oval, rtol=1e-12, atol=1e-13)
    assert t.total >= t.spmm > 0 and t.create > 0 and t.export_csr > 0
    # flop, host-pointer form
    flop = C.c_int64()
    capi.check(g4s.g4s_spgemm_flop(500, p(A[0]), p(A[1]), p(B[0]), C.byref(flop), None, 0))
    assert flop.value == oracle.flop(A[0], A[1], B[0])


def test_spgemm_rejects_bad_ids(g4s):
    from g4s_amd import capi, host
    A = host.CSR.from_host(np.array([0, 1], np.int32), np.array([5], np.int32), np.ones(1), 1, 6)
    B = host.CSR.from_host(np.array([0, 1, 1], np.int32), np.array([0], np.int32), np.ones(1), 2, 2)   # only 2 rows: A's column 5 is out of range
    crpt = torch.empty(2, dtype=torch.int32, device="cuda")
    cnnz = C.c_int64()
    st = g4s.g4s_spgemm_symbolic(1, 2, 2, A.rowptr.data_ptr(), A.colids.data_ptr(), B.rowptr.data_ptr(), B.colids.data_ptr(), crpt.data_ptr(),
                                 C.byref(cnnz), None)
    assert st == capi.ERR_INVALID


def test_spgemm_rmat17_bit_exact_against_oracle_and_scipy(oracle):
    """A 2^17-row R-MAT A·A (the structure of BASELINE config 3 at a size the oracle multiplies in seconds): crpt and ccol bit for bit
    against the oracle AND against scipy's independent implementation; values within 1e-10 · Σ|terms|."""
    from g4s_amd import host
    n = 1 << 17
    A = host.rmat_csr(n, 17, 3 * n, 20240522)
    A.values.abs_()
    Ah = A.to_host()
    Cm = host.HashSpGEMM(A, A)
    crpt, ccol, cval = Cm.to_host()
    orpt, ocol, oval = oracle.spgemm(Ah, Ah, n, sort_output=True)
    assert np.array_equal(crpt, orpt) and np.array_equal(ccol, ocol)
    assert np.all(np.abs(cval - oval) <= TOL * oval + 1e-300)          # values are positive: Σ|terms| = the sum itself
    S = to_scipy(*Ah, n, n)
    S2 = (S @ S).tocsr()
    S2.sort_indices()
    assert np.array_equal(S2.indptr, crpt) and np.array_equal(S2.indices, ccol)
    assert np.all(np.abs(S2.data - cval) <= TOL * oval + 1e-300)
    assert host.get_flop(A, A) == oracle.flop(Ah[0], Ah[1], Ah[0])


@pytest.mark.parametrize("ef", [2, 3])
def test_spgemm_full_size_properties(ef):
    """BASELINE config 3: R-MAT scale 21 (n = 2 097 152), C = A·A, at edge factor 2 (nnz(C) ≈ 0.96e9) and at edge factor 3 — the one
    every reported number uses (nnz(C) 1.94e9, the largest that still fits the reference's int32 row pointer; EF 8 gives 9.56e9 and must
    be refused). Size-independent checks: (A·A)·x == A·(A·x), rows strictly sorted, crpt consistent, integer part reproducible."""
    import ctypes as C
    from g4s_amd import capi, host
    n = 1 << 21
    if ef == 2:
        A8 = host.rmat_csr(n, 21, 8 * n, 20240522)
        crpt = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        cnnz = C.c_int64()
        st = capi.load().g4s_spgemm_symbolic(n, n, n, A8.rowptr.data_ptr(), A8.colids.data_ptr(), A8.rowptr.data_ptr(), A8.colids.data_ptr(),
                                             crpt.data_ptr(), C.byref(cnnz), None)
        assert st == capi.ERR_OVERFLOW and cnnz.value > 2 ** 31      # int32 crpt of the reference (define.h:14) cannot hold it
        del A8, crpt
    A = host.rmat_csr(n, 21, ef * n, 20240522)
    A.values.abs_()                                            # values U(0,1): no cancellation in the check
    Cm = host.HashSpGEMM(A, A)
    assert Cm.rows == n and Cm.nnz == int(Cm.rowptr[-1].item()) and int(Cm.rowptr[0].item()) == 0
    assert bool(torch.all(Cm.rowptr[1:] >= Cm.rowptr[:-1]))
    # strictly ascending columns inside every row
    d = Cm.colids[1:].long() - Cm.colids[:-1].long()
    row_start = torch.zeros(Cm.nnz, dtype=torch.bool, device="cuda")
    starts = Cm.rowptr[:-1][(Cm.rowptr[1:] > Cm.rowptr[:-1])].long()
    row_start[starts] = True
    assert bool(torch.all((d > 0) | row_start[1:]))
    x = host.synth_vector(3, n).abs_()
    lhs = Cm.spmv(x)
    rhs = A.spmv(A.spmv(x))
    assert bool(torch.all((lhs - rhs).abs() <= 1e-10 * rhs.abs() + 1e-300))
    flop = host.get_flop(A, A)
    assert flop >= Cm.nnz
    C2 = host.HashSpGEMM(A, A)
    assert torch.equal(C2.rowptr, Cm.rowptr) and torch.equal(C2.colids, Cm.colids)
    print(f"RMAT-21 A·A (EF {ef}): nnz(A)={A.nnz} flop={flop} nnz(C)={Cm.nnz} compression={flop / Cm.nnz:.2f}")
    del Cm, C2, A
    capi.check(capi.load().g4s_trim())                           # 25 GB of outputs go back to the driver before the next test


def test_spgemm_randomised_seams():
    """tools/stress_spgemm.py, 25 cases: A rows of exactly T / T + 1 / 2T entries, B rows of 63 … 129 entries, empty B rows, column counts around the
    column map's threshold, every workgroup shape, both call forms — against scipy (index arrays bit for bit, values to 1e-10)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_spgemm.py"), "--cases", "25", "--seed", "5"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _shuffle_rows(rp, ci, va, seed):
    """the same matrix with every row's entries in random order"""
    rng = np.random.default_rng(seed)
    ci2, va2 = ci.copy(), va.copy()
    for r in range(len(rp) - 1):
        p = rng.permutation(rp[r + 1] - rp[r]) + rp[r]
        ci2[rp[r]:rp[r + 1]], va2[rp[r]:rp[r + 1]] = ci[p], va[p]
    return rp, ci2, va2


@pytest.mark.parametrize("two_phase", [False, True])
def test_spgemm_takes_unsorted_rows_of_b(oracle, two_phase):
    """Round 5 (missing 1): HashSpGEMM takes any row order — its hash traversal never looks at it (mm/inc/hash_mult.h:579-600) — and so does the
    drop-in now: when the sortedness check fires, the product runs on a private sorted copy of B's rows. Random unsorted B in every row class, both call forms;
    index arrays bit for bit against the oracle (which walks the unsorted rows as the reference does), the caller's arrays untouched."""
    from g4s_amd import host
    rp, ci, va = power_law_csr(6000, 6000, 23, 1500)
    B = _shuffle_rows(rp, ci, va, 11)
    assert not np.array_equal(B[1], ci)
    b = host.CSR.from_host(*B, 6000, 6000)
    keep = b.colids.clone(), b.values.clone()
    _check(oracle, (rp, ci, va), B, 6000, 6000, 6000, two_phase=two_phase)
    a = host.CSR.from_host(rp, ci, va, 6000, 6000)
    c = host.HashSpGEMM(a, b, two_phase=two_phase)
    assert torch.equal(b.colids, keep[0]) and torch.equal(b.values, keep[1]), "the caller's B was modified"
    cs = host.HashSpGEMM(a, a, two_phase=two_phase)                 # the sorted B gives the same C
    assert torch.equal(c.rowptr, cs.rowptr) and torch.equal(c.colids, cs.colids)
    assert torch.allclose(c.values, cs.values, rtol=1e-10, atol=1e-11)   # (sums of up to 1 500 signed products in another order; _check above holds the values to the oracle's)
    # duplicates inside unsorted rows, empty rows, a single descent
    rp2, ci2, va2 = random_csr(300, 300, 0.05, 9)
    bad = ci2.copy()
    r = int(np.argmax(np.diff(rp2) >= 2))
    bad[rp2[r]], bad[rp2[r] + 1] = bad[rp2[r] + 1], bad[rp2[r]]
    _check(oracle, (rp2, ci2, va2), (rp2, bad, va2), 300, 300, 300, two_phase=two_phase)
    dup = ci2.copy()
    dup[rp2[r]] = dup[rp2[r] + 1]                                   # a repeated column in an unsorted row
    _check(oracle, (rp2, ci2, va2), _shuffle_rows(rp2, dup, va2, 5), 300, 300, 300, two_phase=two_phase)


def test_spgemm_chained_product_with_unsorted_intermediate(oracle):
    """C = HashSpGEMM<false, false>(A, A) — table order, unsorted rows (hash_mult.h:530-551) — fed back in as B of the next product, as a chained product does in
    the reference: A·C through the drop-in == the oracle's A·C on the same unsorted C, index arrays bit for bit."""
    rp, ci, va = power_law_csr(3000, 3000, 31, 800)
    A = (rp, ci, np.abs(va))
    crpt, ccol, cval = oracle.spgemm(A, A, 3000, sort_output=False)
    srpt, scol, sval = oracle.spgemm(A, A, 3000, sort_output=True)
    assert not np.array_equal(ccol, scol), "the oracle's table order happens to be sorted: the case tests nothing"
    _check(oracle, A, (crpt, ccol, cval), 3000, 3000, 3000)
    _check(oracle, A, (crpt, ccol, cval), 3000, 3000, 3000, two_phase=True)


def test_spgemm_numeric_only_call_sorts_b_itself(oracle):
    """g4s_spgemm_numeric without a symbolic call before it (the caller brings its own crpt), B unsorted: the numeric phase checks and sorts B itself."""
    from g4s_amd import capi, host
    lib = capi.load()
    rp, ci, va = power_law_csr(4000, 4000, 7, 900)
    B = _shuffle_rows(rp, ci, va, 3)
    orpt, ocol, oval = oracle.spgemm((rp, ci, va), B, 4000)
    a = host.CSR.from_host(rp, ci, va, 4000, 4000)
    b = host.CSR.from_host(*B, 4000, 4000)
    crpt = torch.from_numpy(orpt).cuda()
    ccol = torch.empty(len(ocol), dtype=torch.int32, device="cuda")
    cval = torch.empty(len(ocol), dtype=torch.float64, device="cuda")
    capi.check(lib.g4s_trim())                                      # nothing carried from an earlier product
    capi.check(lib.g4s_spgemm_numeric(4000, 4000, 4000, a.rowptr.data_ptr(), a.colids.data_ptr(), a.values.data_ptr(), b.rowptr.data_ptr(), b.colids.data_ptr(),
                                      b.values.data_ptr(), crpt.data_ptr(), ccol.data_ptr(), cval.data_ptr(), capi.DEVICE_POINTERS | capi.SORT_OUTPUT, None))
    torch.cuda.synchronize()
    assert np.array_equal(ccol.cpu().numpy(), ocol)
    _, _, scale = oracle.spgemm(_abs((rp, ci, va)), _abs(B), 4000)
    assert np.all(np.abs(cval.cpu().numpy() - oval) <= TOL * scale + 1e-300)
