"""Shared input builders for the tests (numpy/scipy only)."""
import numpy as np
import scipy.sparse as sp


def random_csr(rows, cols, density, seed, empty_rows=(), dense_rows=()):
    rng = np.random.default_rng(seed)
    m = sp.random(rows, cols, density=density, format="lil", random_state=rng, data_rvs=lambda k: rng.uniform(-1, 1, k))
    for r in empty_rows:
        m.rows[r] = []
        m.data[r] = []
    for r in dense_rows:
        m[r, :] = rng.uniform(-1, 1, cols)
    m = m.tocsr()
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float64)


def to_scipy(rowptr, colids, values, rows, cols):
    return sp.csr_matrix((values, colids, rowptr), shape=(rows, cols))


def power_law_csr(rows, cols, seed, max_len):
    """Row lengths ~ Zipf, clipped: mixes empty rows, short rows and a few hubs."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(1.6, rows) - 1, max_len).astype(np.int64)
    lens = np.minimum(lens, cols)
    rowptr = np.zeros(rows + 1, np.int64)
    rowptr[1:] = np.cumsum(lens)
    colids = np.empty(rowptr[-1], np.int32)
    for r in np.nonzero(lens)[0]:
        k = int(lens[r])
        if 4 * k < cols:                      # rejection sampling: O(k), not the O(cols) permutation of rng.choice
            u = np.unique(rng.integers(0, cols, size=k + k // 2 + 8))
            while len(u) < k:
                u = np.unique(np.concatenate([u, rng.integers(0, cols, size=k)]))
            u = u[np.sort(rng.permutation(len(u))[:k])]
        else:
            u = np.sort(rng.choice(cols, k, replace=False))
        colids[rowptr[r]:rowptr[r + 1]] = u
    values = rng.uniform(-1, 1, rowptr[-1])
    return rowptr.astype(np.int32), colids, values


def hex_mesh(ex, ey, ez):
    """Structured hexahedral mesh: returns (ien[nel,8], id[nno,3], nno, neq) with CitcomS's numbering rule
    eqn = 3·node + d (citcoms/lib/Construct_arrays.c), 0-based."""
    nx, ny, nz = ex + 1, ey + 1, ez + 1
    nno = nx * ny * nz
    node = lambda i, j, k: (k * ny + j) * nx + i
    ien = []
    for k in range(ez):
        for j in range(ey):
            for i in range(ex):
                ien.append([node(i, j, k), node(i + 1, j, k), node(i + 1, j + 1, k), node(i, j + 1, k),
                            node(i, j, k + 1), node(i + 1, j, k + 1), node(i + 1, j + 1, k + 1), node(i, j + 1, k + 1)])
    ien = np.array(ien, np.int32)
    idmap = (3 * np.arange(nno)[:, None] + np.arange(3)[None, :]).astype(np.int32)
    return ien, idmap, nno, 3 * nno


def spd_blocks(nel, n, seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, (nel, n, n))
    k = a @ a.transpose(0, 2, 1) + n * np.eye(n)[None]
    return np.ascontiguousarray(k.reshape(nel, n * n))


def stokes_problem(ex, ey, ez, seed):
    """A synthetic saddle-point problem on the hexahedral mesh of hex_mesh: SPD 24×24 element stiffness blocks, a random per-element
    divergence vector g (elt_del[e].g[p][0]), boundary equations on a ninth of the nodes, positive node masses and element volumes
    (the weights of CitcomS's global norms). Sizes follow SURVEY.md §8d C5; the data are seeded, not physical."""
    ien, idmap, nno, neq = hex_mesh(ex, ey, ez)
    nel = len(ien)
    rng = np.random.default_rng(1000 + seed)
    K = spd_blocks(nel, 24, seed)
    g = rng.uniform(-1, 1, (nel, 24))
    bc = np.array(sorted(set(idmap[rng.choice(nno, max(1, nno // 9), replace=False)].ravel().tolist())), np.int32)
    F = rng.uniform(-1, 1, neq)
    F[bc] = 0.0
    nmass = rng.uniform(0.5, 1.5, nno)
    area = rng.uniform(0.5, 1.5, nel)
    return {"ien": ien, "id": idmap, "nno": nno, "neq": neq, "K": K, "g": np.ascontiguousarray(g), "bc": bc, "F": F, "nmass": nmass, "area": area,
            "volume": float(area.sum())}


def hex_node_map(ex, ey, ez, idmap):
    """CitcomS's Node_map for the mesh of hex_mesh (construct_node_maps, citcoms/lib/Construct_arrays.c:254-328): 14·3 slots per
    node — the node's own three equations, then those of every lower-numbered neighbour of the 3×3×3 stencil — unused slots = neq."""
    nx, ny, nz = ex + 1, ey + 1, ez + 1
    nno, neq, max_eqn = nx * ny * nz, 3 * nx * ny * nz, 42
    nm = np.full((nno, max_eqn), neq, np.int32)
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                nn = (k * ny + j) * nx + i
                nm[nn, 0:3] = idmap[nn]
                ia = 0
                for dk in (-1, 0, 1):
                    for dj in (-1, 0, 1):
                        for di in (-1, 0, 1):
                            ii, jj, kk = i + di, j + dj, k + dk
                            if not (0 <= ii < nx and 0 <= jj < ny and 0 <= kk < nz):
                                continue
                            ja = (kk * ny + jj) * nx + ii
                            if ja < nn:
                                ia += 1
                                nm[nn, 3 * ia:3 * ia + 3] = idmap[ja]
    return nm, max_eqn


def assemble_csr(ien, idmap, K, neq):
    """The element matrices summed into one CSR matrix (scipy): rowptr, colids, values with sorted, duplicate-free rows."""
    import scipy.sparse as sp
    n = K.shape[1]
    m = int(round(n ** 0.5))
    eq = idmap[ien].reshape(len(ien), m)
    A = sp.coo_matrix((K.ravel(), (np.repeat(eq, m, axis=1).ravel(), np.tile(eq, (1, m)).ravel())), shape=(neq, neq)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def init_gloo(rank, world, rendezvous_file):
    """Process group for the spawned multi-rank tests through a FILE store: a TCP port picked beforehand can be taken by somebody else before
    rank 0 binds it (seen once on a GPU box: EADDRINUSE); a file in the test's tmp_path cannot."""
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"file://{rendezvous_file}", rank=rank, world_size=world)
