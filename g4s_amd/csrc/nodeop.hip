// nodeop.hip — CitcomS's node-assembled stiffness operator: the stored symmetric half `Node_map / Eqn_k1..3`
// (citcoms/lib/Construct_arrays.c:254-328, 335-470) and its mat-vec `n_assemble_del2_u` (Element_calculations.c:516-577).
//
// The reference walks the nodes and, for each stored coefficient, adds one product to the node's own equations and one to the
// neighbour's (`Au[C[i]] += …`): a scatter, i.e. a race on a GPU. At create time the half is therefore expanded once, on the host,
// into the full operator in block form — per node up to 27 neighbour slots (13 lower-numbered, itself, 13 higher-numbered), each a
// 3×3 block — and the mat-vec becomes a gather: 32 lanes per node, lane s multiplies block s by the neighbour's three unknowns,
// a fixed shuffle tree adds the 27 partial 3-vectors. No atomics, reproducible, and 1 944 B per node instead of the 4 608 B per
// element (≈ per node) the element-by-element form streams.
#include "common.hpp"
#include <memory>
#include <vector>

namespace {

constexpr int kSlots = 27;        // neighbour slots per node in the expanded form
constexpr int kLanes = 32;        // lanes per node (27 active)

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        hipError_t e = g4s::device_malloc(&p, n);
        if (e != hipSuccess) return g4s::set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        return G4S_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Au[eq(node, d)] = Σ_s Σ_c block[node][s][d][c] · u[eq(nbr[node][s], c)]
__global__ __launch_bounds__(256) void node_blocks_matvec_kernel(int nno, const int *__restrict__ nbr, const double *__restrict__ blocks,
                                                                  const int *__restrict__ node_eq, const double *__restrict__ u,
                                                                  double *__restrict__ Au, const int *__restrict__ skip)
{
    if (skip && *skip) return;
    const int lane = threadIdx.x & (kLanes - 1);
    const int node = (blockIdx.x * 256 + threadIdx.x) / kLanes;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (node < nno && lane < kSlots) {
        const size_t slot = (size_t)node * kSlots + lane;
        const int nb = nbr[slot];
        const double *b = blocks + slot * 9;
        const double b00 = b[0], b01 = b[1], b02 = b[2], b10 = b[3], b11 = b[4], b12 = b[5], b20 = b[6], b21 = b[7], b22 = b[8];
        const double u0 = u[node_eq[nb * 3]], u1 = u[node_eq[nb * 3 + 1]], u2 = u[node_eq[nb * 3 + 2]];
        a0 = (b00 * u0 + b01 * u1) + b02 * u2;
        a1 = (b10 * u0 + b11 * u1) + b12 * u2;
        a2 = (b20 * u0 + b21 * u1) + b22 * u2;
    }
#pragma unroll
    for (int off = kLanes / 2; off > 0; off >>= 1) {
        a0 += __shfl_down(a0, off, kLanes);
        a1 += __shfl_down(a1, off, kLanes);
        a2 += __shfl_down(a2, off, kLanes);
    }
    if (node < nno && lane == 0) {
        Au[node_eq[node * 3]] = a0;
        Au[node_eq[node * 3 + 1]] = a1;
        Au[node_eq[node * 3 + 2]] = a2;
    }
}

__global__ __launch_bounds__(256) void node_zero_rows_kernel(int n_zero, const int *__restrict__ rows, double *__restrict__ v, const int *__restrict__ skip)
{
    if (skip && *skip) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_zero) v[rows[i]] = 0.0;
}

} // namespace

struct g4s_node_op_s {
    int nno = 0, neq = 0;
    DevBuf nbr, blocks, node_eq;
};

G4S_API g4s_status g4s_node_op_create(g4s_node_op_t *out, int32_t nno, int32_t neq, int32_t max_eqn, const int32_t *node_map,
                                      const int32_t *id, const double *eqn_k1, const double *eqn_k2, const double *eqn_k3)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_REQUIRE(nno >= 0 && neq >= 0 && max_eqn >= 3 && max_eqn % 3 == 0, "bad sizes (max_eqn = 14·3 in CitcomS)");
    G4S_REQUIRE(nno == 0 || (node_map && id && eqn_k1 && eqn_k2 && eqn_k3), "NULL argument");
    std::vector<int> eq_node((size_t)neq, -1);
    for (int64_t k = 0; k < (int64_t)nno * 3; ++k) {
        const int eqn = id[k];
        if (eqn < 0 || eqn >= neq) return g4s::set_error(G4S_ERR_INVALID, "g4s_node_op_create: id[%lld] = %d outside [0,%d)", (long long)k, eqn, neq);
        if (eq_node[eqn] >= 0) return g4s::set_error(G4S_ERR_INVALID, "g4s_node_op_create: equation %d is owned by two (node, dof) pairs", eqn);
        eq_node[eqn] = (int)(k / 3);
    }
    std::vector<int> nbr((size_t)nno * kSlots), used((size_t)nno, 0);
    std::vector<double> blocks((size_t)nno * kSlots * 9, 0.0);
    for (int n = 0; n < nno; ++n)
        for (int s = 0; s < kSlots; ++s) nbr[(size_t)n * kSlots + s] = n;       // padding slots: the node itself with a zero block
    auto block_of = [&](int row_node, int col_node) -> double * {
        int *list = &nbr[(size_t)row_node * kSlots];
        for (int s = 0; s < used[row_node]; ++s)
            if (list[s] == col_node) return &blocks[((size_t)row_node * kSlots + s) * 9];
        if (used[row_node] == kSlots) return nullptr;
        list[used[row_node]] = col_node;
        return &blocks[((size_t)row_node * kSlots + used[row_node]++) * 9];
    };
    const double *B[3] = {eqn_k1, eqn_k2, eqn_k3};
    for (int e = 0; e < nno; ++e) {
        const int32_t *C = node_map + (size_t)e * max_eqn;
        for (int k = 0; k < max_eqn / 3; ++k) {
            const int c0 = C[3 * k];
            if (c0 == neq && C[3 * k + 1] == neq && C[3 * k + 2] == neq) continue;            // unused slot group (dummy equation)
            if (c0 < 0 || c0 >= neq) return g4s::set_error(G4S_ERR_INVALID, "g4s_node_op_create: Node_map[%d][%d] = %d outside [0,%d]", e, 3 * k, c0, neq);
            const int nb = eq_node[c0];
            if (nb < 0 || id[nb * 3] != c0 || id[nb * 3 + 1] != C[3 * k + 1] || id[nb * 3 + 2] != C[3 * k + 2])
                return g4s::set_error(G4S_ERR_UNSUPPORTED, "g4s_node_op_create: slot group %d of node %d is not the three equations of one node", k, e);
            if (k == 0 && nb != e) return g4s::set_error(G4S_ERR_INVALID, "g4s_node_op_create: slot group 0 of node %d is not the node itself", e);
            double *fwd = k >= 1 ? block_of(e, nb) : nullptr;      // Au[eqn_d(e)] += B_d[i]·u[C[i]], i >= 3   (:556-561)
            double *tr = block_of(nb, e);                          // Au[C[i]] += B1[i]·U1 + B2[i]·U2 + B3[i]·U3, all i   (:562-563)
            if ((k >= 1 && !fwd) || !tr) return g4s::set_error(G4S_ERR_UNSUPPORTED, "g4s_node_op_create: a node has more than %d neighbours", kSlots);
            for (int d = 0; d < 3; ++d)
                for (int c = 0; c < 3; ++c) {
                    const double v = B[d][(size_t)e * max_eqn + 3 * k + c];
                    if (fwd) fwd[d * 3 + c] += v;
                    tr[c * 3 + d] += v;
                }
        }
    }
    auto op = std::make_unique<g4s_node_op_s>();
    op->nno = nno; op->neq = neq;
    G4S_TRY(op->nbr.alloc(sizeof(int) * nbr.size()));
    G4S_TRY(op->blocks.alloc(sizeof(double) * blocks.size()));
    G4S_TRY(op->node_eq.alloc(sizeof(int) * (size_t)nno * 3));
    if (nno) {
        G4S_HIP_TRY(hipMemcpy(op->nbr.p, nbr.data(), sizeof(int) * nbr.size(), hipMemcpyHostToDevice));
        G4S_HIP_TRY(hipMemcpy(op->blocks.p, blocks.data(), sizeof(double) * blocks.size(), hipMemcpyHostToDevice));
        G4S_HIP_TRY(hipMemcpy(op->node_eq.p, id, sizeof(int) * (size_t)nno * 3, hipMemcpyHostToDevice));
    }
    *out = op.release();
    return G4S_OK;
}

G4S_API g4s_status g4s_node_op_destroy(g4s_node_op_t op)
{
    delete op;
    return G4S_OK;
}

// internal: the mat-vec that returns at once when *skip_dev != 0 (cg.hip enqueues iterations ahead of the termination test)
int g4s_node_op_apply_unless(g4s_node_op_t op, const double *u_dev, double *Au_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                             const int *skip_dev, void *stream)
{
    G4S_REQUIRE(op && u_dev && Au_dev, "NULL argument");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid_dev), "zero_resid is NULL");
    hipStream_t s = g4s::as_stream(stream);
    if (op->neq && (int64_t)op->nno * 3 != op->neq) G4S_HIP_TRY(hipMemsetAsync(Au_dev, 0, sizeof(double) * (size_t)op->neq, s));
    if (op->nno) {
        const int64_t threads = (int64_t)op->nno * kLanes;
        hipLaunchKernelGGL(node_blocks_matvec_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, op->nno, op->nbr.as<int>(), op->blocks.as<double>(),
                           op->node_eq.as<int>(), u_dev, Au_dev, skip_dev);
    }
    if (n_zero) hipLaunchKernelGGL(node_zero_rows_kernel, dim3((n_zero + 255) / 256), dim3(256), 0, s, n_zero, zero_resid_dev, Au_dev, skip_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_node_op_apply(g4s_node_op_t op, const double *u_dev, double *Au_dev, const int32_t *zero_resid_dev, int32_t n_zero, void *stream)
{
    return g4s_node_op_apply_unless(op, u_dev, Au_dev, zero_resid_dev, n_zero, nullptr, stream);
}
int g4s_node_op_neq(g4s_node_op_t op) { return op ? op->neq : 0; }
