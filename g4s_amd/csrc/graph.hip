// graph.hip — the vertex-centric gather/apply graph interface (B3 of include/g4s.h) on gfx950.
//
// Reference contract (deepmd/source/op/graph.h:21-32; citcoms/lib/global_defs.h:48-49,854-857):
//     for vi in [0,numNodes): for nb in [0,degree): gather(vi, nb, …);  apply(vi, …)
// with host callbacks. Host code cannot run on the GPU, so the three gather/apply pairs that exist in the reference are
// recognised as PATTERNS (registered with g4s_register_pattern) and executed by hand-written kernels:
//   ELEMENT_BLOCK_MATVEC    CitcomS gather(), citcoms/lib/Element_calculations.c:453-471 — element-by-element K·u with
//                           scatter-add into shared equations. Done here node-centrically: each node owns the (element, local
//                           node) terms that scatter into it (the transpose map the reference's stale CUDA built on the host,
//                           citcoms/lib/cgrad_kernel.cu:89-180), so there are no atomics and the sum order is fixed.
//                           HBM-bound (every 24×24 block is read exactly once): not an MFMA shape (one right-hand side).
//   DENSE_ROW_TIMES_MATRIX  DeePMD OptMatmul lambda, deepmd/source/op/opt_matmul.cc:52-58 — a dense fp64 GEMM
//                           result[M×K] = xx[M×N]·w[N×K]: the dense-tile case, on v_mfma_f64_16x16x4_f64.
//   SYM_QUADRATIC_FORM      Cantera gather1/apply1, gather2/apply2, cantera/src/thermo/RedlichKwongMFTP.cpp:927-970.
// Any OTHER callback pair gets the interface's own semantics — the reference's driver loop, on the host, in the reference's order
// (g4s_spmm_dense below; g4s_set_host_callback_policy chooses serial / threadNum threads for race-free gathers / refusal). That loop is what
// "gather degree times per vertex, then apply" means for host code; it is not a CPU twin of any kernel and never calls oracle/.
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (p && n <= bytes) return G4S_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = g4s::device_malloc(&p, n);
        if (e != hipSuccess) return g4s::set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        bytes = n;
        return G4S_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ------------------------------------------------------------------------------------------------ element-block mat-vec
// One wavefront per node. The node's terms (element e, local node a) — at most 8 on a hexahedral mesh — are taken 8 at a time,
// 8 lanes per term: lane q of a term handles columns q, q+8, q+16, … of the element's n = npe·dof unknowns for the `dof` rows
// (dof·a + i) of K_e. Partial sums are combined by shuffles inside the 8 lanes, then across terms in term order (fixed ⇒ the
// result is reproducible). Rows of K_e for one node are contiguous (dof·n doubles), and consecutive nodes are handled by
// consecutive waves.
constexpr int kMaxDof = 4;

__global__ __launch_bounds__(256) void elem_matvec_kernel(int nno, int npe, int dof, const int *__restrict__ node_ptr,
                                                           const int *__restrict__ node_terms, const int *__restrict__ elem_eq,
                                                           const int *__restrict__ node_eq, const double *__restrict__ elt_k,
                                                           const double *__restrict__ u, double *__restrict__ Au, double beta,
                                                           const int *__restrict__ skip)
{
    const int lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= nno || (skip && *skip)) return;                   // skip: the `done` flag of a solver that enqueues iterations ahead
    const int n = npe * dof;
    const int t0 = node_ptr[node], t1 = node_ptr[node + 1];
    const int g = lane >> 3, q = lane & 7;
    double tot[kMaxDof] = {0.0, 0.0, 0.0, 0.0};
    for (int tb = t0; tb < t1; tb += 8) {
        double acc[kMaxDof] = {0.0, 0.0, 0.0, 0.0};
        const int t = tb + g;
        if (t < t1) {
            const int term = node_terms[t];          // e·npe + a
            const int e = term / npe, a = term - e * npe;
            const double *K = elt_k + (size_t)e * n * n + (size_t)(dof * a) * n;
            const int *eq = elem_eq + (size_t)e * n;
            for (int c = q; c < n; c += 8) {
                const double uc = u[eq[c]];
                for (int i = 0; i < dof; ++i) acc[i] += K[i * n + c] * uc;
            }
        }
        for (int i = 0; i < dof; ++i) {
            double v = acc[i];
            v += __shfl_down(v, 4, 8);
            v += __shfl_down(v, 2, 8);
            v += __shfl_down(v, 1, 8);
            // term sums now sit in lanes 0, 8, …, 56: add them in term order
            for (int gg = 0; gg < 8; ++gg) tot[i] += __shfl(v, gg * 8, 64);
        }
    }
    if (lane < dof) {
        const int eqn = node_eq[node * dof + lane];
        double r = tot[0];
        if (lane == 1) r = tot[1];
        if (lane == 2) r = tot[2];
        if (lane == 3) r = tot[3];
        Au[eqn] = beta == 0.0 ? r : r + beta * Au[eqn];
    }
}

// Fixed-shape fast path (the CitcomS shape: 8-node hexahedra, 3 dof, and at most 8 elements around a node). The generic kernel
// above pays eight dependent memory latencies per wave (row pointer → terms → 3 × (equation id → u)); here the node's 8 term slots
// come from a padded table and every lane issues its 3 equation ids and its 9 matrix entries before anything is consumed, so a
// wave pays three: terms → (ids, K) → u.
template <int NPE, int DOF>
__global__ __launch_bounds__(256) void elem_matvec_fixed_kernel(int nno, const int *__restrict__ terms8 /* [nno][8], −1 = none */,
                                                                 const int *__restrict__ elem_eq, const int *__restrict__ node_eq,
                                                                 const double *__restrict__ elt_k, const double *__restrict__ u,
                                                                 double *__restrict__ Au, double beta, const int *__restrict__ skip)
{
    constexpr int N = NPE * DOF, CPL = N / 8;                      // columns per lane (3 for N = 24)
    static_assert(N % 8 == 0, "npe·dof must be a multiple of 8");
    const int lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= nno || (skip && *skip)) return;
    const int g = lane >> 3, q = lane & 7;
    const int term = terms8[node * 8 + g];
    double acc[DOF];
#pragma unroll
    for (int i = 0; i < DOF; ++i) acc[i] = 0.0;
    if (term >= 0) {
        const int e = term / NPE, a = term - e * NPE;
        const double *K = elt_k + (size_t)e * N * N + (size_t)(DOF * a) * N;
        const int *eq = elem_eq + (size_t)e * N;
        int ids[CPL];
        double k[DOF][CPL], uc[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) ids[j] = eq[q + 8 * j];
#pragma unroll
        for (int i = 0; i < DOF; ++i)
#pragma unroll
            for (int j = 0; j < CPL; ++j) k[i][j] = __builtin_nontemporal_load(K + i * N + q + 8 * j);
#pragma unroll
        for (int j = 0; j < CPL; ++j) uc[j] = u[ids[j]];
#pragma unroll
        for (int i = 0; i < DOF; ++i)
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[i] += k[i][j] * uc[j];
    }
    double tot[DOF];
#pragma unroll
    for (int i = 0; i < DOF; ++i) {
        double v = acc[i];
        v += __shfl_down(v, 4, 8);
        v += __shfl_down(v, 2, 8);
        v += __shfl_down(v, 1, 8);
        tot[i] = 0.0;
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) tot[i] += __shfl(v, gg * 8, 64);   // term sums in term order: reproducible
    }
    if (lane < DOF) {
        const int eqn = node_eq[node * DOF + lane];
        double r = tot[0];
#pragma unroll
        for (int i = 1; i < DOF; ++i) if (lane == i) r = tot[i];
        Au[eqn] = beta == 0.0 ? r : r + beta * Au[eqn];
    }
}

} // namespace

struct g4s_elem_op_s {
    int nel = 0, npe = 0, dof = 0, nno = 0, neq = 0;
    DevBuf node_ptr, node_terms, elem_eq, node_eq, terms8;
    bool fixed8 = false;            // every node has <= 8 terms and the shape is (8, 3): use elem_matvec_fixed_kernel
    const double *elt_k = nullptr;  // borrowed device pointer
};

G4S_API g4s_status g4s_elem_op_create(g4s_elem_op_t *out, int32_t numElems, int32_t npe, int32_t dof,
                                      const int32_t *ien, const int32_t *id, int32_t nno, int32_t neq, const double *elt_k_dev)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_REQUIRE(numElems >= 0 && npe > 0 && dof > 0 && dof <= kMaxDof && nno >= 0 && neq >= 0, "bad sizes (dof <= 4)");
    G4S_REQUIRE(ien && id, "ien/id is NULL");
    G4S_REQUIRE((int64_t)numElems * npe * dof * npe * dof < ((int64_t)1 << 40), "element matrix array too large");
    const int n = npe * dof;
    // validate the index maps on the host: an out-of-range id would be a GPU fault, a duplicated equation a race
    std::vector<char> seen((size_t)neq, 0);
    for (int64_t k = 0; k < (int64_t)nno * dof; ++k) {
        const int eqn = id[k];
        if (eqn < 0 || eqn >= neq) return g4s::set_error(G4S_ERR_INVALID, "g4s_elem_op_create: id[%lld] = %d outside [0,%d)", (long long)k, eqn, neq);
        if (seen[eqn]) return g4s::set_error(G4S_ERR_INVALID, "g4s_elem_op_create: equation %d is owned by two (node,dof) pairs", eqn);
        seen[eqn] = 1;
    }
    std::vector<int> cnt((size_t)nno + 1, 0);
    for (int64_t k = 0; k < (int64_t)numElems * npe; ++k) {
        if (ien[k] < 0 || ien[k] >= nno) return g4s::set_error(G4S_ERR_INVALID, "g4s_elem_op_create: ien[%lld] = %d outside [0,%d)", (long long)k, ien[k], nno);
        cnt[ien[k] + 1]++;
    }
    for (int i = 0; i < nno; ++i) cnt[i + 1] += cnt[i];
    std::vector<int> terms((size_t)numElems * npe), cur(cnt.begin(), cnt.end() - 1), eeq((size_t)numElems * n);
    for (int e = 0; e < numElems; ++e)
        for (int a = 0; a < npe; ++a) {
            const int node = ien[e * npe + a];
            terms[cur[node]++] = e * npe + a;           // ascending e within a node: the order the reference visits elements in
            for (int d = 0; d < dof; ++d) eeq[(size_t)e * n + a * dof + d] = id[node * dof + d];
        }
    auto op = std::make_unique<g4s_elem_op_s>();
    op->nel = numElems; op->npe = npe; op->dof = dof; op->nno = nno; op->neq = neq; op->elt_k = elt_k_dev;
    G4S_TRY(op->node_ptr.alloc(sizeof(int) * cnt.size()));
    G4S_TRY(op->node_terms.alloc(sizeof(int) * terms.size()));
    G4S_TRY(op->elem_eq.alloc(sizeof(int) * eeq.size()));
    G4S_TRY(op->node_eq.alloc(sizeof(int) * (size_t)nno * dof));
    G4S_HIP_TRY(hipMemcpy(op->node_ptr.p, cnt.data(), sizeof(int) * cnt.size(), hipMemcpyHostToDevice));
    if (!terms.empty()) G4S_HIP_TRY(hipMemcpy(op->node_terms.p, terms.data(), sizeof(int) * terms.size(), hipMemcpyHostToDevice));
    if (!eeq.empty()) G4S_HIP_TRY(hipMemcpy(op->elem_eq.p, eeq.data(), sizeof(int) * eeq.size(), hipMemcpyHostToDevice));
    if (nno) G4S_HIP_TRY(hipMemcpy(op->node_eq.p, id, sizeof(int) * (size_t)nno * dof, hipMemcpyHostToDevice));
    int max_terms = 0;
    for (int i = 0; i < nno; ++i) max_terms = std::max(max_terms, cnt[i + 1] - cnt[i]);
    if (npe == 8 && dof == 3 && max_terms <= 8 && nno > 0) {
        std::vector<int> t8((size_t)nno * 8, -1);
        for (int i = 0; i < nno; ++i)
            for (int t = cnt[i]; t < cnt[i + 1]; ++t) t8[(size_t)i * 8 + (t - cnt[i])] = terms[t];
        G4S_TRY(op->terms8.alloc(sizeof(int) * t8.size()));
        G4S_HIP_TRY(hipMemcpy(op->terms8.p, t8.data(), sizeof(int) * t8.size(), hipMemcpyHostToDevice));
        op->fixed8 = true;
    }
    *out = op.release();
    return G4S_OK;
}

namespace {
// diag[eq(node,i)] = Σ over the node's terms of K_e[p·n + p], p = dof·a + i  (build_diagonal_of_K, Element_calculations.c:580-611)
__global__ void elem_diagonal_kernel(int nno, int npe, int dof, const int *__restrict__ node_ptr, const int *__restrict__ node_terms,
                                     const int *__restrict__ node_eq, const double *__restrict__ elt_k, double *__restrict__ diag)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nno * dof) return;
    const int node = idx / dof, i = idx - node * dof, n = npe * dof;
    double s = 0.0;
    for (int t = node_ptr[node]; t < node_ptr[node + 1]; ++t) {
        const int term = node_terms[t], e = term / npe, a = term - e * npe, p = dof * a + i;
        s += elt_k[(size_t)e * n * n + (size_t)p * n + p];
    }
    diag[node_eq[idx]] = s;
}
} // namespace

extern "C" g4s_status g4s_elem_op_diagonal_sum(g4s_elem_op_t op, double *diag_dev, void *stream)
{
    G4S_REQUIRE(op && diag_dev && (op->elt_k || op->nel == 0), "NULL argument / no element matrices bound");
    hipStream_t s = g4s::as_stream(stream);
    if (op->neq) G4S_HIP_TRY(hipMemsetAsync(diag_dev, 0, sizeof(double) * (size_t)op->neq, s));
    const int total = op->nno * op->dof;
    if (total) hipLaunchKernelGGL(elem_diagonal_kernel, dim3((total + 255) / 256), dim3(256), 0, s, op->nno, op->npe, op->dof, op->node_ptr.as<int>(),
                                  op->node_terms.as<int>(), op->node_eq.as<int>(), op->elt_k, diag_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}
int g4s_elem_op_neq(g4s_elem_op_t op) { return op ? op->neq : 0; }
int g4s_elem_op_view(g4s_elem_op_t op, g4s::ElemOpView *out)
{
    G4S_REQUIRE(op && out, "NULL argument");
    *out = g4s::ElemOpView{op->nel, op->npe, op->dof, op->nno, op->neq, op->node_ptr.as<int>(), op->node_terms.as<int>(), op->elem_eq.as<int>(), op->node_eq.as<int>()};
    return G4S_OK;
}

G4S_API g4s_status g4s_elem_op_destroy(g4s_elem_op_t op)
{
    delete op;
    return G4S_OK;
}

static int elem_op_launch(g4s_elem_op_t op, const double *elt_k, const double *u, double *Au, double beta, hipStream_t s, const int *skip = nullptr)
{
    // equations no node owns receive nothing: with beta == 0 they must read 0 (Element_calculations.c:495-496 zeroes Au first).
    // When every equation has an owner (neq == nno·dof, the CitcomS numbering) the kernel writes all of Au and the memset is skipped.
    if (beta == 0.0 && op->neq && (int64_t)op->nno * op->dof != op->neq) G4S_HIP_TRY(hipMemsetAsync(Au, 0, sizeof(double) * (size_t)op->neq, s));
    if (op->nno) {
        if (op->fixed8)
            hipLaunchKernelGGL((elem_matvec_fixed_kernel<8, 3>), dim3((op->nno + 3) / 4), dim3(256), 0, s, op->nno, op->terms8.as<int>(),
                               op->elem_eq.as<int>(), op->node_eq.as<int>(), elt_k, u, Au, beta, skip);
        else
            hipLaunchKernelGGL(elem_matvec_kernel, dim3((op->nno + 3) / 4), dim3(256), 0, s, op->nno, op->npe, op->dof, op->node_ptr.as<int>(),
                               op->node_terms.as<int>(), op->elem_eq.as<int>(), op->node_eq.as<int>(), elt_k, u, Au, beta, skip);
        G4S_HIP_TRY(hipGetLastError());
    }
    return G4S_OK;
}

// g4s_elem_op_apply that returns at once when *skip_dev != 0 (cg.hip: iterations enqueued past the one that converged)
int g4s_elem_op_apply_unless(g4s_elem_op_t op, const double *u_dev, double *Au_dev, const int *skip_dev, void *stream)
{
    G4S_REQUIRE(op && u_dev && Au_dev, "NULL argument");
    G4S_REQUIRE(op->elt_k || op->nel == 0, "no element matrices bound");
    return elem_op_launch(op, op->elt_k, u_dev, Au_dev, 0.0, g4s::as_stream(stream), skip_dev);
}

G4S_API g4s_status g4s_elem_op_apply(g4s_elem_op_t op, const double *u_dev, double *Au_dev, void *stream)
{
    G4S_REQUIRE(op && u_dev && Au_dev, "NULL argument");
    G4S_REQUIRE(op->elt_k || op->nel == 0, "no element matrices bound");
    return elem_op_launch(op, op->elt_k, u_dev, Au_dev, 0.0, g4s::as_stream(stream));
}

// ------------------------------------------------------------------------------------------------ dense rows × matrix (fp64 MFMA)
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int kGemmKC = 64;    // rows of w staged per step
constexpr int kGemmNC = 128;   // columns of w / result per workgroup pass (8 MFMA column tiles)

// result[M×K] = xx[M×N]·w[N×K]. Workgroup = 4 waves = 64 rows of xx; wave = 16 rows × up to 128 columns (8 accumulator tiles
// of v_mfma_f64_16x16x4_f64: A lane map row = lane&15, k = lane>>4; B k = lane>>4, col = lane&15; D col = lane&15,
// row = (lane>>4) + 4·reg — cdna_hip_programming.md §3). w is staged through LDS in 64×128 panels (64 KiB); xx fragments are
// read straight from global (each 16×N strip is private to one wave and stays in L1 across the k loop).
// WT: w is stored transposed (K×N row-major, i.e. element (r, c) of the N×K operand is w[c·N + r]) — the dxx = grad·wᵀ product of
// the op's gradient reads the forward weights in place.
template <bool WT>
__global__ __launch_bounds__(256) void dense_rows_times_matrix_kernel(int M, int N, int K, const double *__restrict__ xx,
                                                                       const double *__restrict__ w, double *__restrict__ result)
{
    __shared__ double ws[kGemmKC * kGemmNC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 64 + wave * 16;
    const int arow = row0 + (lane & 15), kk = lane >> 4;
    const bool arow_ok = arow < M;
    for (int c0 = 0; c0 < K; c0 += kGemmNC) {
        const int ntiles = min(8, (K - c0 + 15) / 16);             // column tiles that hold real columns (uniform)
        double4_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < N; k0 += kGemmKC) {
            // this wave's A fragments of the whole 64-deep panel go out first (16 independent loads per lane): their latency
            // hides under the staging of the w panel instead of stalling every MFMA group
            double afrag[kGemmKC / 4];
#pragma unroll
            for (int sidx = 0; sidx < kGemmKC / 4; ++sidx) {
                const int k = k0 + 4 * sidx + kk;
                afrag[sidx] = (arow_ok && k < N) ? xx[(size_t)arow * N + k] : 0.0;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < kGemmKC * kGemmNC; idx += 256) {
                const int r = idx / kGemmNC, c = idx - r * kGemmNC;
                ws[idx] = (k0 + r < N && c0 + c < K) ? (WT ? w[(size_t)(c0 + c) * N + k0 + r] : w[(size_t)(k0 + r) * K + c0 + c]) : 0.0;
            }
            __syncthreads();
            const int ksteps = min(kGemmKC, N - k0 + 3) / 4;       // k-steps that hold real rows of w (uniform)
#pragma unroll
            for (int sidx = 0; sidx < kGemmKC / 4; ++sidx) {
                if (sidx < ksteps) {
                    const double a = afrag[sidx];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (t < ntiles) {
                            const double b = ws[(4 * sidx + kk) * kGemmNC + t * 16 + (lane & 15)];
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int col = c0 + t * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + (lane >> 4) + 4 * r;
                if (row < M && col < K) result[(size_t)row * K + col] = acc[t][r];
            }
        }
    }
}

// Persistent variant for the embedding-net sizes (N·roundup(K,16)·8 ≤ 96 KiB): w is staged in LDS ONCE per workgroup and stays
// there; the 8 waves of a workgroup walk 16-row strips of xx in a grid-stride loop, and a wave loads the A fragments of its NEXT
// strip while the MFMAs of the current one run. KT = column tiles of 16 (K ≤ 16·KT ≤ 128); NS = k-steps of 4 (N ≤ 4·NS).
template <int KT, bool WT>
__global__ __launch_bounds__(512) void dense_rows_times_matrix_resident_kernel(int M, int N, int K, int NS, const double *__restrict__ xx,
                                                                                const double *__restrict__ w, double *__restrict__ result)
{
    extern __shared__ double wres[];                               // [4·NS][16·KT], zero padded
    constexpr int KP = 16 * KT;
    constexpr int MAXS = 32;                                       // N ≤ 128
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cl = lane & 15;
    for (int idx = threadIdx.x; idx < 4 * NS * KP; idx += 512) {
        const int r = idx / KP, c = idx - r * KP;
        wres[idx] = (r < N && c < K) ? (WT ? w[(size_t)c * N + r] : w[(size_t)r * K + c]) : 0.0;
    }
    __syncthreads();
    const int nstrips = (M + 15) / 16, stride = gridDim.x * 8;
    int strip = blockIdx.x * 8 + wave;
    double a_cur[MAXS], a_nxt[MAXS];
    auto load_a = [&](int sp, double (&a)[MAXS]) {
        const int row = sp * 16 + cl;
        const bool ok = sp < nstrips && row < M;
#pragma unroll
        for (int sidx = 0; sidx < MAXS; ++sidx) {
            const int k = 4 * sidx + kk;
            a[sidx] = (sidx < NS && ok && k < N) ? xx[(size_t)row * N + k] : 0.0;
        }
    };
    load_a(strip, a_cur);
    for (; strip < nstrips; strip += stride) {
        load_a(strip + stride, a_nxt);
        double4_t acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sidx = 0; sidx < MAXS; ++sidx) {
            if (sidx < NS) {
#pragma unroll
                for (int t = 0; t < KT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[sidx], wres[(4 * sidx + kk) * KP + t * 16 + cl], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int col = t * 16 + cl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = strip * 16 + kk + 4 * r;
                if (row < M && col < K) result[(size_t)row * K + col] = acc[t][r];
            }
        }
#pragma unroll
        for (int sidx = 0; sidx < MAXS; ++sidx) a_cur[sidx] = a_nxt[sidx];
    }
}

// The resident kernel rewritten for the memory pipeline (round 3). The ISA of the version above showed: every A-fragment load under its own
// per-lane branch (32 of them per strip), an s_waitcnt vmcnt(0) in front of the first MFMA — i.e. the "prefetch" of the next strip was waited for on the
// spot, behind the result stores just issued — and an lgkmcnt(0) between each pair of MFMAs and the LDS read of their B fragments. Here:
//   * a lane loads PAIRS of consecutive k (16-byte loads: 16 rows × 64 B per instruction instead of 16 × 32 B); pair g of lane (row i, kk) holds
//     k = 8g + 2kk + j, j = 0, 1, and MFMA step (g, j) takes element j — so the rows of w sit in LDS in the order 8g + 4j + kk (a lane's four kk are
//     consecutive LDS rows, as before: no bank conflict). N must be even; GT = ceil(N/8) pairs, a template parameter (4, 7, 13, 16 ↔ N ≤ 32, 50–56, 98–104, 122–128);
//   * all loads are unconditional (row clamped to M − 1, a pair past N clamped to the row's first pair and replaced by zeros when it is used);
//   * EVERY lane issues all 4·KT stores of a strip (a lane outside M × K writes to its wave's dump slot): no per-lane branches around the stores;
//   * the B fragments of step s + 1 are read from LDS before the MFMAs of step s are issued.
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int KT, int GT, bool WT>
__global__ __launch_bounds__(512) void dense_rows_times_matrix_resident2_kernel(int M, int N, int K, const double *__restrict__ xx, const double *__restrict__ w,
                                                                                 double *__restrict__ result, double *__restrict__ dump)
{
    extern __shared__ double wres[];                               // [8·G][16·KT] in the permuted row order, zero padded
    constexpr int KP = 16 * KT;
    constexpr int G = GT;                                          // = ceil(N/8): the launcher picks the instantiation (a runtime bound would put every load under a branch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = lane >> 4, cl = lane & 15;
    for (int idx = threadIdx.x; idx < 8 * G * KP; idx += 512) {
        const int rp = idx / KP, c = idx - rp * KP;                // LDS row rp = 8g + 4j + kk' holds k = 8g + 2kk' + j
        const int g = rp >> 3, j = (rp >> 2) & 1, k2 = rp & 3, r = 8 * g + 2 * k2 + j;
        wres[idx] = (r < N && c < K) ? (WT ? w[(size_t)c * N + r] : w[(size_t)r * K + c]) : 0.0;
    }
    __syncthreads();
    const int nstrips = (M + 15) / 16, stride = gridDim.x * 8;
    double *const my_dump = dump + (blockIdx.x * 8 + wave);
    struct Tile { double2_t a[GT]; };
    auto request = [&](Tile &T, int sp) {
        const int row = min(sp * 16 + cl, M - 1);                  // (a strip past the end repeats the last row: loaded, never stored)
        const double *base = xx + (size_t)row * N;
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const int k0 = 8 * g + 2 * kk;
            T.a[g] = *reinterpret_cast<const double2_t *>(base + (k0 < N ? k0 : 0));
        }
    };
    auto work = [&](const Tile &T, int sp) {
        double4_t acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        double bf[2][KT];
        auto read_b = [&](double (&b)[KT], int step) {             // step = 2g + j → LDS rows 8g + 4j + kk
            const double *src = wres + (size_t)(4 * step + kk) * KP + cl;
#pragma unroll
            for (int t = 0; t < KT; ++t) b[t] = src[t * 16];
        };
        read_b(bf[0], 0);
#pragma unroll
        for (int g = 0; g < GT; ++g) {
            const bool in = 8 * g + 2 * kk < N;                    // (only the last group can hold pairs past N)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int step = 2 * g + j;
                if (step + 1 < 2 * G) read_b(bf[(step + 1) & 1], step + 1);   // compile time; issued before this step's MFMAs
                __builtin_amdgcn_sched_barrier(0);                 // (… and kept there: left alone the scheduler sinks the reads behind the MFMAs)
                const double a = (g + 1 < GT || in) ? T.a[g][j] : 0.0;
#pragma unroll
                for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[step & 1][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                 // without it the scheduler hoists all 2·GT·KT LDS reads to the top and spills
            }
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int col = t * 16 + cl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = sp * 16 + kk + 4 * r;
                double *dst = (row < M && col < K) ? result + (size_t)row * K + col : my_dump;
                *dst = acc[t][r];
            }
        }
    };
    // two A tiles take turns (no register copies): the next strip is requested before the current one's MFMAs; with every store issued by every lane the number
    // of memory operations between a tile's request and its first use is static, and hipcc's waits leave exactly those in flight
    int strip = blockIdx.x * 8 + wave;
    if (strip >= nstrips) return;                                  // (after the only barrier)
    Tile A, B;
    request(A, strip);
    for (;;) {
        request(B, strip + stride);
        work(A, strip);
        strip += stride;
        if (strip >= nstrips) break;
        request(A, strip + stride);
        work(B, strip);
        strip += stride;
        if (strip >= nstrips) break;
    }
}

// dw[N×K] = xxᵀ·grad: a reduction over the M rows (1e5–1e6) into a small matrix. One workgroup per (row range, 128×128 block
// of dw). The rows go through LDS in slabs of 32 — [32][128 xx columns | 128 grad columns | 8 pad] doubles — double buffered:
// thread t owns LDS column t and fetches its 32 rows of slab s+1 into registers (one base pointer + row stride per thread, all 32
// loads in flight, no dependent instruction until they are parked) before the MFMAs of slab s, and parks them in the other buffer
// afterwards; one barrier per slab. Columns past N / K and rows past the range are parked as zeros, so the MFMA loop has no
// conditions: wave w owns the 16-row tiles w and w+4 of the block × all eight 16-column tiles (16 accumulators of
// v_mfma_f64_16x16x4_f64: A[i][kk] = xx[m+kk][n0+i], B[kk][j] = grad[m+kk][k0+j]). The pad makes the four kk rows of an operand
// read fall on different banks. Two earlier versions: operands straight from HBM one 4-row step ahead — latency-bound, 11 TFLOP/s;
// LDS slabs with the out-of-range mask applied at the load — the compiler put an s_waitcnt after every load, 9 TFLOP/s.
// Range results go to partials[range][N][K]; dense_reduce_slabs_kernel adds them in order (no atomics).
#ifndef G4S_DW_SLAB
#define G4S_DW_SLAB 16
#endif
#ifndef G4S_DW_WGS
#define G4S_DW_WGS 512
#endif
constexpr int kDwSlab = G4S_DW_SLAB, kDwLd = 264;
template <int KT>                                                  // 16-column tiles of the block that hold real columns of dw
__global__ __launch_bounds__(256) void dense_rows_transposed_times_rows_kernel(int M, int N, int K, int rows_per_wg,
                                                                                const double *__restrict__ xx, const double *__restrict__ g,
                                                                                double *__restrict__ partials)
{
    extern __shared__ double slab[];                               // 2 × [kDwSlab][kDwLd]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int il = lane & 15, kk = lane >> 4;
    const int nb = blockIdx.y * 128, kb = blockIdx.z * 128;
    const int m0 = blockIdx.x * rows_per_wg, m1 = min(M, m0 + rows_per_wg);
    // this thread's column of the slab: an xx column (t < 128) or a grad column
    const bool in_x = t < 128;
    const int col = in_x ? nb + t : kb + t - 128, width = in_x ? N : K;
    const bool col_ok = col < width;
    const double *src = (in_x ? xx : g) + min(col, width - 1);
    const int ntl = min(2, max(0, (min(N - nb, 128) - 16 * wave + 63) / 64));   // row tiles w, w+4 of this wave that hold real rows of dw
    for (int i = t; i < 2 * kDwSlab * 8; i += 256) {               // the pad columns are read by no one but keep them defined
        const int b = i / (kDwSlab * 8), r = (i / 8) % kDwSlab;
        slab[b * kDwSlab * kDwLd + r * kDwLd + 256 + (i & 7)] = 0.0;
    }
    double4_t acc[2][KT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
    double stage[kDwSlab];
    auto fetch = [&](int m) {                                      // clamped rows: always in bounds, masked when parked
#pragma unroll
        for (int u = 0; u < kDwSlab; ++u) stage[u] = src[(size_t)min(m + u, m1 - 1) * width];
    };
    auto park = [&](double *buf, int m) {
#pragma unroll
        for (int u = 0; u < kDwSlab; ++u) buf[u * kDwLd + t] = (col_ok && m + u < m1) ? stage[u] : 0.0;
    };
    if (m0 < m1) { fetch(m0); park(slab, m0); }
    __syncthreads();
    int cur = 0;
    for (int m = m0; m < m1; m += kDwSlab) {
        const bool more = m + kDwSlab < m1;
        if (more) fetch(m + kDwSlab);
        const double *buf = slab + cur * (kDwSlab * kDwLd);
#pragma unroll
        for (int st = 0; st < kDwSlab / 4; ++st) {
            const double *rowp = buf + (4 * st + kk) * kDwLd + il;
            double b[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) b[j] = rowp[128 + 16 * j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i < ntl) {                                     // wave-uniform: one scalar branch per row tile, none per MFMA
                    const double a = rowp[16 * (wave + 4 * i)];
#pragma unroll
                    for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) park(slab + (cur ^ 1) * (kDwSlab * kDwLd), m + kDwSlab);
        cur ^= 1;
        __syncthreads();
    }
    double *out = partials + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int k = kb + 16 * j + il;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 16 * (wave + 4 * i) + kk + 4 * r;
                if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][r];
            }
        }
}

// out[i] = Σ_b partials[b][i], b ascending within each of 16 interleaved groups, groups combined in a fixed tree: the same bits
// every run. 64 elements × 16 groups per workgroup (one thread summing all ranges alone is a chain of dependent loads).
__global__ __launch_bounds__(1024) void dense_reduce_slabs_kernel(int slabs, size_t elems, const double *__restrict__ partials, double *__restrict__ out)
{
    __shared__ double part[16][64];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + e;
    double s = 0.0;
    if (i < elems)
        for (int b = grp; b < slabs; b += 16) s += partials[(size_t)b * elems + i];
    part[grp][e] = s;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (grp < half) part[grp][e] += part[grp + half][e];
        __syncthreads();
    }
    if (grp == 0 && i < elems) out[i] = part[0][e];
}

// ------------------------------------------------------------------------------------------------ symmetric quadratic form
// result[0] += Σ_i Σ_{j<i} x_i x_j (a[num·(i+m·j)] + a[num·(j+m·i)]) + Σ_i x_i² a[num·(i+m·i)];  result[1]: see g4s.h.
// One workgroup; thread t owns rows t, t+256, …; workgroup tree reduction in a fixed shape.
__global__ __launch_bounds__(256) void sym_quadratic_form_kernel(int m, int numbers, const double *__restrict__ a, const double *__restrict__ x,
                                                                  const double *__restrict__ b, double *__restrict__ out2)
{
    __shared__ double s0[256], s1[256];
    double r0 = 0.0, r1 = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) {
        for (int j = 0; j < i; ++j) {
            const size_t c1 = (size_t)i + (size_t)m * j, c2 = (size_t)j + (size_t)m * i;
            const double tmp = x[i] * x[j];
            r0 += tmp * (a[numbers * c1] + a[numbers * c2]);
            if (numbers > 1) r1 += tmp * (a[numbers * c1 + 1] + a[numbers * c2 + 1]);
        }
        const size_t c = (size_t)i + (size_t)m * i;
        const double tmp = x[i] * x[i];
        r0 += tmp * a[numbers * c];
        if (numbers > 1) r1 += tmp * a[numbers * c + 1];
        else if (b) r1 += x[i] * b[i];
    }
    s0[threadIdx.x] = r0; s1[threadIdx.x] = r1;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { s0[threadIdx.x] += s0[threadIdx.x + off]; s1[threadIdx.x] += s1[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = s0[0]; out2[1] = s1[0]; }
}

} // namespace

namespace {
template <bool WT>
int dense_rows_times_matrix_launch(int32_t M, int32_t N, int32_t K, const double *xx_dev, const double *w_dev, double *result_dev, void *stream)
{
    const int KT = (K + 15) / 16, NS = (N + 3) / 4;
    const size_t lds = sizeof(double) * 4 * (size_t)NS * 16 * KT;
    // the rewritten resident kernel: even N (16-byte pairs of xx), the column-tile counts of the embedding-net sizes (K = 25/50/100/128 → 2/4/7/8 tiles)
    const int G = (N + 7) / 8, GT = G;
    const size_t lds2 = sizeof(double) * 8 * (size_t)G * 16 * KT;
    if (N >= 2 && N % 2 == 0 && N <= 128 && K <= 128 && M >= 4096 && (KT == 2 || KT == 4 || KT == 7 || KT == 8) && (G == 4 || G == 7 || G == 13 || (G == 16 && KT <= 4)) /* (7–8 tiles × 16 groups would spill) */ && lds2 <= 140 * 1024 &&
        (reinterpret_cast<uintptr_t>(xx_dev) & 15u) == 0) {
        hipStream_t s = g4s::as_stream(stream);
        const int strips = (M + 15) / 16, grid = std::min(256, (strips + 7) / 8);
        double *dump = nullptr;
        G4S_TRY(g4s::scratch_alloc(reinterpret_cast<void **>(&dump), sizeof(double) * 8 * (size_t)grid, s));
        auto launch = [&](auto kern) -> int {
            G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds2, s, M, N, K, xx_dev, w_dev, result_dev, dump);
            return G4S_OK;
        };
        int st = G4S_OK;
#define G4S_DENSE2_GT(KT_)                                                                                           \
        switch (GT) {                                                                                                \
        case 4: st = launch(dense_rows_times_matrix_resident2_kernel<KT_, 4, WT>); break;                            \
        case 7: st = launch(dense_rows_times_matrix_resident2_kernel<KT_, 7, WT>); break;                            \
        case 13: st = launch(dense_rows_times_matrix_resident2_kernel<KT_, 13, WT>); break;                          \
        default: st = launch(dense_rows_times_matrix_resident2_kernel<KT_, 16, WT>); break;                          \
        }
        switch (KT) {
        case 2: G4S_DENSE2_GT(2) break;
        case 4: G4S_DENSE2_GT(4) break;
        case 7: G4S_DENSE2_GT(7) break;
        default: G4S_DENSE2_GT(8) break;
        }
#undef G4S_DENSE2_GT
        const hipError_t le = hipGetLastError();
        g4s::scratch_free(dump, s);
        G4S_TRY(st);
        G4S_HIP_TRY(le);
        return G4S_OK;
    }
    if (N >= 1 && N <= 128 && K <= 128 && lds <= 96 * 1024 && M >= 4096) {
        // embedding-net shapes: w resident in LDS, persistent strips
        const int strips = (M + 15) / 16, grid = std::min(256 * 1, (strips + 7) / 8);
        auto launch = [&](auto kern) -> int {
            G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, g4s::as_stream(stream), M, N, K, NS, xx_dev, w_dev, result_dev);
            return G4S_OK;
        };
        int st = G4S_OK;
        switch (KT) {
        case 1: st = launch(dense_rows_times_matrix_resident_kernel<1, WT>); break;
        case 2: st = launch(dense_rows_times_matrix_resident_kernel<2, WT>); break;
        case 3: st = launch(dense_rows_times_matrix_resident_kernel<3, WT>); break;
        case 4: st = launch(dense_rows_times_matrix_resident_kernel<4, WT>); break;
        case 5: st = launch(dense_rows_times_matrix_resident_kernel<5, WT>); break;
        case 6: st = launch(dense_rows_times_matrix_resident_kernel<6, WT>); break;
        case 7: st = launch(dense_rows_times_matrix_resident_kernel<7, WT>); break;
        default: st = launch(dense_rows_times_matrix_resident_kernel<8, WT>); break;
        }
        G4S_TRY(st);
    } else {
        hipLaunchKernelGGL(dense_rows_times_matrix_kernel<WT>, dim3((M + 63) / 64), dim3(256), 0, g4s::as_stream(stream), M, N, K, xx_dev, w_dev, result_dev);
    }
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}
} // namespace

G4S_API g4s_status g4s_dense_rows_times_matrix(int32_t M, int32_t N, int32_t K, const double *xx_dev, const double *w_dev,
                                               double *result_dev, void *stream)
{
    G4S_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    if (M == 0 || K == 0) return G4S_OK;
    G4S_REQUIRE(result_dev && (N == 0 || (xx_dev && w_dev)), "NULL argument");
    return dense_rows_times_matrix_launch<false>(M, N, K, xx_dev, w_dev, result_dev, stream);
}

G4S_API g4s_status g4s_dense_rows_times_matrix_grad(int32_t M, int32_t N, int32_t K, const double *xx_dev, const double *w_dev,
                                                    const double *grad_dev, double *dxx_dev, double *dw_dev, void *stream)
{
    G4S_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative dimension");
    hipStream_t s = g4s::as_stream(stream);
    // dxx[M×N] = grad[M×K]·wᵀ: the forward kernels with the roles of N and K swapped and w read transposed in place
    if (dxx_dev && M > 0 && N > 0) {
        G4S_REQUIRE(K == 0 || (grad_dev && w_dev), "NULL argument");
        G4S_TRY(dense_rows_times_matrix_launch<true>(M, K, N, grad_dev, w_dev, dxx_dev, stream));
    }
    // dw[N×K] = xxᵀ·grad
    if (dw_dev && N > 0 && K > 0) {
        if (M == 0) { G4S_HIP_TRY(hipMemsetAsync(dw_dev, 0, sizeof(double) * (size_t)N * K, s)); return G4S_OK; }
        G4S_REQUIRE(xx_dev && grad_dev, "NULL argument");
        const int wgs = std::min(G4S_DW_WGS, (M + kDwSlab - 1) / kDwSlab);
        const int rows_per_wg = ((M + wgs - 1) / wgs + kDwSlab - 1) / kDwSlab * kDwSlab;
        const int used = (M + rows_per_wg - 1) / rows_per_wg;
        const size_t elems = (size_t)N * K, lds = sizeof(double) * 2 * kDwSlab * kDwLd;
        double *partials = nullptr;                                // stream-ordered scratch: no host sync, the pool keeps the pages
        G4S_TRY(g4s::scratch_alloc(reinterpret_cast<void **>(&partials), sizeof(double) * elems * used, s));
        auto launch = [&](auto kern) -> int {
            G4S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(used, (N + 127) / 128, (K + 127) / 128), dim3(256), lds, s, M, N, K, rows_per_wg, xx_dev, grad_dev, partials);
            return G4S_OK;
        };
        int st = G4S_OK;
        switch (K > 128 ? 8 : (K + 15) / 16) {                     // blocks of a wider dw all run the 8-tile kernel (the last one masks its tail)
        case 1: st = launch(dense_rows_transposed_times_rows_kernel<1>); break;
        case 2: st = launch(dense_rows_transposed_times_rows_kernel<2>); break;
        case 3: st = launch(dense_rows_transposed_times_rows_kernel<3>); break;
        case 4: st = launch(dense_rows_transposed_times_rows_kernel<4>); break;
        case 5: st = launch(dense_rows_transposed_times_rows_kernel<5>); break;
        case 6: st = launch(dense_rows_transposed_times_rows_kernel<6>); break;
        case 7: st = launch(dense_rows_transposed_times_rows_kernel<7>); break;
        default: st = launch(dense_rows_transposed_times_rows_kernel<8>); break;
        }
        if (st != G4S_OK) { g4s::scratch_free(partials, s); return st; }
        hipLaunchKernelGGL(dense_reduce_slabs_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(1024), 0, s, used, elems, partials, dw_dev);
        const hipError_t launch_err = hipGetLastError();
        g4s::scratch_free(partials, s);
        G4S_HIP_TRY(launch_err);
    }
    return G4S_OK;
}

G4S_API g4s_status g4s_sym_quadratic_form(int32_t m, int32_t numbers, const double *a, const double *x, const double *b, double *result)
{
    G4S_REQUIRE(m >= 0 && numbers >= 1 && result, "bad argument");
    if (m == 0) return G4S_OK;
    G4S_REQUIRE(a && x, "NULL argument");
    DevBuf da, dx, db, dout;
    const size_t na = (size_t)m * m * numbers;
    G4S_TRY(da.alloc(sizeof(double) * na));
    G4S_TRY(dx.alloc(sizeof(double) * (size_t)m));
    G4S_TRY(dout.alloc(sizeof(double) * 2));
    G4S_HIP_TRY(hipMemcpy(da.p, a, sizeof(double) * na, hipMemcpyHostToDevice));
    G4S_HIP_TRY(hipMemcpy(dx.p, x, sizeof(double) * (size_t)m, hipMemcpyHostToDevice));
    if (b && numbers == 1) {
        G4S_TRY(db.alloc(sizeof(double) * (size_t)m));
        G4S_HIP_TRY(hipMemcpy(db.p, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(sym_quadratic_form_kernel, dim3(1), dim3(256), 0, nullptr, m, numbers, da.as<double>(), dx.as<double>(),
                       (b && numbers == 1) ? db.as<double>() : nullptr, dout.as<double>());
    G4S_HIP_TRY(hipGetLastError());
    double h[2] = {0.0, 0.0};
    G4S_HIP_TRY(hipMemcpy(h, dout.p, sizeof(h), hipMemcpyDeviceToHost));
    result[0] += h[0];      // the callbacks accumulate into result (RedlichKwongMFTP.cpp:930-931, 962-963)
    result[1] += h[1];
    return G4S_OK;
}

// ------------------------------------------------------------------------------------------------ pattern registry + spmm_dense
namespace {

struct Pattern {
    g4s_pattern_desc desc{};
    std::vector<int32_t> ien, id;
    // device-side state reused across calls
    g4s_elem_op_t op = nullptr;
    DevBuf elt_k, u, Au, xx, w, res;
    double *stage = nullptr;                   // pinned host staging buffer for the row-pointer → contiguous pack
    size_t stage_bytes = 0;
    const double **cached_weights = nullptr;   // edgeWeight pointer whose contents elt_k currently holds (static_weights)
    ~Pattern() { delete op; if (stage) (void)hipHostFree(stage); }
    int stage_alloc(size_t bytes)
    {
        if (stage && bytes <= stage_bytes) return G4S_OK;
        if (stage) { (void)hipHostFree(stage); stage = nullptr; stage_bytes = 0; }
        if (hipHostMalloc((void **)&stage, bytes ? bytes : 1) != hipSuccess) return g4s::set_error(G4S_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        stage_bytes = bytes;
        return G4S_OK;
    }
};

std::mutex g_mu;
std::atomic<int> g_host_policy{G4S_HOST_CALLBACKS_SERIAL};
thread_local int t_host_policy = -1;                                // >= 0: this thread's calls use it instead of the process-wide policy (g4s::ScopedRaceFree, graph.hpp)
std::map<std::pair<void *, void *>, std::unique_ptr<Pattern>> g_patterns;

std::pair<void *, void *> key_of(fun_gather g, fun_apply a) { return {reinterpret_cast<void *>(g), reinterpret_cast<void *>(a)}; }

} // namespace

G4S_API g4s_status g4s_register_pattern(fun_gather gather, fun_apply apply, const g4s_pattern_desc *desc)
{
    G4S_REQUIRE(gather && desc, "NULL argument");
    auto p = std::make_unique<Pattern>();
    p->desc = *desc;
    switch (desc->kind) {
    case G4S_PATTERN_ELEMENT_BLOCK_MATVEC:
        G4S_REQUIRE(desc->ien && desc->id && desc->num_elems >= 0 && desc->nodes_per_elem > 0 && desc->dof > 0 && desc->nno >= 0 && desc->neq >= 0,
                    "incomplete ELEMENT_BLOCK_MATVEC descriptor");
        G4S_REQUIRE(desc->edge_weight_base == 0 || desc->edge_weight_base == 1, "edge_weight_base must be 0 or 1");
        p->id.assign(desc->id, desc->id + (size_t)desc->nno * desc->dof);
        p->ien.assign(desc->ien, desc->ien + (size_t)desc->num_elems * desc->nodes_per_elem);
        p->desc.id = nullptr;    // the copies above are what later calls use
        p->desc.ien = nullptr;
        break;
    case G4S_PATTERN_DENSE_ROW_TIMES_MATRIX:
        G4S_REQUIRE(desc->inner >= 0, "inner (N) must be >= 0");
        break;
    case G4S_PATTERN_SYM_QUADRATIC_FORM:
        G4S_REQUIRE(desc->numbers >= 1, "numbers must be >= 1");
        break;
    default:
        return g4s::set_error(G4S_ERR_INVALID, "g4s_register_pattern: unknown pattern kind %d", desc->kind);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_patterns[key_of(gather, apply)] = std::move(p);
    return G4S_OK;
}

G4S_API g4s_status g4s_set_host_callback_policy(int32_t policy)
{
    G4S_REQUIRE(policy == G4S_HOST_CALLBACKS_SERIAL || policy == G4S_HOST_CALLBACKS_PARALLEL || policy == G4S_HOST_CALLBACKS_REFUSE, "unknown policy");
    g_host_policy.store(policy);
    return G4S_OK;
}

G4S_API g4s_status g4s_set_host_callback_policy_thread(int32_t policy, int32_t *previous)
{
    G4S_REQUIRE(policy == -1 || policy == G4S_HOST_CALLBACKS_SERIAL || policy == G4S_HOST_CALLBACKS_PARALLEL || policy == G4S_HOST_CALLBACKS_REFUSE, "unknown policy");
    if (previous) *previous = t_host_policy;
    t_host_policy = policy;
    return G4S_OK;
}

G4S_API g4s_status g4s_unregister_pattern(fun_gather gather, fun_apply apply)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_patterns.erase(key_of(gather, apply));
    return G4S_OK;
}

G4S_API g4s_status g4s_spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight, const double *vertexStates,
                                  double *temp, double *result, fun_gather gather, fun_apply apply, double *time, int threadNum)
{
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_patterns.find(key_of(gather, apply));
    if (it == g_patterns.end()) {
        // An arbitrary callback pair is host code the device cannot run. It is not an error of the caller either: the interface promises
        // "gather degree times per vertex, then apply" for ANY pair (deepmd/source/op/graph.h:21-32; citcoms/lib/global_defs.h:48-49,854-857),
        // so the general case runs the reference's driver loop on the host, in the reference's order. This is the interface's semantics for
        // callbacks, not a CPU version of a device kernel: the three patterns with kernels never come here.
        const int policy = t_host_policy >= 0 ? t_host_policy : g_host_policy.load();
        lk.unlock();                                               // callbacks may call back into the library
        if (policy == G4S_HOST_CALLBACKS_REFUSE)
            return g4s::set_error(G4S_ERR_UNSUPPORTED, "spmm_dense: this (gather, apply) pair is not registered with g4s_register_pattern and "
                                                       "host callbacks are refused (g4s_set_host_callback_policy(G4S_HOST_CALLBACKS_REFUSE))");
        if (!gather) return g4s::set_error(G4S_ERR_INVALID, "spmm_dense: gather is NULL");
        const auto t0 = std::chrono::steady_clock::now();
        // the reference runs 8 OpenMP threads whatever the gather does (graph.h:23); here more than one thread only when the caller has
        // declared its gathers race-free, and then threadNum of them, vertices handed out one at a time (schedule(dynamic, 1), graph.h:24)
        const int threads = policy == G4S_HOST_CALLBACKS_PARALLEL ? std::max(1, std::min(threadNum, (int)std::min<uint32_t>(numNodes, 256u))) : 1;
        std::atomic<uint32_t> next{0};
        auto worker = [&]() {
            for (uint32_t vi = next.fetch_add(1); vi < numNodes; vi = next.fetch_add(1)) {
                for (uint32_t nb = 0; nb < degree; ++nb) gather((int)vi, (int)nb, edgeWeight, vertexStates, result);
                if (apply) apply((int)vi, edgeWeight, vertexStates, result);
            }
        };
        if (threads == 1) worker();
        else {
            std::vector<std::thread> pool;
            for (int i = 1; i < threads; ++i) pool.emplace_back(worker);
            worker();
            for (auto &th : pool) th.join();
        }
        (void)temp;                                                // the reference's callbacks receive result only (global_defs.h:48-49)
        if (time) *time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return G4S_OK;
    }
    Pattern &P = *it->second;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    G4S_HIP_TRY(hipEventCreate(&e0));
    G4S_HIP_TRY(hipEventCreate(&e1));
    int st = G4S_OK;
    auto finish = [&](int code) {
        if (code == G4S_OK && time) {
            float ms = 0.f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *time = ms * 1e-3;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return code;
    };
    switch (P.desc.kind) {
    case G4S_PATTERN_ELEMENT_BLOCK_MATVEC: {
        const int npe = P.desc.nodes_per_elem, dof = P.desc.dof, n = npe * dof, base = P.desc.edge_weight_base;
        if (degree != (uint32_t)npe) return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: degree %u != nodes_per_elem %d", degree, npe));
        if (numNodes != (uint32_t)P.desc.num_elems) return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: numNodes %u != registered num_elems %d", numNodes, P.desc.num_elems));
        if (!edgeWeight || !vertexStates || !result) return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: NULL argument"));
        const size_t kbytes = sizeof(double) * (size_t)numNodes * n * n;
        if (!P.op || P.op->nel != (int)numNodes) {
            delete P.op; P.op = nullptr; P.cached_weights = nullptr;
            st = g4s_elem_op_create(&P.op, (int)numNodes, npe, dof, P.ien.data(), P.id.data(), P.desc.nno, P.desc.neq, nullptr);
            if (st != G4S_OK) return finish(st);
        }
        if ((st = P.elt_k.alloc(kbytes)) != G4S_OK || (st = P.u.alloc(sizeof(double) * (size_t)P.desc.neq)) != G4S_OK ||
            (st = P.Au.alloc(sizeof(double) * (size_t)P.desc.neq)) != G4S_OK)
            return finish(st);
        // element matrices arrive as an array of row pointers (Drive_solvers.c:52-55): pack them into pinned memory, one H2D copy
        if (!(P.desc.static_weights && P.cached_weights == edgeWeight)) {
            if ((st = P.stage_alloc(kbytes)) != G4S_OK) return finish(st);
            for (uint32_t e = 0; e < numNodes; ++e) memcpy(P.stage + (size_t)e * n * n, edgeWeight[e + base], sizeof(double) * n * n);
            if (kbytes && hipMemcpyAsync(P.elt_k.p, P.stage, kbytes, hipMemcpyHostToDevice, nullptr) != hipSuccess)
                return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: H2D copy of the element matrices failed"));
            P.cached_weights = edgeWeight;
        }
        if (hipMemcpyAsync(P.u.p, vertexStates, sizeof(double) * (size_t)P.desc.neq, hipMemcpyHostToDevice, nullptr) != hipSuccess ||
            hipMemcpyAsync(P.Au.p, result, sizeof(double) * (size_t)P.desc.neq, hipMemcpyHostToDevice, nullptr) != hipSuccess)
            return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: H2D copy of u/Au failed"));
        (void)hipEventRecord(e0, nullptr);
        st = elem_op_launch(P.op, P.elt_k.as<double>(), P.u.as<double>(), P.Au.as<double>(), 1.0, nullptr); // gather does Au[aa] += …
        (void)hipEventRecord(e1, nullptr);
        if (st != G4S_OK) return finish(st);
        if (hipMemcpy(result, P.Au.p, sizeof(double) * (size_t)P.desc.neq, hipMemcpyDeviceToHost) != hipSuccess)
            return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: D2H copy of Au failed"));
        (void)temp;
        return finish(G4S_OK);
    }
    case G4S_PATTERN_DENSE_ROW_TIMES_MATRIX: {
        const int M = (int)numNodes, K = (int)degree, N = P.desc.inner;
        if (!edgeWeight || !vertexStates || !result) return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: NULL argument"));
        if ((st = P.xx.alloc(sizeof(double) * (size_t)M * N)) != G4S_OK || (st = P.w.alloc(sizeof(double) * (size_t)N * K)) != G4S_OK ||
            (st = P.res.alloc(sizeof(double) * (size_t)M * K)) != G4S_OK)
            return finish(st);
        // rows of xx arrive as row pointers (opt_matmul.cc:46-50); contiguous runs are merged into one copy
        for (int e = 0; e < M;) {
            int run = 1;
            while (e + run < M && edgeWeight[e + run] == edgeWeight[e] + (size_t)run * N) ++run;
            if (N && hipMemcpyAsync(P.xx.as<double>() + (size_t)e * N, edgeWeight[e], sizeof(double) * (size_t)run * N, hipMemcpyHostToDevice, nullptr) != hipSuccess)
                return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: H2D copy of xx failed"));
            e += run;
        }
        if (N && K && hipMemcpyAsync(P.w.p, vertexStates, sizeof(double) * (size_t)N * K, hipMemcpyHostToDevice, nullptr) != hipSuccess)
            return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: H2D copy of w failed"));
        (void)hipEventRecord(e0, nullptr);
        st = g4s_dense_rows_times_matrix(M, N, K, P.xx.as<double>(), P.w.as<double>(), P.res.as<double>(), nullptr);
        (void)hipEventRecord(e1, nullptr);
        if (st != G4S_OK) return finish(st);
        if (M && K && hipMemcpy(result, P.res.p, sizeof(double) * (size_t)M * K, hipMemcpyDeviceToHost) != hipSuccess)
            return finish(g4s::set_error(G4S_ERR_HIP, "spmm_dense: D2H copy of the result failed"));
        return finish(G4S_OK);
    }
    case G4S_PATTERN_SYM_QUADRATIC_FORM: {
        if (!edgeWeight || !edgeWeight[0] || !vertexStates || !result) return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: NULL argument"));
        (void)hipEventRecord(e0, nullptr);
        st = g4s_sym_quadratic_form((int)numNodes, P.desc.numbers, edgeWeight[0], vertexStates, temp, result);
        (void)hipEventRecord(e1, nullptr);
        return finish(st);
    }
    }
    return finish(g4s::set_error(G4S_ERR_INVALID, "spmm_dense: corrupt pattern"));
}

G4S_API void spmm_dense(uint32_t numNodes, uint32_t degree, const double **edgeWeight, const double *vertexStates,
                        double *temp, double *result, fun_gather gather, fun_apply apply, double *time, int threadNum)
{
    // The reference symbol returns void (citcoms/lib/global_defs.h:854-857): failures cannot be reported, so they are fatal.
    if (g4s_spmm_dense(numNodes, degree, edgeWeight, vertexStates, temp, result, gather, apply, time, threadNum) != G4S_OK) {
        fprintf(stderr, "g4s: spmm_dense failed: %s\n", g4s_last_error());
        abort();
    }
}
