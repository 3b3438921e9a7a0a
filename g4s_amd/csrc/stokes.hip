// stokes.hip — the incompressibility (Uzawa / Schur-complement CG) iteration of CitcomS around the velocity solve.
//
// Follows solve_Ahat_p_fhat_CG, citcoms/lib/Stokes_flow_Incomp.c:188-452, in its update order, for the incompressible case
// (inv_gruneisen == 0, so initial_vel_residual :839-881 runs) with solve_del2_u's conjugate-gradient branch
// (General_matrix_functions.c:89-94) as the velocity solve — that is g4s_conj_grad (cg.hip) on the element-by-element operator.
// Operators: assemble_div_u / assemble_grad_p (Element_calculations.c:701-779) on the per-element gradient vectors
// g[e][p] = elt_del[e].g[p][0]; norms: global_v_norm2 / global_p_norm2 / global_div_norm2 / global_pdot
// (Global_operations.c:565-656), one process. Every vector stays in HBM; the host owns the outer loop's scalar logic
// (keep_iterating :150-162, the "two consecutive converging iterations" rule); δ, α and the norms are computed on the device and
// nine doubles come back once per outer iteration.
#include "common.hpp"
#include "readback.hpp"
#include "cg_async.hpp"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {

constexpr int kBlocks = 256, kThreads = 256;

__device__ __forceinline__ double block_sum(double v, double *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return s;
}

struct Sum3 { double a, b, c; };
struct Sum6 { double v[6]; };

// f(i) does the element-wise work of index i and returns up to three terms to be summed over i. Two-level sums with a fixed
// shape (grid-stride partials per workgroup → finish_sums_kernel): reproducible.
template <typename F>
__global__ __launch_bounds__(kThreads) void map_sum_kernel(int n, F f, double *__restrict__ part)
{
    __shared__ double sh[4];
    Sum3 acc{0.0, 0.0, 0.0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kBlocks * kThreads) {
        const Sum3 v = f(i);
        acc.a += v.a; acc.b += v.b; acc.c += v.c;
    }
    const double a = block_sum(acc.a, sh), b = block_sum(acc.b, sh), c = block_sum(acc.c, sh);
    if (threadIdx.x == 0) { part[blockIdx.x] = a; part[kBlocks + blockIdx.x] = b; part[2 * kBlocks + blockIdx.x] = c; }
}

// One workgroup: the three sums in a fixed shape, then ep(s0, s1, s2) on one thread — the scalar algebra of the outer loop (δ, α,
// the norms) stays on the device, in the slots of a small scalar array the following kernels read.
template <typename E>
__global__ __launch_bounds__(kThreads) void finish_sums_kernel(const double *__restrict__ part, E ep)
{
    __shared__ double sh[4];
    static_assert(kBlocks == kThreads, "one partial per thread");
    double r[3];
    for (int k = 0; k < 3; ++k) r[k] = block_sum(part[k * kBlocks + threadIdx.x], sh);
    if (threadIdx.x == 0) ep(r[0], r[1], r[2]);
}

// the same with six terms (the five norms at the end of an outer iteration in one pass instead of two)
template <typename F>
__global__ __launch_bounds__(kThreads) void map_sum6_kernel(int n, F f, double *__restrict__ part)
{
    __shared__ double sh[4];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += kBlocks * kThreads) {
        const Sum6 v = f(i);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += v.v[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double t = block_sum(acc[k], sh);
        if (threadIdx.x == 0) part[k * kBlocks + blockIdx.x] = t;
    }
}
template <typename E>
__global__ __launch_bounds__(kThreads) void finish_sums6_kernel(const double *__restrict__ part, E ep)
{
    __shared__ double sh[4];
    double r[6];
    for (int k = 0; k < 6; ++k) r[k] = block_sum(part[k * kBlocks + threadIdx.x], sh);
    if (threadIdx.x == 0) ep(r);
}

template <typename F>
__global__ __launch_bounds__(kThreads) void map_kernel(int n, F f)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) f(i);
}

// divU[e] = Σ_a (g0·U[j1] + g1·U[j2] + g2·U[j3]) in a order — the per-element sequence of assemble_div_u (bit-identical).
// Eight lanes per element: lane a forms the term of local node a (its gathers of U are independent of the other lanes'), then the
// terms are added in a = 0, 1, 2, … order through shuffles, as the source's loop adds them. One thread per element was a chain of
// 24 dependent gathers (11 µs for 8192 elements).
__global__ __launch_bounds__(kThreads) void div_u_kernel(int nel, int npe, int dof, const int *__restrict__ elem_eq, const double *__restrict__ g,
                                                          const double *__restrict__ U, double *__restrict__ divU)
{
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    const int e = idx >> 3, sub = idx & 7;
    const int n = npe * dof;
    const bool ok = e < nel;
    double s = 0.0;
    for (int a0 = 0; a0 < npe; a0 += 8) {                          // npe = 8 in CitcomS: one round
        const int a = a0 + sub;
        double t = 0.0;
        if (ok && a < npe)
            for (int d = 0; d < dof; ++d) {
                const double q = g[(size_t)e * n + a * dof + d] * U[elem_eq[(size_t)e * n + a * dof + d]];
                t = d == 0 ? q : t + q;
            }
        for (int k = 0; k < 8 && a0 + k < npe; ++k) s = s + __shfl(t, (threadIdx.x & ~7) + k, 64);   // every lane of the group keeps the same running sum
    }
    if (ok && sub == 0) divU[e] = s;
}

// gradP[eq(node, d)] = Σ over the node's (element, local node) terms, ascending element, of g·P[e], elements with P == 0 skipped:
// the sequence of additions assemble_grad_p makes into that equation (bit-identical). A gather: no atomics.
__global__ __launch_bounds__(kThreads) void grad_p_kernel(int nno, int npe, int dof, const int *__restrict__ node_ptr,
                                                           const int *__restrict__ node_terms, const int *__restrict__ node_eq,
                                                           const double *__restrict__ g, const double *__restrict__ P, double *__restrict__ gradP)
{
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= nno * dof) return;
    const int node = idx / dof, d = idx - node * dof, n = npe * dof;
    double s = 0.0;
    for (int t = node_ptr[node]; t < node_ptr[node + 1]; ++t) {
        const int term = node_terms[t], e = term / npe, a = term - e * npe;
        const double pe = P[e];
        if (pe == 0.0) continue;
        const double q = g[(size_t)e * n + a * dof + d] * pe;
        s = s + q;
    }
    gradP[node_eq[idx]] = s;
}

__global__ __launch_bounds__(kThreads) void zero_rows_kernel(int n_zero, const int *__restrict__ rows, double *__restrict__ v)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_zero) v[rows[i]] = 0.0;
}

// BPI[e] = 1 / Σ_p g[e][p]·(BI[eq(e,p)]·g[e][p])  (assemble_dAhatp_entry + build_diagonal_of_Ahat)
__global__ __launch_bounds__(kThreads) void pressure_precond_kernel(int nel, int n, const int *__restrict__ elem_eq, const double *__restrict__ g,
                                                                     const double *__restrict__ BI, double *__restrict__ BPI)
{
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= nel) return;
    double divU = 0.0;
    for (int p = 0; p < n; ++p) {
        const double ge = g[(size_t)e * n + p];
        const double gradP = BI[elem_eq[(size_t)e * n + p]] * ge;
        const double q = ge * gradP;
        divU = divU + q;
    }
    BPI[e] = divU != 0.0 ? 1.0 / divU : 1.0;
}

inline int grid_for(int n) { return (n + kThreads - 1) / kThreads; }

struct Scratch {
    void *p = nullptr; hipStream_t s = nullptr;
    ~Scratch() { g4s::scratch_free(p, s); }
};

} // namespace

G4S_API g4s_status g4s_elem_op_div_u(g4s_elem_op_t op, const double *g_dev, const double *U_dev, double *divU_dev, void *stream)
{
    g4s::ElemOpView v;
    G4S_TRY(g4s_elem_op_view(op, &v));
    G4S_REQUIRE(v.nel == 0 || (g_dev && U_dev && divU_dev), "NULL argument");
    if (v.nel) hipLaunchKernelGGL(div_u_kernel, dim3(grid_for(v.nel * 8)), dim3(kThreads), 0, g4s::as_stream(stream), v.nel, v.npe, v.dof, v.elem_eq, g_dev, U_dev, divU_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_elem_op_grad_p(g4s_elem_op_t op, const double *g_dev, const double *P_dev, double *gradP_dev,
                                      const int32_t *zero_resid_dev, int32_t n_zero, void *stream)
{
    g4s::ElemOpView v;
    G4S_TRY(g4s_elem_op_view(op, &v));
    G4S_REQUIRE(v.neq == 0 || gradP_dev, "NULL argument");
    G4S_REQUIRE(v.nel == 0 || (g_dev && P_dev), "NULL argument");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid_dev), "zero_resid is NULL");
    hipStream_t s = g4s::as_stream(stream);
    // equations no node owns receive nothing and must read 0 (the source zeroes gradP first)
    if (v.neq && (int64_t)v.nno * v.dof != v.neq) G4S_HIP_TRY(hipMemsetAsync(gradP_dev, 0, sizeof(double) * (size_t)v.neq, s));
    if (v.nno) hipLaunchKernelGGL(grad_p_kernel, dim3(grid_for(v.nno * v.dof)), dim3(kThreads), 0, s, v.nno, v.npe, v.dof, v.node_ptr, v.node_terms, v.node_eq, g_dev, P_dev, gradP_dev);
    if (n_zero) hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for(n_zero)), dim3(kThreads), 0, s, n_zero, zero_resid_dev, gradP_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_elem_op_pressure_preconditioner(g4s_elem_op_t op, const double *g_dev, const double *BI_dev, double *BPI_dev, void *stream)
{
    g4s::ElemOpView v;
    G4S_TRY(g4s_elem_op_view(op, &v));
    G4S_REQUIRE(v.nel == 0 || (g_dev && BI_dev && BPI_dev), "NULL argument");
    if (v.nel) hipLaunchKernelGGL(pressure_precond_kernel, dim3(grid_for(v.nel)), dim3(kThreads), 0, g4s::as_stream(stream), v.nel, v.npe * v.dof, v.elem_eq, g_dev, BI_dev, BPI_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_stokes_uzawa_cg(g4s_elem_op_t op, g4s_csr_t K_csr, const double *g, const double *BI, const double *BPI, const double *nmass,
                                       const double *area, double volume, const int32_t *zero_resid, int32_t n_zero, const double *FF,
                                       double *V, double *P, const g4s_stokes_params *prm, g4s_stokes_result *res, double *hist,
                                       int32_t hist_lines, void *stream)
{
    g4s::ElemOpView v;
    G4S_TRY(g4s_elem_op_view(op, &v));
    G4S_REQUIRE(prm && res, "params / result is NULL");
    G4S_REQUIRE(v.nel > 0 && v.neq > 0 && v.nno > 0, "empty operator");
    G4S_REQUIRE(g && BI && BPI && nmass && area && FF && V && P, "NULL argument");
    G4S_REQUIRE(volume > 0.0, "volume must be positive");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid), "zero_resid is NULL");
    G4S_REQUIRE(hist_lines >= 0 && (hist_lines == 0 || hist), "hist is NULL");
    hipStream_t s = g4s::as_stream(stream);
    const int neq = v.neq, nel = v.nel, nno = v.nno, dof = v.dof;
    const size_t nq = ((size_t)neq * 8 + 255) / 256 * 256, np = ((size_t)nel * 8 + 255) / 256 * 256;
    Scratch scr; scr.s = s;
    G4S_TRY(g4s::scratch_alloc(&scr.p, 3 * nq + 6 * np + sizeof(double) * (6 * kBlocks + 16), s));
    char *base = static_cast<char *>(scr.p);
    double *F = reinterpret_cast<double *>(base), *u1 = reinterpret_cast<double *>(base + nq), *tmp = reinterpret_cast<double *>(base + 2 * nq);
    double *r1 = reinterpret_cast<double *>(base + 3 * nq), *r2 = reinterpret_cast<double *>(base + 3 * nq + np),
           *z1 = reinterpret_cast<double *>(base + 3 * nq + 2 * np), *s1 = reinterpret_cast<double *>(base + 3 * nq + 3 * np),
           *s2 = reinterpret_cast<double *>(base + 3 * nq + 4 * np), *Fp = reinterpret_cast<double *>(base + 3 * nq + 5 * np);
    double *part = reinterpret_cast<double *>(base + 3 * nq + 6 * np), *sums_dev = part + 6 * kBlocks;
    const int *node_eq = v.node_eq;
    // V and P of the NEXT outer iteration are written beside the current ones (ping-pong): an iteration that was enqueued behind a velocity
    // solve which turns out not to have met its test can then simply be enqueued again, from unchanged inputs
    Scratch scr2; scr2.s = s;
    const size_t nm = ((size_t)neq + 255) / 256 * 256;
    G4S_TRY(g4s::scratch_alloc(&scr2.p, nq + np + nm, s));
    unsigned char *bc_mask = reinterpret_cast<unsigned char *>(static_cast<char *>(scr2.p) + nq + np);   // boundary-equation mask of every velocity solve of this call
    G4S_TRY(g4s::cg_build_mask(neq, zero_resid, n_zero, bc_mask, s));
    double *Vc = V, *Vn = reinterpret_cast<double *>(scr2.p), *Pc = P, *Pn = reinterpret_cast<double *>(static_cast<char *>(scr2.p) + nq);
    // G4S_STOKES_SYNC=1: wait for every velocity solve before enqueuing what follows it (the round-2 behaviour, for A/B)
    const bool speculate = !(getenv("G4S_STOKES_SYNC") && atoi(getenv("G4S_STOKES_SYNC")) != 0);

    // sc: device scalars of the loop. A reduction = map_sum over the vectors + a one-workgroup finish whose epilogue writes the
    // derived scalars; nothing comes to the host until fetch() at the end of an outer iteration.
    // <r1, z1> of the NEXT outer iteration is formed in the closing reduction of the current one (r2 is final there; round 5, as in g4s_stokes_uzawa_cg_dist): the two
    // slots of R1Z1 / DELTA alternate by the iteration's parity — the closing step may run twice (a speculation that did not hold) and must find the current pair untouched.
    enum { R1Z1_0, R1Z1_1, DELTA_0, DELTA_1, ALPHA, VDOTV, U1DOTU1, PDOTP, S2S2, DIVN, NSC };
    static_assert(NSC <= 16, "scalar slots");
    double *sc = sums_dev;
    double hsc[NSC] = {0};
    auto reduce = [&](int n, auto f, auto ep) {
        hipLaunchKernelGGL(map_sum_kernel, dim3(kBlocks), dim3(kThreads), 0, s, n, f, part);
        hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(kThreads), 0, s, part, ep);
    };
    auto fetch = [&]() -> int {
        G4S_HIP_TRY(hipGetLastError());
        G4S_HIP_TRY(g4s::read_small(hsc, sc, sizeof(hsc), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        return G4S_OK;
    };
    auto each = [&](int n, auto f) { hipLaunchKernelGGL(map_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, n, f); };
    auto strip = [&](double *x) { if (n_zero) hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for(n_zero)), dim3(kThreads), 0, s, n_zero, zero_resid, x); };
    auto v_terms = [=] __device__(const double *X, int i) {        // (Σ_d X[eq(i,d)]²)·NMass[i], the summand of global_v_norm2
        double t = 0.0;
        for (int d = 0; d < dof; ++d) { const double x = X[node_eq[i * dof + d]]; const double q = x * x; t = d == 0 ? q : t + q; }
        return t * nmass[i];
    };
    int64_t inner_total = 0;
    const double inner_acc = prm->imp * prm->inner_accuracy_scale * prm->v_res;
    auto solve_del2_u = [&](const double *rhs, double *d0, int *valid) -> int {   // General_matrix_functions.c:48-146, CG branch
        int32_t cycles = prm->v_steps_low;
        double residual = 0.0;
        G4S_TRY(g4s_conj_grad(K_csr ? nullptr : op, K_csr, neq, BI, zero_resid, n_zero, rhs, d0, inner_acc, &cycles, &residual, s));
        inner_total += cycles;
        *valid = residual < inner_acc ? 1 : 0;
        return G4S_OK;
    };

    // ---- initial_vel_residual (:839-881): F = FF − grad(P) − K·V, stripped; K·u1 = F; V += u1
    int valid = 0;
    G4S_TRY(g4s_elem_op_grad_p(op, g, P, u1, zero_resid, n_zero, s));
    each(neq, [=] __device__(int i) { F[i] = FF[i] - u1[i]; });
    if (K_csr) G4S_TRY(g4s_spmv(K_csr, V, u1, 1.0, 0.0, s));
    else G4S_TRY(g4s_elem_op_apply(op, V, u1, s));
    strip(u1);
    each(neq, [=] __device__(int i) { F[i] = F[i] - u1[i]; });
    strip(F);
    G4S_TRY(solve_del2_u(F, u1, &valid));
    strip(u1);
    each(neq, [=] __device__(int i) { V[i] = V[i] + u1[i]; });

    // ---- r1 = div(V); incompressibility = sqrt(|r1|²_div / (1e-32 + |V|²))
    G4S_TRY(g4s_elem_op_div_u(op, g, V, r1, s));
    G4S_HIP_TRY(hipMemsetAsync(sc, 0, sizeof(double) * NSC, s));
    hipLaunchKernelGGL(map_sum6_kernel, dim3(kBlocks), dim3(kThreads), 0, s, std::max(nno, nel), [=] __device__(int i) {
        Sum6 o{{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
        if (i < nno) o.v[0] = v_terms(V, i);
        if (i < nel) {
            o.v[1] = r1[i] * r1[i] / area[i]; o.v[2] = P[i] * P[i] * area[i];
            const double z = BPI[i] * r1[i];                          // z1 = BPI∘r1 and <r1, z1> of the first outer iteration (:296-318)
            z1[i] = z; o.v[3] = r1[i] * z;
        }
        return o;
    }, part);
    hipLaunchKernelGGL(finish_sums6_kernel, dim3(1), dim3(kThreads), 0, s, part, [=] __device__(const double (&r)[6]) {
        sc[VDOTV] = r[0]; sc[DIVN] = r[1]; sc[PDOTP] = r[2]; sc[R1Z1_0] = r[3];
    });
    G4S_TRY(fetch());
    double vdotv = hsc[VDOTV] / volume, pdotp = hsc[PDOTP] / volume;
    double incompressibility = std::sqrt(hsc[DIVN] / volume / (1e-32 + vdotv));
    double dvelocity = 1.0, dpressure = 1.0;
    int count = 0, converging = 0, lines = 0;
    auto record = [&]() {
        if (lines < hist_lines) { double *h = hist + 5 * (size_t)lines; h[0] = std::sqrt(vdotv); h[1] = std::sqrt(pdotp); h[2] = dvelocity; h[3] = dpressure; h[4] = incompressibility; }
        ++lines;
    };
    record();
    for (;;) {
        const bool keep = prm->check_continuity_convergence ? (incompressibility > prm->imp || converging < 2)
                                                            : (incompressibility > prm->imp && converging < 2);   // keep_iterating :150-162
        if (!(count < prm->steps_max && keep)) break;
        // z1 = BPI∘r1, <r1, z1> and δ = <r1, z1> / <r0, z0> (:296-318) sit in slot `cur`: written by the closing reduction of the previous iteration
        const int cur = count & 1, nxt = cur ^ 1;
        if (hsc[R1Z1_0 + cur] == 0.0) return g4s::set_error(G4S_ERR_INVALID, "g4s_stokes_uzawa_cg: <r1, z1> = 0 at the head of iteration %d (the source asserts)", count);
        const bool first = count == 0;
        each(nel, [=] __device__(int i) { s2[i] = first ? z1[i] : z1[i] + sc[DELTA_0 + cur] * s1[i]; });
        // K·u1 = grad(s2). The solve's first batch (one iteration more than the previous solve needed) is only enqueued; everything that follows it
        // in this outer iteration is enqueued behind it at once, and ONE synchronisation brings back the solve's state and the nine scalars.
        // Before round 3 the host waited for the solve, then enqueued the rest and waited again: at Cookbook2's size the loop was host-bound
        // (38 launches of 3–11 µs and two read-backs per outer iteration: 0.21 ms, of which the device needs 0.13 — tools/uzawa_graph_probe.py).
        G4S_TRY(g4s_elem_op_grad_p(op, g, s2, tmp, zero_resid, n_zero, s));
        g4s::CgAsync *cg = nullptr;
        G4S_TRY(g4s::cg_async_start(&cg, K_csr ? nullptr : op, K_csr, neq, BI, zero_resid, n_zero, tmp, u1, inner_acc, prm->v_steps_low, s, n_zero ? bc_mask : nullptr));
        struct CgFree { g4s::CgAsync *c; ~CgFree() { g4s::cg_async_free(c); } } cg_guard{cg};
        int32_t cycles = 0;
        double residual = 0.0;
        bool held = true;
        if (!speculate) {
            G4S_TRY(g4s::cg_async_read(cg));
            G4S_HIP_TRY(g4s::reads_sync(s));
            G4S_TRY(g4s::cg_async_settle(cg, &held, &cycles, &residual));
        }
        auto rest_of_iteration = [&]() -> int {
            G4S_TRY(g4s_elem_op_div_u(op, g, u1, Fp, s));
            // α = <r1, z1> / <s2, div(u1)>; r2, P, V (:336-354)
            reduce(nel, [=] __device__(int i) { return Sum3{s2[i] * Fp[i], 0.0, 0.0}; },
                   [=] __device__(double a, double, double) { sc[ALPHA] = sc[R1Z1_0 + cur] / a; });
            each(std::max(nel, neq), [=] __device__(int i) {        // one launch for the three updates
                const double alpha = sc[ALPHA];
                if (i < nel) { r2[i] = r1[i] - alpha * Fp[i]; Pn[i] = Pc[i] + alpha * s2[i]; }
                if (i < neq) Vn[i] = Vc[i] - alpha * u1[i];
            });
            G4S_TRY(g4s_elem_op_div_u(op, g, Vn, Fp, s));           // (Fp = div(u1) has gone into r2: the buffer is free)
            // the five norms of the iteration and, since r2 is final, z1 = BPI∘r2 and <r1, z1> of the NEXT one, in one two-level pass
            hipLaunchKernelGGL(map_sum6_kernel, dim3(kBlocks), dim3(kThreads), 0, s, std::max(nno, nel), [=] __device__(int i) {
                Sum6 o{{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
                if (i < nno) { o.v[0] = v_terms(Vn, i); o.v[1] = v_terms(u1, i); }
                if (i < nel) {
                    o.v[2] = Pn[i] * Pn[i] * area[i]; o.v[3] = s2[i] * s2[i] * area[i]; o.v[4] = Fp[i] * Fp[i] / area[i];
                    const double z = BPI[i] * r2[i];
                    z1[i] = z; o.v[5] = r2[i] * z;
                }
                return o;
            }, part);
            hipLaunchKernelGGL(finish_sums6_kernel, dim3(1), dim3(kThreads), 0, s, part, [=] __device__(const double (&r)[6]) {
                sc[VDOTV] = r[0]; sc[U1DOTU1] = r[1]; sc[PDOTP] = r[2]; sc[S2S2] = r[3]; sc[DIVN] = r[4];
                sc[R1Z1_0 + nxt] = r[5]; sc[DELTA_0 + nxt] = r[5] / sc[R1Z1_0 + cur];      // δ of the next iteration = its <r1, z1> over this one's (:296-318, :405)
            });
            return G4S_OK;
        };
        G4S_TRY(rest_of_iteration());
        if (speculate) {
            G4S_TRY(g4s::cg_async_read(cg));
            G4S_TRY(fetch());
            G4S_TRY(g4s::cg_async_settle(cg, &held, &cycles, &residual));
            if (!held) G4S_TRY(rest_of_iteration());              // the first batch did not meet the test: u1 is final only now — once more, from the same V, P, r1, s2
        }
        inner_total += cycles;
        valid = residual < inner_acc ? 1 : 0;
        if (!speculate || !held) G4S_TRY(fetch());                 // (speculation that held: the scalars came back with the solve's state)
        const double alpha = hsc[ALPHA];
        vdotv = hsc[VDOTV] / volume;
        pdotp = hsc[PDOTP] / volume;
        dvelocity = alpha * std::sqrt(hsc[U1DOTU1] / volume / (1e-32 + vdotv));
        dpressure = alpha * std::sqrt(hsc[S2S2] / volume / (1e-32 + pdotp));
        incompressibility = std::sqrt(hsc[DIVN] / volume / (1e-32 + vdotv));
        ++count;
        record();
        if (!valid) converging = 0;
        else if (prm->check_pressure_convergence) converging = (dvelocity < prm->imp && dpressure < prm->imp) ? converging + 1 : 0;
        else converging = dvelocity < prm->imp ? converging + 1 : 0;
        std::swap(s1, s2);
        std::swap(r1, r2);
        std::swap(Vc, Vn);
        std::swap(Pc, Pn);
    }
    if (Vc != V) {                                                 // an odd number of iterations: the result sits in the scratch pair
        G4S_HIP_TRY(hipMemcpyAsync(V, Vc, sizeof(double) * (size_t)neq, hipMemcpyDeviceToDevice, s));
        G4S_HIP_TRY(hipMemcpyAsync(P, Pc, sizeof(double) * (size_t)nel, hipMemcpyDeviceToDevice, s));
    }
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(g4s::reads_sync(s));
    res->outer_iterations = count;
    res->inner_iterations = inner_total;
    res->last_solve_valid = valid;
    res->incompressibility = incompressibility;
    res->v_norm = std::sqrt(vdotv);
    res->p_norm = std::sqrt(pdotp);
    res->dvelocity = dvelocity;
    res->dpressure = dpressure;
    return G4S_OK;
}

// ------------------------------------------------------------------------------------------------ the same iteration on a partitioned operator
namespace g4s { int dist_product(g4s_spmv_dist_t A, const g4s_transport *tr, const double *x, double *y, void *stream); }   // cg.hip

namespace {
// the raw local sums of a reduction, before the all-reduce (the derived scalars are formed after it, by one thread)
__global__ __launch_bounds__(kThreads) void finish_raw_kernel(const double *__restrict__ part, double *__restrict__ raw)
{
    __shared__ double sh[4];
    double r[3];
    for (int k = 0; k < 3; ++k) r[k] = block_sum(part[k * kBlocks + threadIdx.x], sh);
    if (threadIdx.x == 0) { raw[0] = r[0]; raw[1] = r[1]; raw[2] = r[2]; }
}
__global__ __launch_bounds__(kThreads) void finish_raw6_kernel(const double *__restrict__ part, double *__restrict__ raw)
{
    __shared__ double sh[4];
    for (int k = 0; k < 6; ++k) {
        const double r = block_sum(part[k * kBlocks + threadIdx.x], sh);
        if (threadIdx.x == 0) raw[k] = r;
    }
}
template <typename E>
__global__ void scalar_kernel(E ep) { if (threadIdx.x == 0 && blockIdx.x == 0) ep(); }
} // namespace

G4S_API g4s_status g4s_stokes_uzawa_cg_dist(g4s_spmv_dist_t K, g4s_spmv_dist_t D, g4s_spmv_dist_t Dt, const g4s_transport *tr, int32_t neq, int32_t nel,
                                            const double *BI, const double *BPI, const double *vmass, const double *area, double volume,
                                            const int32_t *zero_resid, int32_t n_zero, const double *FF, double *V, double *P,
                                            const g4s_stokes_params *prm, g4s_stokes_result *res, double *hist, int32_t hist_lines, void *stream)
{
    G4S_REQUIRE(K && D && Dt && tr && tr->allreduce_sum_f64 && prm && res, "NULL argument");
    G4S_REQUIRE(neq > 0 && nel > 0, "every rank must own at least one equation and one element");
    G4S_REQUIRE(BI && BPI && vmass && area && FF && V && P, "NULL argument");
    G4S_REQUIRE(volume > 0.0, "volume must be positive");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid), "zero_resid is NULL");
    G4S_REQUIRE(hist_lines >= 0 && (hist_lines == 0 || hist), "hist is NULL");
    hipStream_t s = g4s::as_stream(stream);
    const size_t nq = ((size_t)neq * 8 + 255) / 256 * 256, np = ((size_t)nel * 8 + 255) / 256 * 256;
    Scratch scr; scr.s = s;
    G4S_TRY(g4s::scratch_alloc(&scr.p, 3 * nq + 6 * np + sizeof(double) * (6 * kBlocks + 32), s));
    char *base = static_cast<char *>(scr.p);
    double *F = reinterpret_cast<double *>(base), *u1 = reinterpret_cast<double *>(base + nq), *tmp = reinterpret_cast<double *>(base + 2 * nq);
    double *r1 = reinterpret_cast<double *>(base + 3 * nq), *r2 = reinterpret_cast<double *>(base + 3 * nq + np),
           *z1 = reinterpret_cast<double *>(base + 3 * nq + 2 * np), *s1 = reinterpret_cast<double *>(base + 3 * nq + 3 * np),
           *s2 = reinterpret_cast<double *>(base + 3 * nq + 4 * np), *Fp = reinterpret_cast<double *>(base + 3 * nq + 5 * np);
    double *part = reinterpret_cast<double *>(base + 3 * nq + 6 * np), *sc = part + 6 * kBlocks, *raw = sc + 16;
    // Round 5: <r1, z1> of the NEXT outer iteration is formed in the closing reduction of the current one (r2 is final there), so the two slots of R1Z1 / DELTA
    // alternate by the iteration's parity — the closing step may run twice (a speculation that did not hold) and must find the current pair untouched.
    enum { R1Z1_0, R1Z1_1, DELTA_0, DELTA_1, ALPHA, VDOTV, U1DOTU1, PDOTP, S2S2, DIVN, NSC };
    double hsc[NSC] = {0};
    // a reduction: local partial sums in the fixed two-level shape → three raw sums → summed over the ranks → the derived scalars
    auto reduce = [&](int n, auto f, auto ep) -> int {
        hipLaunchKernelGGL(map_sum_kernel, dim3(kBlocks), dim3(kThreads), 0, s, n, f, part);
        hipLaunchKernelGGL(finish_raw_kernel, dim3(1), dim3(kThreads), 0, s, part, raw);
        G4S_TRY(tr->allreduce_sum_f64(tr->ctx, raw, 3, stream));
        hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(64), 0, s, ep);
        return G4S_OK;
    };
    // the same with six sums in ONE all-reduce message: everything an outer iteration can sum once its new V, P and r are known (round 5: was two reductions
    // at its end and one at the head of the next — three all-reduces, now one; Global_operations.c:534-656 reduces each norm on its own)
    auto reduce6 = [&](int n, auto f, auto ep) -> int {
        hipLaunchKernelGGL(map_sum6_kernel, dim3(kBlocks), dim3(kThreads), 0, s, n, f, part);
        hipLaunchKernelGGL(finish_raw6_kernel, dim3(1), dim3(kThreads), 0, s, part, raw);
        G4S_TRY(tr->allreduce_sum_f64(tr->ctx, raw, 6, stream));
        hipLaunchKernelGGL(scalar_kernel, dim3(1), dim3(64), 0, s, ep);
        return G4S_OK;
    };
    auto fetch = [&]() -> int {
        G4S_HIP_TRY(hipGetLastError());
        G4S_HIP_TRY(g4s::read_small(hsc, sc, sizeof(hsc), s));
        G4S_HIP_TRY(g4s::reads_sync(s));
        return G4S_OK;
    };
    auto each = [&](int n, auto f) { hipLaunchKernelGGL(map_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, n, f); };
    auto strip = [&](double *x) { if (n_zero) hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for(n_zero)), dim3(kThreads), 0, s, n_zero, zero_resid, x); };
    auto grad_p = [&](const double *p_in, double *out) -> int { G4S_TRY(g4s::dist_product(Dt, tr, p_in, out, stream)); strip(out); return G4S_OK; };   // assemble_grad_p + strip_bcs
    auto div_u = [&](const double *u_in, double *out) -> int { return g4s::dist_product(D, tr, u_in, out, stream); };                                   // assemble_div_u
    int64_t inner_total = 0;
    const double inner_acc = prm->imp * prm->inner_accuracy_scale * prm->v_res;
    // one CG workspace for every velocity solve of the call
    struct Ws { g4s_cg_ws_t w = nullptr; ~Ws() { (void)g4s_cg_ws_destroy(w); } } ws;
    G4S_TRY(g4s_cg_ws_create(&ws.w, neq));
    g4s::cg_ws_hold_mask(ws.w, true);                              // zero_resid is this call's own argument: the same list for every velocity solve
    // V and P of the NEXT outer iteration are written beside the current ones (ping-pong), as in g4s_stokes_uzawa_cg: an iteration enqueued behind a
    // velocity solve whose first batch turns out not to have met its test is enqueued again, from unchanged inputs
    Scratch scr2; scr2.s = s;
    G4S_TRY(g4s::scratch_alloc(&scr2.p, nq + np, s));
    double *Vc = V, *Vn = reinterpret_cast<double *>(scr2.p), *Pc = P, *Pn = reinterpret_cast<double *>(static_cast<char *>(scr2.p) + nq);
    const bool speculate = !(getenv("G4S_STOKES_SYNC") && atoi(getenv("G4S_STOKES_SYNC")) != 0);
    struct CgFree { g4s::DistCgAsync *c; ~CgFree() { g4s::dist_cg_async_free(c); } };

    // ---- initial_vel_residual (:839-881): F = FF − grad(P) − K·V, stripped; K·u1 = F; V += u1
    int valid = 0;
    G4S_TRY(grad_p(P, u1));
    each(neq, [=] __device__(int i) { F[i] = FF[i] - u1[i]; });
    G4S_TRY(g4s::dist_product(K, tr, V, u1, stream));
    strip(u1);
    each(neq, [=] __device__(int i) { F[i] = F[i] - u1[i]; });
    strip(F);
    {
        // this solve's result feeds sums whose inputs it also writes in place (V += u1): waited for, not speculated on — once per call
        g4s::DistCgAsync *cg = nullptr;
        G4S_TRY(g4s::dist_cg_async_start(&cg, ws.w, K, tr, BI, zero_resid, n_zero, F, u1, inner_acc, prm->v_steps_low, stream));
        CgFree guard{cg};
        G4S_TRY(g4s::dist_cg_async_read(cg));
        G4S_HIP_TRY(g4s::reads_sync(s));
        bool held = true;
        int32_t cycles = 0;
        double residual = 0.0;
        G4S_TRY(g4s::dist_cg_async_settle(cg, &held, &cycles, &residual));
        inner_total += cycles;
        valid = residual < inner_acc ? 1 : 0;
    }
    each(neq, [=] __device__(int i) { V[i] = V[i] + u1[i]; });

    G4S_TRY(div_u(V, r1));
    G4S_HIP_TRY(hipMemsetAsync(sc, 0, sizeof(double) * NSC, s));
    G4S_TRY(reduce6(std::max(neq, nel), [=] __device__(int i) {
        Sum6 o{{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
        if (i < neq) o.v[0] = V[i] * V[i] * vmass[i];
        if (i < nel) {
            o.v[1] = r1[i] * r1[i] / area[i]; o.v[2] = P[i] * P[i] * area[i];
            const double z = BPI[i] * r1[i];                          // z1 = BPI·r1 and <r1, z1> of the first outer iteration (:263-273)
            z1[i] = z; o.v[3] = r1[i] * z;
        }
        return o;
    }, [=] __device__() { sc[VDOTV] = raw[0]; sc[DIVN] = raw[1]; sc[PDOTP] = raw[2]; sc[R1Z1_0] = raw[3]; }));
    G4S_TRY(fetch());
    double vdotv = hsc[VDOTV] / volume, pdotp = hsc[PDOTP] / volume;
    double incompressibility = std::sqrt(hsc[DIVN] / volume / (1e-32 + vdotv));
    double dvelocity = 1.0, dpressure = 1.0;
    int count = 0, converging = 0, lines = 0;
    auto record = [&]() {
        if (lines < hist_lines) { double *h = hist + 5 * (size_t)lines; h[0] = std::sqrt(vdotv); h[1] = std::sqrt(pdotp); h[2] = dvelocity; h[3] = dpressure; h[4] = incompressibility; }
        ++lines;
    };
    record();
    for (;;) {
        const bool keep = prm->check_continuity_convergence ? (incompressibility > prm->imp || converging < 2)
                                                            : (incompressibility > prm->imp && converging < 2);   // keep_iterating :150-162
        if (!(count < prm->steps_max && keep)) break;              // every rank holds the same all-reduced scalars: the same verdict everywhere
        const int cur = count & 1, nxt = cur ^ 1;                  // this iteration's <r1, z1> and δ sit in slot `cur` (written by the closing reduction of the previous one)
        if (hsc[R1Z1_0 + cur] == 0.0) return g4s::set_error(G4S_ERR_INVALID, "g4s_stokes_uzawa_cg_dist: <r1, z1> = 0 at the head of iteration %d (the source asserts)", count);
        const bool first = count == 0;
        each(nel, [=] __device__(int i) { s2[i] = first ? z1[i] : z1[i] + sc[DELTA_0 + cur] * s1[i]; });
        G4S_TRY(grad_p(s2, tmp));
        // K·u1 = grad(s2): the solve's first batch is only enqueued (products, exchanges and all-reduces included — with an RCCL transport none of them
        // waits for the host); the rest of the outer iteration goes behind it at once and ONE synchronisation brings back the solve's state and the
        // nine scalars. Until round 4 this loop waited three times per outer iteration (the solve's state, the solve's end, the scalars).
        g4s::DistCgAsync *cg = nullptr;
        G4S_TRY(g4s::dist_cg_async_start(&cg, ws.w, K, tr, BI, zero_resid, n_zero, tmp, u1, inner_acc, prm->v_steps_low, stream));
        CgFree guard{cg};
        int32_t cycles = 0;
        double residual = 0.0;
        bool held = true;
        if (!speculate) {
            G4S_TRY(g4s::dist_cg_async_read(cg));
            G4S_HIP_TRY(g4s::reads_sync(s));
            G4S_TRY(g4s::dist_cg_async_settle(cg, &held, &cycles, &residual));
        }
        auto rest_of_iteration = [&]() -> int {
            G4S_TRY(div_u(u1, Fp));
            G4S_TRY(reduce(nel, [=] __device__(int i) { return Sum3{s2[i] * Fp[i], 0.0, 0.0}; }, [=] __device__() { sc[ALPHA] = sc[R1Z1_0 + cur] / raw[0]; }));
            each(std::max(nel, neq), [=] __device__(int i) {
                const double alpha = sc[ALPHA];
                if (i < nel) { r2[i] = r1[i] - alpha * Fp[i]; Pn[i] = Pc[i] + alpha * s2[i]; }
                if (i < neq) Vn[i] = Vc[i] - alpha * u1[i];
            });
            G4S_TRY(div_u(Vn, Fp));                                 // (Fp = D·u1 has gone into r2: the buffer is free)
            // the five norms of this iteration and, since r2 is final, z1 = BPI·r2 and <r1, z1> of the NEXT one — one all-reduce message of six sums
            G4S_TRY(reduce6(std::max(neq, nel), [=] __device__(int i) {
                Sum6 o{{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
                if (i < neq) { o.v[0] = Vn[i] * Vn[i] * vmass[i]; o.v[1] = u1[i] * u1[i] * vmass[i]; }
                if (i < nel) {
                    o.v[2] = Pn[i] * Pn[i] * area[i]; o.v[3] = s2[i] * s2[i] * area[i]; o.v[4] = Fp[i] * Fp[i] / area[i];
                    const double z = BPI[i] * r2[i];
                    z1[i] = z; o.v[5] = r2[i] * z;
                }
                return o;
            }, [=] __device__() {
                sc[VDOTV] = raw[0]; sc[U1DOTU1] = raw[1]; sc[PDOTP] = raw[2]; sc[S2S2] = raw[3]; sc[DIVN] = raw[4];
                sc[R1Z1_0 + nxt] = raw[5]; sc[DELTA_0 + nxt] = raw[5] / sc[R1Z1_0 + cur];
            }));
            return G4S_OK;
        };
        G4S_TRY(rest_of_iteration());
        if (speculate) {
            G4S_TRY(g4s::dist_cg_async_read(cg));
            G4S_TRY(fetch());
            G4S_TRY(g4s::dist_cg_async_settle(cg, &held, &cycles, &residual));
            if (!held) G4S_TRY(rest_of_iteration());              // u1 is final only now: once more, from the same V, P, r1, s2 (all ranks alike)
        }
        inner_total += cycles;
        valid = residual < inner_acc ? 1 : 0;
        if (!speculate || !held) G4S_TRY(fetch());
        const double alpha = hsc[ALPHA];
        vdotv = hsc[VDOTV] / volume;
        pdotp = hsc[PDOTP] / volume;
        dvelocity = alpha * std::sqrt(hsc[U1DOTU1] / volume / (1e-32 + vdotv));
        dpressure = alpha * std::sqrt(hsc[S2S2] / volume / (1e-32 + pdotp));
        incompressibility = std::sqrt(hsc[DIVN] / volume / (1e-32 + vdotv));
        ++count;
        record();
        if (!valid) converging = 0;
        else if (prm->check_pressure_convergence) converging = (dvelocity < prm->imp && dpressure < prm->imp) ? converging + 1 : 0;
        else converging = dvelocity < prm->imp ? converging + 1 : 0;
        std::swap(s1, s2);
        std::swap(r1, r2);
        std::swap(Vc, Vn);
        std::swap(Pc, Pn);
    }
    if (Vc != V) {                                                 // an odd number of iterations: the result sits in the scratch pair
        G4S_HIP_TRY(hipMemcpyAsync(V, Vc, sizeof(double) * (size_t)neq, hipMemcpyDeviceToDevice, s));
        G4S_HIP_TRY(hipMemcpyAsync(P, Pc, sizeof(double) * (size_t)nel, hipMemcpyDeviceToDevice, s));
    }
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(g4s::reads_sync(s));
    res->outer_iterations = count;
    res->inner_iterations = inner_total;
    res->last_solve_valid = valid;
    res->incompressibility = incompressibility;
    res->v_norm = std::sqrt(vdotv);
    res->p_norm = std::sqrt(pdotp);
    res->dvelocity = dvelocity;
    res->dpressure = dpressure;
    return G4S_OK;
}

