// cg.hip — device-resident Jacobi-preconditioned conjugate gradient around the mat-vec.
//
// Follows conj_grad, citcoms/lib/General_matrix_functions.c:307-424, in its update order (z = BI∘r; β; p; Ap; α; d, r; residual)
// with global_vdot = plain dot product (single process: Global_operations.c:534-562 with no skipped halo equations) and
// assemble_del2_u(..., strip_bcs = 1) = mat-vec followed by zeroing the boundary equations (BC_util.c:89-102).
// The reference's own CUDA attempt copied p and Ap across PCIe every iteration and kept the dot products on the host
// (citcoms/lib/cgrad_kernel.cu:419-467); here every vector stays in HBM and each iteration is the mat-vec plus three fused
// vector kernels. Dot products are two-level (fixed grid → partials → serial sum in a fixed order): reproducible.
// Scalars (dot products, α, β, the residual, the iteration count) never leave the device, and neither does the termination test of
// the source: `cg_direction_kernel` evaluates `(residual > acc && count < steps) || count == 0` from the partial sums and, when it
// fails, raises a `done` flag that turns the rest of the enqueued iterations into no-ops. The host enqueues iterations in batches
// (sized from the previous solve and a geometric fit of the residual) and reads 40 bytes of state after each batch, so the GPU runs launches back to back instead of idling across a
// D2H round trip per iteration (62 → see DESIGN.md §4.4 µs per Cookbook2 iteration).
#include "common.hpp"
#include "readback.hpp"
#include "cg_async.hpp"
namespace g4s { int64_t dist_smallest_slab(g4s_spmv_dist_t h); }   // dist.hip
#include <algorithm>
#include <cmath>
#include <functional>

// opaque handle types of the two operators, defined in graph.hip / spmv.hip
extern "C" g4s_status g4s_elem_op_apply(g4s_elem_op_t op, const double *u_dev, double *Au_dev, void *stream);
extern "C" g4s_status g4s_spmv(g4s_csr_t A, const double *x_dev, double *y_dev, double alpha, double beta, void *stream);

namespace {

constexpr int kDotBlocks = 256;   // partial sums per dot product
constexpr int kThreads = 256;

// Workgroups of the vector kernels: one per 256 unknowns, at most kDotBlocks (= partial sums per dot product; unused slots stay zero).
inline int dot_blocks(int n) { const int b = (n + 255) / 256; return b < 1 ? 1 : (b > 256 ? 256 : b); }

struct CgState { double r1z1, r0z0, residual, residual0; int count, done; };   // residual0 = |F|, kept for the host's batch-size fit

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

// Σ of the kDotBlocks partial sums, identical in every workgroup: thread t takes partial t, then the fixed-shape block reduction
// (a serial loop per thread, the first version, is 256 dependent L2 round trips: ≈15 µs per dot product on an idle GPU)
__device__ __forceinline__ double sum_partials(const double *__restrict__ part, double *sh)
{
    static_assert(kDotBlocks == kThreads, "one partial per thread");
    return block_sum(part[threadIdx.x], sh);
}

// r1 = F; d0 = 0; z = BI∘r1; partial r1·r1 and r1·z   (General_matrix_functions.c:345-362, first pass of :365-367)
__global__ __launch_bounds__(kThreads) void cg_init_kernel(int n, const double *__restrict__ F, const double *__restrict__ BI, double *__restrict__ r1,
                                                            double *__restrict__ d0, double *__restrict__ z, double *__restrict__ part_rr,
                                                            double *__restrict__ part_rz, CgState *__restrict__ st)
{
    __shared__ double sh[4];
    // the state of a new solve (no other block of this kernel reads it; a memset launch less per solve)
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->r1z1 = 0.0; st->r0z0 = 0.0; st->residual = 0.0; st->residual0 = 0.0; st->count = 0; st->done = 0; }
    double rr = 0.0, rz = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const double f = F[i], zi = BI[i] * f;
        r1[i] = f; d0[i] = 0.0; z[i] = zi;
        rr += f * f;
        rz += f * zi;
    }
    rr = block_sum(rr, sh);
    rz = block_sum(rz, sh);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = rr; part_rz[blockIdx.x] = rz; }
    // slots of workgroups that do not exist stay zero (a multi-rank caller all-reduces all kDotBlocks slots in place between the steps)
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x) { part_rr[threadIdx.x] = 0.0; part_rz[threadIdx.x] = 0.0; }
}

// The loop head of :364 and the direction update of :369-379, on the device:
//   residual = sqrt(Σ part_rr); if !((residual > acc && count < steps) || count == 0) → done (every block reaches the same verdict
//   from the same sums; one thread records it); else dotr1z1 = Σ part_rz; p2 = z (count == 0) | z + (dotr1z1/dotr0z0)·p1.
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(int n, int steps, double acc, const double *__restrict__ part_rr,
                                                                 const double *__restrict__ part_rz, CgState *__restrict__ st,
                                                                 const double *__restrict__ z, const double *__restrict__ p1, double *__restrict__ p2)
{
    __shared__ double sh[4];
    if (st->done) return;
    const double residual = sqrt(sum_partials(part_rr, sh));
    const int count = st->count;
    const bool run = (residual > acc && count < steps) || count == 0;
    if (!run) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->residual = residual; st->done = 1; }
        return;
    }
    const double r1z1 = sum_partials(part_rz, sh);
    const bool first = count == 0;
    const double beta = first ? 0.0 : r1z1 / st->r0z0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) p2[i] = first ? z[i] : z[i] + beta * p1[i];
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { st->r1z1 = r1z1; st->residual = residual; if (first) st->residual0 = residual; }   // r0z0 is overwritten by cg_update_kernel, after every block has read it
}

// the verdict alone, once per batch, so that the host reads a final (residual, done) without enqueuing another iteration
__global__ __launch_bounds__(kThreads) void cg_peek_kernel(int steps, double acc, const double *__restrict__ part_rr, CgState *__restrict__ st)
{
    __shared__ double sh[4];
    if (st->done) return;
    const double residual = sqrt(sum_partials(part_rr, sh));
    if (threadIdx.x == 0) {
        st->residual = residual;
        if (!((residual > acc && st->count < steps) || st->count == 0)) st->done = 1;
    }
}

__global__ __launch_bounds__(kThreads) void cg_strip_kernel(int n_zero, const int *__restrict__ zero_resid, double *__restrict__ v)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n_zero) v[zero_resid[i]] = 0.0;
}

// boundary rows of Ap := 0 (strip_bcs_from_residual, through a byte mask built once per solve) fused with the partial p2·Ap
__global__ __launch_bounds__(kThreads) void cg_pAp_kernel(int n, const CgState *__restrict__ st, const unsigned char *__restrict__ bc_mask,
                                                           const double *__restrict__ p2, double *__restrict__ Ap, double *__restrict__ part)
{
    __shared__ double sh[4];
    if (st->done) return;
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        double a = Ap[i];
        if (bc_mask && bc_mask[i]) { a = 0.0; Ap[i] = 0.0; }
        acc += p2[i] * a;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x) part[threadIdx.x] = 0.0;
}

__global__ void cg_mask_kernel(int n_zero, const int *__restrict__ zero_resid, unsigned char *__restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_zero) mask[zero_resid[i]] = 1;
}

// alpha = dotprod == 0 ? 1e-3 : dotr1z1/dotprod; d0 += alpha·p2; r2 = r1 − alpha·Ap; partial r2·r2   (:383-394), and — so that
// the next iteration starts at its direction update — z = BI∘r2 with the partial r2·z of :365-367; count++, dotr0z0 := dotr1z1.
__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, const double *__restrict__ part_pAp, CgState *__restrict__ st,
                                                              const double *__restrict__ BI, const double *__restrict__ p2,
                                                              const double *__restrict__ Ap, const double *__restrict__ r1, double *__restrict__ r2,
                                                              double *__restrict__ d0, double *__restrict__ z, double *__restrict__ part_rr,
                                                              double *__restrict__ part_rz)
{
    __shared__ double sh[4];
    if (st->done) return;
    const double pAp = sum_partials(part_pAp, sh);
    const double r1z1 = st->r1z1;
    const double alpha = (pAp == 0.0) ? 1.0e-3 : r1z1 / pAp;
    double rr = 0.0, rz = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        d0[i] += alpha * p2[i];
        const double r = r1[i] - alpha * Ap[i];
        const double zi = BI[i] * r;
        r2[i] = r;
        z[i] = zi;
        rr += r * r;
        rz += r * zi;
    }
    rr = block_sum(rr, sh);
    rz = block_sum(rz, sh);
    if (threadIdx.x == 0) { part_rr[blockIdx.x] = rr; part_rz[blockIdx.x] = rz; }
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x) { part_rr[threadIdx.x] = 0.0; part_rz[threadIdx.x] = 0.0; }
    __syncthreads();
    // count and r0z0 are read by every block of this kernel only through `st->done` / `st->r1z1` above: safe to write here
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->r0z0 = r1z1; st->count = st->count + 1; }
}

__global__ __launch_bounds__(kThreads) void elem_inverse_diagonal_finish_kernel(int n, double *__restrict__ BI)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) BI[i] = BI[i] != 0.0 ? 1.0 / BI[i] : 0.0;
}

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

} // namespace

// defined in graph.hip (needs the operator's node→term map)
extern "C" g4s_status g4s_elem_op_diagonal_sum(g4s_elem_op_t op, double *diag_dev, void *stream);
int g4s_elem_op_neq(g4s_elem_op_t op);
int g4s_elem_op_apply_unless(g4s_elem_op_t op, const double *u_dev, double *Au_dev, const int *skip_dev, void *stream);
int g4s_node_op_apply_unless(g4s_node_op_t op, const double *u_dev, double *Au_dev, const int32_t *zero_resid_dev, int32_t n_zero, const int *skip_dev, void *stream);   // nodeop.hip

G4S_API g4s_status g4s_elem_op_inverse_diagonal(g4s_elem_op_t op, double *BI_dev, void *stream)
{
    G4S_REQUIRE(op && BI_dev, "NULL argument");
    int neq = 0;
    G4S_TRY(g4s_elem_op_diagonal_sum(op, BI_dev, stream));
    neq = g4s_elem_op_neq(op);
    if (neq > 0) hipLaunchKernelGGL(elem_inverse_diagonal_finish_kernel, dim3((neq + kThreads - 1) / kThreads), dim3(kThreads), 0, g4s::as_stream(stream), neq, BI_dev);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

namespace g4s {
// One conjugate-gradient solve as an object, so that a caller can enqueue its first batch of iterations, go on enqueuing work that uses the
// result SPECULATIVELY, and read the solve's state together with its own scalars in one host synchronisation (stokes.hip: the Uzawa loop was two
// synchronisations per outer iteration — this one and its nine scalars — with the host's launches and the device's execution taking turns).
//   start(batch)  arena, boundary mask, r/z/d0 initialised, `batch` iterations and the loop test enqueued; no synchronisation
//   read_state_async(&h) copies the state to the host (caller synchronises); complete(h) runs further batches (synchronising) until done;
//   finish() strips the boundary rows of d0 (conj_grad :409) — harmless to enqueue speculatively, enqueued again after a continued solve.
// Iterations past the one that meets the test are no-ops on the device, so a batch that overshoots costs launches, never results.
using MatVec = std::function<int(const double *, double *, const int *, hipStream_t)>;   // matvec(p, Ap, done_flag, stream); free to return at once when *done_flag != 0

// byte mask of the boundary equations (strip_bcs_from_residual through cg_pAp_kernel): neq bytes
int cg_build_mask(int32_t neq, const int32_t *zero_resid, int32_t n_zero, unsigned char *mask, hipStream_t s)
{
    G4S_HIP_TRY(hipMemsetAsync(mask, 0, (size_t)neq, s));
    if (n_zero) hipLaunchKernelGGL(cg_mask_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid, mask);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

struct CgRun {
    MatVec matvec;
    int neq = 0, n_zero = 0, steps = 0, enqueued = 0;
    const double *BI = nullptr;
    const int32_t *zero_resid = nullptr;
    double *d0 = nullptr, acc = 0.0;
    hipStream_t s = nullptr;
    void *arena = nullptr;
    double *r1 = nullptr, *r2 = nullptr, *z = nullptr, *p1 = nullptr, *p2 = nullptr, *Ap = nullptr, *part_rz = nullptr, *part_pAp = nullptr, *part_rr = nullptr;
    CgState *st = nullptr;
    unsigned char *bc_mask = nullptr;
    ~CgRun() { g4s::scratch_free(arena, s); }

    int enqueue(int todo)
    {
        for (int it = 0; it < todo; ++it) {
            hipLaunchKernelGGL(cg_direction_kernel, dim3(dot_blocks(neq)), dim3(kThreads), 0, s, neq, steps, acc, part_rr, part_rz, st, z, p1, p2);
            G4S_TRY(matvec(p2, Ap, &st->done, s));
            hipLaunchKernelGGL(cg_pAp_kernel, dim3(dot_blocks(neq)), dim3(kThreads), 0, s, neq, st, bc_mask, p2, Ap, part_pAp);
            hipLaunchKernelGGL(cg_update_kernel, dim3(dot_blocks(neq)), dim3(kThreads), 0, s, neq, part_pAp, st, BI, p2, Ap, r1, r2, d0, z, part_rr, part_rz);
            std::swap(r1, r2);      // the pointer rotation of General_matrix_functions.c:398-402 (irrelevant for no-op iterations: nothing reads them again)
            std::swap(p1, p2);
        }
        enqueued += todo;
        hipLaunchKernelGGL(cg_peek_kernel, dim3(1), dim3(kThreads), 0, s, steps, acc, part_rr, st);
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }

    int start(const MatVec &mv, int32_t neq_, const double *BI_, const int32_t *zero_resid_, int32_t n_zero_, const double *F, double *d0_, double acc_, int32_t steps_,
              int batch, hipStream_t stream, const unsigned char *shared_mask = nullptr)
    {
        G4S_REQUIRE(neq_ > 0 && BI_ && F && d0_, "bad argument");
        G4S_REQUIRE(n_zero_ >= 0 && (n_zero_ == 0 || zero_resid_), "zero_resid is NULL");
        matvec = mv; neq = neq_; BI = BI_; zero_resid = zero_resid_; n_zero = n_zero_; d0 = d0_; acc = acc_; steps = steps_; s = stream;
        const size_t nb = sizeof(double) * (size_t)neq;
        // one stream-ordered arena for the six work vectors, the partial sums, the state and the boundary mask: after the first solve
        // the pool hands the same pages back without a driver call (nine hipMalloc/hipFree pairs cost more than a short solve)
        const size_t nbp = (nb + 255) / 256 * 256;
        const size_t arena_bytes = 6 * nbp + sizeof(double) * 3 * kDotBlocks + 256 + (n_zero ? ((size_t)neq + 255) / 256 * 256 : 0);
        G4S_TRY(g4s::scratch_alloc(&arena, arena_bytes, s));
        char *base = static_cast<char *>(arena);
        r1 = reinterpret_cast<double *>(base); r2 = reinterpret_cast<double *>(base + nbp); z = reinterpret_cast<double *>(base + 2 * nbp);
        p1 = reinterpret_cast<double *>(base + 3 * nbp); p2 = reinterpret_cast<double *>(base + 4 * nbp); Ap = reinterpret_cast<double *>(base + 5 * nbp);
        part_rz = reinterpret_cast<double *>(base + 6 * nbp); part_pAp = part_rz + kDotBlocks; part_rr = part_pAp + kDotBlocks;
        st = reinterpret_cast<CgState *>(part_rr + kDotBlocks);
        static_assert(sizeof(CgState) <= 256, "state slot");
        // no memset: cg_init_kernel resets the state, and every kernel that writes partial sums zeroes the slots of workgroups that do not exist
        if (n_zero && shared_mask) bc_mask = const_cast<unsigned char *>(shared_mask);   // built once by the caller for all its solves (cg_build_mask)
        else if (n_zero) {
            bc_mask = reinterpret_cast<unsigned char *>(st) + 256;
            G4S_TRY(cg_build_mask(neq, zero_resid, n_zero, bc_mask, s));
        }
        hipLaunchKernelGGL(cg_init_kernel, dim3(dot_blocks(neq)), dim3(kThreads), 0, s, neq, F, BI, r1, d0, z, part_rr, part_rz, st);
        return enqueue(std::max(1, std::min(batch, steps + 1)));
    }

    int read_state_async(CgState *h) { G4S_HIP_TRY(g4s::read_small(h, st, sizeof(CgState), s)); return G4S_OK; }

    // h: the state after the batches enqueued so far (read by the caller after a synchronisation). Runs on until the loop test is met.
    int complete(CgState &h)
    {
        int batch = std::min(32, std::max(2, enqueued * 2));
        while (!h.done) {
            // next batch: the iterations a geometric fit of the residual history says are left (+1), between 2 and 32
            if (h.count > 0 && h.residual > acc && acc > 0.0 && h.residual0 > h.residual) {
                const double rate = std::log(h.residual / h.residual0) / h.count;        // < 0
                const double left = std::log(acc / h.residual) / rate;
                if (left > 0.0 && left < 1e6) batch = std::max(2, std::min(32, (int)std::ceil(left) + 1));
            }
            G4S_TRY(enqueue(std::max(1, std::min(batch, steps - enqueued + 1))));
            G4S_TRY(read_state_async(&h));
            G4S_HIP_TRY(g4s::reads_sync(s));
            batch = std::min(32, batch * 2);
        }
        return G4S_OK;
    }

    int finish()
    {
        if (n_zero) hipLaunchKernelGGL(cg_strip_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid, d0);   // :409
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }
};

// first batch: one more than the previous solve of this thread needed — consecutive velocity solves of an Uzawa iteration take nearly the same
// number of iterations, so the whole solve is usually one batch, one read-back and no wasted launches
int &cg_last_iterations() { static thread_local int last = 3; return last; }
int cg_first_batch()
{
    if (const char *e = getenv("G4S_CG_FIRST_BATCH")) return std::max(1, atoi(e));   // tests: 1 makes every longer solve outrun its first batch
    return std::max(2, std::min(32, cg_last_iterations() + 1));
}

MatVec cg_matvec_for(g4s_elem_op_t op, g4s_csr_t A)
{
    if (op) return [op](const double *p, double *Ap, const int *done, hipStream_t s) { return g4s_elem_op_apply_unless(op, p, Ap, done, s); };
    return [A](const double *p, double *Ap, const int *, hipStream_t s) { return g4s_spmv(A, p, Ap, 1.0, 0.0, s); };
}

// the synchronous solve: start, read, continue until done, strip, synchronise
int conj_grad_impl(const MatVec &matvec, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                   const double *F, double *d0, double acc, int32_t *cycles, double *residual_out, void *stream)
{
    G4S_REQUIRE(cycles, "cycles is NULL");
    hipStream_t s = g4s::as_stream(stream);
    CgRun run;
    G4S_TRY(run.start(matvec, neq, BI, zero_resid, n_zero, F, d0, acc, *cycles, cg_first_batch(), s));
    CgState h{};
    G4S_TRY(run.read_state_async(&h));
    G4S_HIP_TRY(g4s::reads_sync(s));
    G4S_TRY(run.complete(h));
    if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s conj_grad: %d iterations, %d enqueued, residual %.3e (acc %.3e)\n", h.count, run.enqueued, h.residual, acc);
    cg_last_iterations() = h.count;
    *cycles = h.count;
    G4S_TRY(run.finish());
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (residual_out) *residual_out = h.residual;
    return G4S_OK;
}
} // namespace g4s

namespace {
using g4s::MatVec;
using g4s::conj_grad_impl;
} // namespace
namespace g4s {
struct CgAsync {
    CgRun run;
    CgState h{};
    bool read_pending = false;   // an asynchronous copy into h has been enqueued and nobody has settled since: the object must outlive it
    hipStream_t s = nullptr;
};

int cg_async_start(CgAsync **out, g4s_elem_op_t op, g4s_csr_t A, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                   const double *F, double *d0, double acc, int32_t steps, hipStream_t s, const unsigned char *bc_mask)
{
    *out = nullptr;
    G4S_REQUIRE((op != nullptr) != (A != nullptr), "exactly one of op / A must be given");
    auto c = new (std::nothrow) CgAsync();
    if (!c) return set_error(G4S_ERR_NOMEM, "host allocation failed");
    int st = c->run.start(cg_matvec_for(op, A), neq, BI, zero_resid, n_zero, F, d0, acc, steps, cg_first_batch(), s, bc_mask);
    if (st == G4S_OK) st = c->run.finish();
    if (st != G4S_OK) { delete c; return st; }
    c->s = s;
    *out = c;
    return G4S_OK;
}

int cg_async_read(CgAsync *c) { c->read_pending = true; return c->run.read_state_async(&c->h); }

int cg_async_settle(CgAsync *c, bool *speculation_held, int32_t *cycles, double *residual)
{
    c->read_pending = false;                                       // (the caller has synchronised: that is the contract of settle)
    *speculation_held = c->h.done != 0;
    if (!c->h.done) {
        G4S_TRY(c->run.complete(c->h));
        G4S_TRY(c->run.finish());
    }
    cg_last_iterations() = c->h.count;
    if (cycles) *cycles = c->h.count;
    if (residual) *residual = c->h.residual;
    return G4S_OK;
}

// An error between read and settle (the caller's own work failed) frees the object while the copy into c->h may still be in flight: wait for it first.
void cg_async_free(CgAsync *c) { if (c && c->read_pending) (void)g4s::reads_sync(c->s); if (c) g4s::reads_forget(c, c + 1); delete c; }
} // namespace g4s


G4S_API g4s_status g4s_conj_grad(g4s_elem_op_t op, g4s_csr_t A, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                                 const double *F, double *d0, double acc, int32_t *cycles, double *residual_out, void *stream)
{
    G4S_REQUIRE((op != nullptr) != (A != nullptr), "exactly one of op / A must be given");
    return conj_grad_impl(g4s::cg_matvec_for(op, A), neq, BI, zero_resid, n_zero, F, d0, acc, cycles, residual_out, stream);
}

G4S_API g4s_status g4s_conj_grad_node(g4s_node_op_t op, int32_t neq, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                                      const double *F, double *d0, double acc, int32_t *cycles, double *residual_out, void *stream)
{
    G4S_REQUIRE(op, "op is NULL");
    // the boundary rows of Ap are zeroed by cg_pAp_kernel's mask, as for the other operators
    MatVec mv = [op](const double *p, double *Ap, const int *done, hipStream_t s) { return g4s_node_op_apply_unless(op, p, Ap, nullptr, 0, done, s); };
    return conj_grad_impl(mv, neq, BI, zero_resid, n_zero, F, d0, acc, cycles, residual_out, stream);
}

// ------------------------------------------------------------------------------------------------ step-wise form (multi-GPU)
// The same kernels with the loop opened up, for a row-partitioned operator: every rank holds its slab of the
// vectors; the 256 partial sums of each dot product are all-reduced element-wise by the caller between the steps (RCCL / gloo
// through torch.distributed — one 2 KiB or 4 KiB collective), after which every rank's kernels add the same 256 numbers in the same
// order and reach the same α, β and verdict. The caller also owns the mat-vec (exchange p, local SpMV into Ap).
//   begin → [all-reduce rr, rz] → direction → state (done?) → buffers: p → caller: Ap = A·p → reduce_pAp → [all-reduce pAp]
//   → update → [all-reduce rr, rz] → direction → …  → end
struct g4s_cg_ws_s {
    int n = 0;
    void *arena = nullptr;
    double *r1 = nullptr, *r2 = nullptr, *z = nullptr, *p1 = nullptr, *p2 = nullptr, *Ap = nullptr;
    double *part = nullptr;          // [rz | pAp | rr], kDotBlocks each
    CgState *st = nullptr;
    unsigned char *mask = nullptr;
    bool use_mask = false;
    // the boundary mask of the last g4s_cg_begin and what it was built from; hold_mask (g4s::cg_ws_hold_mask): the owner promises that list keeps its contents
    // for the workspace's life, so a begin with the same list does not rebuild it (a memset and a launch per solve of an Uzawa iteration)
    const int32_t *mask_src = nullptr;
    int mask_n = -1;
    bool hold_mask = false;
};

G4S_API g4s_status g4s_cg_ws_create(g4s_cg_ws_t *out, int32_t n_local)
{
    G4S_REQUIRE(out && n_local > 0, "bad argument");
    *out = nullptr;
    auto ws = new (std::nothrow) g4s_cg_ws_s();
    if (!ws) return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    ws->n = n_local;
    const size_t nbp = (sizeof(double) * (size_t)n_local + 255) / 256 * 256, mb = ((size_t)n_local + 255) / 256 * 256;
    const int rc = g4s::big_alloc(&ws->arena, 6 * nbp + sizeof(double) * 3 * kDotBlocks + 256 + mb);
    if (rc != G4S_OK) { delete ws; return rc; }
    char *b = static_cast<char *>(ws->arena);
    ws->r1 = reinterpret_cast<double *>(b); ws->r2 = reinterpret_cast<double *>(b + nbp); ws->z = reinterpret_cast<double *>(b + 2 * nbp);
    ws->p1 = reinterpret_cast<double *>(b + 3 * nbp); ws->p2 = reinterpret_cast<double *>(b + 4 * nbp); ws->Ap = reinterpret_cast<double *>(b + 5 * nbp);
    ws->part = reinterpret_cast<double *>(b + 6 * nbp);
    ws->st = reinterpret_cast<CgState *>(ws->part + 3 * kDotBlocks);
    ws->mask = reinterpret_cast<unsigned char *>(ws->st) + 256;
    // once, here: the partial sums and the state (g4s_cg_begin used to zero them at every solve; the kernels reset what they use — cg_init_kernel the state,
    // every kernel that writes partial sums the slots of workgroups that do not exist — and a multi-rank caller's first all-reduce must not meet never-written memory)
    if (hipMemset(ws->part, 0, sizeof(double) * 3 * kDotBlocks + 256) != hipSuccess) { (void)g4s::big_free(ws->arena); delete ws; return g4s::set_error(G4S_ERR_HIP, "g4s_cg_ws_create: hipMemset failed"); }
    *out = ws;
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_ws_destroy(g4s_cg_ws_t ws)
{
    if (ws) { (void)g4s::big_free(ws->arena); delete ws; }
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_begin(g4s_cg_ws_t ws, const double *F_dev, const double *BI_dev, double *d0_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                                void *stream)
{
    G4S_REQUIRE(ws && F_dev && BI_dev && d0_dev, "NULL argument");
    G4S_REQUIRE(n_zero >= 0 && (n_zero == 0 || zero_resid_dev), "zero_resid is NULL");
    hipStream_t s = g4s::as_stream(stream);
    ws->use_mask = n_zero > 0;
    if (n_zero && !(ws->hold_mask && ws->mask_src == zero_resid_dev && ws->mask_n == n_zero)) {
        G4S_HIP_TRY(hipMemsetAsync(ws->mask, 0, (size_t)ws->n, s));
        hipLaunchKernelGGL(cg_mask_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid_dev, ws->mask);
        ws->mask_src = zero_resid_dev; ws->mask_n = n_zero;
    }
    hipLaunchKernelGGL(cg_init_kernel, dim3(dot_blocks(ws->n)), dim3(kThreads), 0, s, ws->n, F_dev, BI_dev, ws->r1, d0_dev, ws->z, ws->part + 2 * kDotBlocks, ws->part, ws->st);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_direction(g4s_cg_ws_t ws, int32_t steps, double acc, void *stream)
{
    G4S_REQUIRE(ws, "ws is NULL");
    hipLaunchKernelGGL(cg_direction_kernel, dim3(dot_blocks(ws->n)), dim3(kThreads), 0, g4s::as_stream(stream), ws->n, steps, acc, ws->part + 2 * kDotBlocks, ws->part, ws->st,
                       ws->z, ws->p1, ws->p2);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_state(g4s_cg_ws_t ws, int32_t *count, int32_t *done, double *residual, void *stream)
{
    G4S_REQUIRE(ws, "ws is NULL");
    hipStream_t s = g4s::as_stream(stream);
    CgState h{};
    G4S_HIP_TRY(g4s::read_small(&h, ws->st, sizeof(CgState), s));
    G4S_HIP_TRY(g4s::reads_sync(s));
    if (count) *count = h.count;
    if (done) *done = h.done;
    if (residual) *residual = h.residual;
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_buffers(g4s_cg_ws_t ws, double **p_dev, double **Ap_dev, double **partials_dev)
{
    G4S_REQUIRE(ws, "ws is NULL");
    if (p_dev) *p_dev = ws->p2;
    if (Ap_dev) *Ap_dev = ws->Ap;
    if (partials_dev) *partials_dev = ws->part;
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_reduce_pAp(g4s_cg_ws_t ws, void *stream)
{
    G4S_REQUIRE(ws, "ws is NULL");
    hipLaunchKernelGGL(cg_pAp_kernel, dim3(dot_blocks(ws->n)), dim3(kThreads), 0, g4s::as_stream(stream), ws->n, ws->st, ws->use_mask ? ws->mask : nullptr, ws->p2, ws->Ap,
                       ws->part + kDotBlocks);
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_update(g4s_cg_ws_t ws, const double *BI_dev, double *d0_dev, void *stream)
{
    G4S_REQUIRE(ws && BI_dev && d0_dev, "NULL argument");
    hipLaunchKernelGGL(cg_update_kernel, dim3(dot_blocks(ws->n)), dim3(kThreads), 0, g4s::as_stream(stream), ws->n, ws->part + kDotBlocks, ws->st, BI_dev, ws->p2, ws->Ap,
                       ws->r1, ws->r2, d0_dev, ws->z, ws->part + 2 * kDotBlocks, ws->part);
    G4S_HIP_TRY(hipGetLastError());
    std::swap(ws->r1, ws->r2);      // the pointer rotation of General_matrix_functions.c:398-402
    std::swap(ws->p1, ws->p2);
    return G4S_OK;
}

G4S_API g4s_status g4s_cg_end(g4s_cg_ws_t ws, double *d0_dev, const int32_t *zero_resid_dev, int32_t n_zero, void *stream)
{
    G4S_REQUIRE(ws && d0_dev, "NULL argument");
    hipStream_t s = g4s::as_stream(stream);
    if (n_zero) hipLaunchKernelGGL(cg_strip_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, s, n_zero, zero_resid_dev, d0_dev);   // :409
    G4S_HIP_TRY(hipGetLastError());
    G4S_HIP_TRY(g4s::reads_sync(s));
    return G4S_OK;
}

// ------------------------------------------------------------------------------------------------ the whole distributed solve in C
// conj_grad on a row-partitioned operator with nothing but this library underneath: the product is g4s_spmv_dist_apply (packed exchange
// of the direction vector over RCCL, overlapped with the own-column part), the dot products' 256 partial sums are all-reduced
// element-wise by g4s_comm_allreduce_sum_f64 on the same communicator (every rank then adds the same 256 numbers in the same order
// and reaches the same α, β and verdict) — the two collectives of the CitcomS loop it replaces (Regional_parallel_related.c:744-789
// neighbour exchange, Global_operations.c:534-562 MPI_Allreduce). One host read of (count, done, residual) per iteration.
namespace {
g4s_status rccl_allreduce_cb(void *ctx, double *buf, int64_t count, void *stream) { return g4s_comm_allreduce_sum_f64(ctx, buf, count, stream); }
} // namespace

G4S_API g4s_status g4s_transport_rccl(void *comm, g4s_transport *out)
{
    G4S_REQUIRE(comm && out, "NULL argument");
    out->ctx = comm;
    out->allreduce_sum_f64 = rccl_allreduce_cb;
    out->exchange = nullptr;                                       // the handles' own RCCL wiring (g4s_spmv_dist_connect_rccl)
    return G4S_OK;
}

namespace g4s {
// one distributed product through a transport
int dist_product(g4s_spmv_dist_t A, const g4s_transport *tr, const double *x, double *y, void *stream)
{
    if (!tr->exchange) return g4s_spmv_dist_apply(A, x, y, stream);
    G4S_TRY(g4s_spmv_dist_begin(A, x, y, stream));
    G4S_TRY(tr->exchange(tr->ctx, A, stream));
    return g4s_spmv_dist_finish(A, y, stream);
}
} // namespace g4s

namespace g4s {
// The partitioned solve as an object (the counterpart of CgRun above): start() enqueues the set-up, the first batch of iterations and the loop test
// without a host wait; the caller may go on enqueuing work that uses d0 speculatively and read the state together with its own scalars
// (stokes.hip: g4s_stokes_uzawa_cg_dist — one host wait per outer iteration). Every rank reads the same all-reduced sums, so every rank sees
// the same (count, done, residual) and takes the same turn.
struct DistCgAsync {
    g4s_spmv_dist_t A = nullptr;
    const g4s_transport *tr = nullptr;
    g4s_cg_ws_t ws = nullptr;
    const double *BI = nullptr;
    const int32_t *zero_resid = nullptr;
    double *d0 = nullptr, *part = nullptr, acc = 0.0;
    int n_zero = 0, steps = 0, enqueued = 0;
    void *stream = nullptr;
    CgState h{};

    // Iterations past the one that meets the test are no-ops in the CG kernels; their product and all-reduces still run, on data nothing
    // reads again (the partial sums are rewritten by the next solve's first kernel).
    int enqueue(int todo)
    {
        double *p = nullptr, *Ap = nullptr;
        for (int it = 0; it < todo; ++it) {
            G4S_TRY(g4s_cg_direction(ws, steps, acc, stream));
            G4S_TRY(g4s_cg_buffers(ws, &p, &Ap, nullptr));
            G4S_TRY(g4s::dist_product(A, tr, p, Ap, stream));
            G4S_TRY(g4s_cg_reduce_pAp(ws, stream));
            G4S_TRY(tr->allreduce_sum_f64(tr->ctx, part + kDotBlocks, kDotBlocks, stream));
            G4S_TRY(g4s_cg_update(ws, BI, d0, stream));
            G4S_TRY(tr->allreduce_sum_f64(tr->ctx, part, 3 * kDotBlocks, stream));          // [0, 256) r·z and [512, 768) r·r; the middle third is rewritten before its next use
        }
        enqueued += todo;
        return g4s_cg_direction(ws, steps, acc, stream);                                    // the loop test behind the batch (a no-op once done)
    }
    int start(int batch, const double *F)
    {
        G4S_TRY(g4s_cg_begin(ws, F, BI, d0, zero_resid, n_zero, stream));
        G4S_TRY(g4s_cg_buffers(ws, nullptr, nullptr, &part));
        G4S_TRY(tr->allreduce_sum_f64(tr->ctx, part, 3 * kDotBlocks, stream));              // r·z and r·r of the start vector
        return enqueue(std::max(1, std::min(batch, steps + 1)));
    }
    bool read_pending = false;                                                              // see CgAsync
    int read() { read_pending = true; G4S_HIP_TRY(g4s::read_small(&h, ws->st, sizeof(CgState), g4s::as_stream(stream))); return G4S_OK; }
    int complete()                                                                          // h: read after a synchronisation
    {
        int batch = std::min(32, std::max(2, enqueued * 2));
        while (!h.done) {
            G4S_TRY(enqueue(std::max(1, std::min(batch, steps - enqueued + 1))));
            G4S_TRY(read());
            G4S_HIP_TRY(g4s::reads_sync(g4s::as_stream(stream)));
            read_pending = false;
            batch = std::min(32, batch * 2);
        }
        return G4S_OK;
    }
    int finish()                                                                            // conj_grad :409, no synchronisation
    {
        if (n_zero) hipLaunchKernelGGL(cg_strip_kernel, dim3((n_zero + kThreads - 1) / kThreads), dim3(kThreads), 0, g4s::as_stream(stream), n_zero, zero_resid, d0);
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }
};

int dist_cg_async_start(DistCgAsync **out, g4s_cg_ws_t ws, g4s_spmv_dist_t A, const g4s_transport *tr, const double *BI, const int32_t *zero_resid, int32_t n_zero,
                        const double *F, double *d0, double acc, int32_t steps, void *stream)
{
    *out = nullptr;
    auto c = new (std::nothrow) DistCgAsync();
    if (!c) return set_error(G4S_ERR_NOMEM, "host allocation failed");
    c->A = A; c->tr = tr; c->ws = ws; c->BI = BI; c->zero_resid = zero_resid; c->n_zero = n_zero; c->d0 = d0; c->acc = acc; c->steps = steps; c->stream = stream;
    int st = c->start(cg_first_batch(), F);
    if (st == G4S_OK) st = c->finish();
    if (st != G4S_OK) { delete c; return st; }
    *out = c;
    return G4S_OK;
}
void cg_ws_hold_mask(g4s_cg_ws_t ws, bool hold) { if (ws) { ws->hold_mask = hold; if (!hold) { ws->mask_src = nullptr; ws->mask_n = -1; } } }
int dist_cg_async_read(DistCgAsync *c) { return c->read(); }
int dist_cg_async_settle(DistCgAsync *c, bool *speculation_held, int32_t *cycles, double *residual)
{
    c->read_pending = false;
    *speculation_held = c->h.done != 0;
    if (!c->h.done) {
        G4S_TRY(c->complete());
        G4S_TRY(c->finish());
    }
    cg_last_iterations() = c->h.count;
    if (cycles) *cycles = c->h.count;
    if (residual) *residual = c->h.residual;
    return G4S_OK;
}
void dist_cg_async_free(DistCgAsync *c) { if (c && c->read_pending) (void)g4s::reads_sync(g4s::as_stream(c->stream)); if (c) g4s::reads_forget(c, c + 1); delete c; }
} // namespace g4s

G4S_API g4s_status g4s_conj_grad_dist_tr(g4s_spmv_dist_t A, const g4s_transport *tr, int32_t n_local, const double *BI_dev, const int32_t *zero_resid_dev,
                                         int32_t n_zero, const double *F_dev, double *d0_dev, double acc, int32_t steps, int32_t *cycles, double *residual,
                                         void *stream)
{
    G4S_REQUIRE(A && tr && tr->allreduce_sum_f64, "NULL argument");
    // every rank sees the same partition: an empty slab is refused by ALL ranks here, in front of the first collective (a rank that returned alone would leave
    // the others waiting in it)
    G4S_REQUIRE(g4s::dist_smallest_slab(A) > 0, "every rank of the partition must own at least one row");
    G4S_REQUIRE(BI_dev && F_dev && d0_dev, "NULL argument");
    G4S_REQUIRE(n_local > 0 && n_zero >= 0 && (n_zero == 0 || zero_resid_dev), "bad size");
    g4s_cg_ws_t ws = nullptr;
    G4S_TRY(g4s_cg_ws_create(&ws, n_local));
    // Iterations are enqueued in batches with ONE read of (count, done, residual) behind each batch — the first as long as the previous solve of
    // this thread plus one (g4s::cg_first_batch, as the single-GPU solve): with an RCCL transport nothing inside a batch waits for the host.
    auto run = [&]() -> int {
        g4s::DistCgAsync *c = nullptr;
        G4S_TRY(g4s::dist_cg_async_start(&c, ws, A, tr, BI_dev, zero_resid_dev, n_zero, F_dev, d0_dev, acc, steps, stream));
        struct Free { g4s::DistCgAsync *c; ~Free() { g4s::dist_cg_async_free(c); } } guard{c};
        G4S_TRY(g4s::dist_cg_async_read(c));
        G4S_HIP_TRY(g4s::reads_sync(g4s::as_stream(stream)));
        bool held = true;
        G4S_TRY(g4s::dist_cg_async_settle(c, &held, cycles, residual));
        G4S_HIP_TRY(g4s::reads_sync(g4s::as_stream(stream)));
        return G4S_OK;
    };
    const int st = run();
    (void)g4s_cg_ws_destroy(ws);
    return st;
}

G4S_API g4s_status g4s_conj_grad_dist(g4s_spmv_dist_t A, void *comm, int32_t n_local, const double *BI_dev, const int32_t *zero_resid_dev, int32_t n_zero,
                                      const double *F_dev, double *d0_dev, double acc, int32_t steps, int32_t *cycles, double *residual, void *stream)
{
    G4S_REQUIRE(A && comm, "NULL argument");
    g4s_transport tr;
    G4S_TRY(g4s_transport_rccl(comm, &tr));
    return g4s_conj_grad_dist_tr(A, &tr, n_local, BI_dev, zero_resid_dev, n_zero, F_dev, d0_dev, acc, steps, cycles, residual, stream);
}
