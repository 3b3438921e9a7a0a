// spmv_bcsr.hip — block-row SpMV for assembled finite-element matrices handed to g4s_csr_create as plain CSR (north_star: "the dense-tile
// fallback when a row block is effectively dense").
//
// An assembled FE stiffness matrix with `b` unknowns per node (CitcomS: 3; citcoms/lib/Construct_arrays.c:254-328 is that structure) is a
// matrix of dense b×b blocks: the b rows of a node hold the same columns, and those columns come in aligned runs of b. A mat-vec has
// 2 flops per 12-byte entry, so a dense block gains nothing from the matrix cores — what it buys is the column index: ONE block-column id
// per b×b block instead of b² column ids (b = 3: 8.44 B per entry instead of 12). This file detects that structure on the device at
// g4s_csr_create (every block-row: equal lengths, identical column lists, aligned runs), stores the values block-major with one int per
// block, and runs the same kind of kernel as the row-streaming CSR path: a workgroup streams a run of whole block-rows (≤ 2048 entries,
// coalesced, nontemporal), gathers x (b consecutive entries per block), parks the products in LDS, and ONE LANE PER ROW adds its row's
// products left to right — block by block, column by column inside a block: the stored order of the CSR row, multiply then add, so every
// row is bit-identical to the oracle (the CSR kernel reduces rows this long with several lanes and shuffles).
#include "common.hpp"
#include "spmv_bcsr.hpp"
#include <algorithm>
#include <memory>
#include <vector>

namespace g4s {
namespace {

#ifndef G4S_BCSR_TILE
#define G4S_BCSR_TILE 1024
#endif
// entries (8 KiB of products) per workgroup: see the note behind the kernel for the sizes measured
constexpr int kWG = 256, kTile = G4S_BCSR_TILE, kUnroll = kTile / kWG, kMaxBrows = 512;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n)
    {
        if (p) { (void)hipFree(p); p = nullptr; }
        const hipError_t e = g4s::device_malloc(&p, n ? n : 1);
        if (e != hipSuccess) return set_error(e == hipErrorOutOfMemory ? G4S_ERR_NOMEM : G4S_ERR_HIP, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
        bytes = n;
        return G4S_OK;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// fail |= 1 unless block-row n (rows bn … bn+b−1) is b equal-length rows with identical, aligned column runs
__global__ void bcsr_check_kernel(int nbr, int b, int tile_blocks, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colids, int *__restrict__ fail)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nbr) return;
    const int r0 = n * b, k0 = rowptr[r0], L = rowptr[r0 + 1] - k0;
    bool ok = L % b == 0 && L / b <= tile_blocks;
    for (int d = 1; d < b && ok; ++d) ok = rowptr[r0 + d + 1] - rowptr[r0 + d] == L && rowptr[r0 + d] == k0 + d * L;
    for (int k = 0; k < L && ok; ++k) {
        const int c = colids[k0 + k];
        ok = c == (c / b) * b + (k % b) && (k % b != 0 || k == 0 || c > colids[k0 + k - 1]);   // aligned run; block columns ascending
        for (int d = 1; d < b && ok; ++d) ok = colids[k0 + d * L + k] == c;
    }
    if (!ok) atomicOr(fail, 1);
}

// one thread per row: values into block-major order, the block-column ids from the block-row's first row
__global__ void bcsr_fill_kernel(int rows, int b, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colids, const double *__restrict__ values,
                                 double *__restrict__ bval, int32_t *__restrict__ bcol)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int n = r / b, d = r - n * b, bb = b * b;
    const int kb = rowptr[n * b], k0 = rowptr[r], L = rowptr[r + 1] - k0;
    for (int k = 0; k < L; ++k) {
        const int j = k / b, c = k - j * b;
        bval[(size_t)kb + (size_t)j * bb + d * b + c] = values[k0 + k];
        if (d == 0 && c == 0) bcol[kb / bb + j] = colids[k0 + k] / b;
    }
}

struct Item { int brow0, nbrows, blk0, nblk; };   // a run of whole block-rows: ≤ kTile entries, ≤ kMaxBrows block-rows

template <int B, bool NT>
__global__ __launch_bounds__(kWG) void spmv_bcsr_kernel(const Item *__restrict__ items, const int32_t *__restrict__ rowptr /* of the CSR matrix: block-row n starts at block rowptr[B·n] / B² */,
                                                         const int32_t *__restrict__ bcol, const double *__restrict__ bval, const double *__restrict__ x,
                                                         double *__restrict__ y, double alpha, double beta)
{
    constexpr int BB = B * B;
    __shared__ double prod[kTile];
    __shared__ int first_blk[kMaxBrows + 1];
    const Item it = items[blockIdx.x];
    const int tid = threadIdx.x, ne = it.nblk * BB;
    for (int i = tid; i <= it.nbrows; i += kWG) first_blk[i] = rowptr[(it.brow0 + i) * B] / BB - it.blk0;
    if (ne > 0) {
        // branch-free loads (lanes past the end re-read the last entry): all value / block-id loads, then all gathers, in flight together
        const long long e0 = (long long)it.blk0 * BB;
        const int last = ne - 1;
        double v[kUnroll];
        int col[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int e = min(u * kWG + tid, last);
            v[u] = NT ? __builtin_nontemporal_load(bval + e0 + e) : bval[e0 + e];
            const int blk = e / BB;
            col[u] = bcol[it.blk0 + blk] * B + (e - blk * BB) % B;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = v[u] * x[col[u]];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) prod[u * kWG + tid] = v[u];   // slots past ne are written, never read
    }
    __syncthreads();
    // one lane per row; products of row (n, d): blocks of n in order, inside a block the B entries of line d — the CSR row's stored order
    for (int r = tid; r < it.nbrows * B; r += kWG) {
        const int n = r / B, d = r - n * B;
        const int j0 = first_blk[n], j1 = first_blk[n + 1];
        double s = 0.0;
        int j = j0;
        for (; j + 4 <= j1; j += 4) {                              // four blocks' products read together, then added in order: the chain waits for LDS once per 4·B adds
            double t[4 * B];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < B; ++c) t[q * B + c] = prod[(j + q) * BB + d * B + c];
#pragma unroll
            for (int q = 0; q < 4 * B; ++q) s += t[q];
        }
        for (; j < j1; ++j) {
#pragma unroll
            for (int c = 0; c < B; ++c) s += prod[j * BB + d * B + c];
        }
        const int row = it.brow0 * B + r;
        y[row] = beta == 0.0 ? alpha * s : alpha * s + beta * y[row];
    }
}

// Round 4 measured a persistent, register-double-buffered form of this kernel (a few workgroups per CU walking the items with a stride, the loads of the next
// item issued before the row sums of the current one — VERDICT r3 item 6) on the Cookbook2 matrix, one process per build: 8.8 µs with two workgroups per CU,
// 7.7 with three, 7.0 with four, 12.7 with one, against 6.8 for this one-item-per-workgroup form — the product is bound by the latency of ONE dependent pair of
// loads (block id → x) per workgroup and by nothing else, so the more workgroups are in flight at once the better, and a loop over items only serialises them.
// What did help: 1 024-entry tiles (four block-rows per workgroup, twice the workgroups): 6.4 µs (1 280 and 1 536: the same; 768: 7.1; 4 096: 7.3).
} // namespace

struct BcsrPlan {
    int b = 0, rows = 0, n_items = 0;
    bool use_nt = true;
    const int32_t *d_rowptr = nullptr;   // borrowed from the CSR handle
    DevBuf bval, bcol, items;
    long long bytes = 0;
};

int bcsr_try_build(BcsrPlan **out, int rows, int cols, long long nnz, const int32_t *d_rowptr, const int32_t *d_colids, const double *d_values, bool use_nt)
{
    *out = nullptr;
    if (rows < 64 || nnz <= 0) return G4S_OK;
    for (int b : {3, 2, 4}) {
        const int bb = b * b;
        if (rows % b || cols % b || nnz % bb || nnz < 2ll * b * rows) continue;   // at least two blocks per block-row on average: otherwise the CSR kernel is as good
        const int nbr = rows / b;
        DevBuf fail;
        G4S_TRY(fail.alloc(sizeof(int)));
        G4S_HIP_TRY(hipMemset(fail.p, 0, sizeof(int)));
        hipLaunchKernelGGL(bcsr_check_kernel, dim3((nbr + 255) / 256), dim3(256), 0, nullptr, nbr, b, kTile / bb, d_rowptr, d_colids, fail.as<int>());
        int h_fail = 0;
        G4S_HIP_TRY(hipGetLastError());
        G4S_HIP_TRY(hipMemcpy(&h_fail, fail.p, sizeof(int), hipMemcpyDeviceToHost));
        if (h_fail) continue;
        auto P = std::make_unique<BcsrPlan>();
        P->b = b; P->rows = rows; P->use_nt = use_nt; P->d_rowptr = d_rowptr;
        G4S_TRY(P->bval.alloc(sizeof(double) * (size_t)nnz));
        G4S_TRY(P->bcol.alloc(sizeof(int32_t) * (size_t)(nnz / bb)));
        hipLaunchKernelGGL(bcsr_fill_kernel, dim3((rows + 255) / 256), dim3(256), 0, nullptr, rows, b, d_rowptr, d_colids, d_values, P->bval.as<double>(), P->bcol.as<int32_t>());
        G4S_HIP_TRY(hipGetLastError());
        // work items on the host: whole block-rows, ≤ kTile entries, ≤ kMaxBrows block-rows (rowptr of the block-rows' first rows only)
        std::vector<int32_t> rp((size_t)rows + 1);
        G4S_HIP_TRY(hipMemcpy(rp.data(), d_rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
        std::vector<Item> items;
        for (int n = 0; n < nbr;) {
            const int blk0 = rp[(size_t)n * b] / bb;
            int m = n;
            while (m < nbr && m - n < kMaxBrows && (rp[(size_t)(m + 1) * b] / bb - blk0) * bb <= kTile) ++m;
            if (m == n) return set_error(G4S_ERR_INVALID, "bcsr: a block-row exceeds the tile (checked before: internal error)");
            items.push_back(Item{n, m - n, blk0, rp[(size_t)m * b] / bb - blk0});
            n = m;
        }
        P->n_items = (int)items.size();
        G4S_TRY(P->items.alloc(sizeof(Item) * items.size()));
        G4S_HIP_TRY(hipMemcpy(P->items.p, items.data(), sizeof(Item) * items.size(), hipMemcpyHostToDevice));
        G4S_HIP_TRY(hipDeviceSynchronize());
        P->bytes = (long long)(P->bval.bytes + P->bcol.bytes + P->items.bytes);
        if (getenv("G4S_DEBUG")) fprintf(stderr, "g4s block-row SpMV plan: %d x %d blocks, %lld blocks, %d work items, %.1f MB\n", b, b, nnz / bb, P->n_items, P->bytes / 1e6);
        *out = P.release();
        return G4S_OK;
    }
    return G4S_OK;
}

// new values into the block-major copy (the pattern is the plan's: the block-column ids are rewritten with what they already hold)
int bcsr_update_values(BcsrPlan *P, const int32_t *d_colids, const double *d_values, hipStream_t s)
{
    hipLaunchKernelGGL(bcsr_fill_kernel, dim3((P->rows + 255) / 256), dim3(256), 0, s, P->rows, P->b, P->d_rowptr, d_colids, d_values, P->bval.as<double>(), P->bcol.as<int32_t>());
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

void bcsr_destroy(BcsrPlan *P) { delete P; }
long long bcsr_bytes(const BcsrPlan *P) { return P ? P->bytes : 0; }
int bcsr_block(const BcsrPlan *P) { return P ? P->b : 0; }

int bcsr_spmv(BcsrPlan *P, const double *x, double *y, double alpha, double beta, hipStream_t s)
{
    if (!P->n_items) return G4S_OK;

#define G4S_BCSR_LAUNCH(B)                                                                                                                              \
    do {                                                                                                                                               \
        if (P->use_nt) hipLaunchKernelGGL((spmv_bcsr_kernel<B, true>), dim3(P->n_items), dim3(kWG), 0, s, P->items.as<Item>(), P->d_rowptr, P->bcol.as<int32_t>(), P->bval.as<double>(), x, y, alpha, beta); \
        else hipLaunchKernelGGL((spmv_bcsr_kernel<B, false>), dim3(P->n_items), dim3(kWG), 0, s, P->items.as<Item>(), P->d_rowptr, P->bcol.as<int32_t>(), P->bval.as<double>(), x, y, alpha, beta);         \
    } while (0)
    if (P->b == 3) G4S_BCSR_LAUNCH(3);
    else if (P->b == 2) G4S_BCSR_LAUNCH(2);
    else G4S_BCSR_LAUNCH(4);
#undef G4S_BCSR_LAUNCH
    G4S_HIP_TRY(hipGetLastError());
    return G4S_OK;
}

} // namespace g4s
