// dist.hip — row-partitioned SpMV over several GPUs behind the C-ABI (SURVEY.md §8b B2 "g4s_spmv_dist_*", §8e): one process per GPU,
// RCCL point-to-point over xGMI for the x entries a rank's rows reference, the own-column product overlapped with that exchange.
//
// The reference has no multi-device code on this path. Its distributed pattern — the one a CitcomS-shaped caller already has — is the
// per-mat-vec neighbour exchange of citcoms/lib/Regional_parallel_related.c:744-789 (pack the shared entries, MPI_Sendrecv with each
// neighbour, unpack) and the dot-product all-reduce of citcoms/lib/Global_operations.c:534-562; the row split over ranks is the
// equal-work rule of mm/inc/BIN.h:101-122. Here:
//   * rank r owns rows [off[r], off[r+1]) of A and the same slab of x and y (square operator, 1-D row partition);
//   * at create the local rows are cut into A_own (columns inside the own slab, renumbered to the slab) and A_rem (all other columns,
//     renumbered 0…n_ref−1 in ascending global order: only REFERENCED columns exist — the halo planes of a stencil, ≈20 % of x for
//     an eighth of the R-MAT matrix). A peer k's referenced entries form one contiguous segment of the compact x_rem;
//   * every owner learns once which of its entries each peer wants ("give" lists; over RCCL by g4s_spmv_dist_connect_rccl, or by
//     any transport the caller has through g4s_spmv_dist_want / _set_give);
//   * a product: pack (one gather kernel fills every peer's send segment) → exchange on a side stream (ncclGroupStart, one
//     ncclSend + ncclRecv per peer with traffic, ncclGroupEnd: each pair over its own xGMI link, received straight into its
//     segment of x_rem, no unpack) ‖ y = A_own·x_local on the caller's stream → wait → y += A_rem·x_rem.
// RCCL is loaded with dlopen at first use (librccl.so.1: the copy the host framework already mapped, or ROCm's), so libg4s_hip.so has
// no link-time dependency on it and single-GPU users never touch it.
//   * G4S_DIST_ALLGATHER (north_star's "RCCL all-gather of the dense vector"; also what bench.py falls back to): no index lists at all —
//     every rank's slab of x, padded to the longest slab, lands in one buffer of world·pad entries by a single in-place ncclAllGather;
//     remote column c of owner k is renumbered k·pad + (c − off[k]). More bytes than the packed exchange (the whole vector travels to
//     everyone), one collective instead of 2·(world − 1) messages, and nothing to wire at set-up.
// The split itself (own / remote columns, renumbering, want lists) is host-side set-up logic and is exported as g4s_dist_split_rows
// (no GPU needed: a host with its own transport can use it, and the CPU tests run it under gloo); so is the equal-work row partition
// g4s_row_partition (mm/inc/BIN.h:101-122).
#include "common.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------- RCCL entry points, resolved lazily
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;   // optional
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                            // optional
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return G4S_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return g4s::set_error(G4S_ERR_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    Rccl r;
    r.lib = h;
#define G4S_RCCL_SYM(field, sym)                                                               \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, sym));                              \
    if (!r.field) return g4s::set_error(G4S_ERR_UNSUPPORTED, "RCCL symbol %s is missing", sym);
    G4S_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") G4S_RCCL_SYM(CommInitRank, "ncclCommInitRank") G4S_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    G4S_RCCL_SYM(Send, "ncclSend") G4S_RCCL_SYM(Recv, "ncclRecv") G4S_RCCL_SYM(AllReduce, "ncclAllReduce")
    G4S_RCCL_SYM(GroupStart, "ncclGroupStart") G4S_RCCL_SYM(GroupEnd, "ncclGroupEnd") G4S_RCCL_SYM(GetErrorString, "ncclGetErrorString")
    G4S_RCCL_SYM(AllGather, "ncclAllGather")
#undef G4S_RCCL_SYM
    r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl = r;
    return G4S_OK;
}

#define G4S_RCCL_TRY(expr)                                                                                           \
    do {                                                                                                             \
        ncclResult_t r__ = (expr);                                                                                   \
        if (r__ != ncclSuccess) return g4s::set_error(G4S_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r__)); \
    } while (0)

// send[i] = x[idx[i]]: the packed entries of every peer, one launch
__global__ void dist_pack_kernel(long long n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ send)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) send[i] = x[idx[i]];
}

// test hook: keeps a stream busy for a bounded number of 100 MHz ticks, then leaves
__global__ void dist_stall_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// y[rows[i]] += t[i]: the remote-column part of the few rows that have one
__global__ void dist_add_rows_kernel(int n, const int32_t *__restrict__ rows, const double *__restrict__ t, double *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[rows[i]] += t[i];
}

} // namespace

struct g4s_spmv_dist_s {
    int rank = 0, world = 1;
    bool loopback = false;
    int nseg = 1;                               // segments of the exchange buffers: one per rank — or, in loopback, G4S_DIST_LOOPBACK_PEERS (default 7) pieces of
                                                // the "remote" half of the slab, each travelling rank 0 → rank 0 as its own ncclSend / ncclRecv pair: the
                                                // message pattern of an 8-GPU node's rank, rehearsed on one GPU
    bool merged = false;                        // the own columns are few: one product on a compact x that holds own and remote entries alike
    bool allgather = false;                     // the remote x is the whole vector, slab by slab, padded to `pad` entries per rank
    int64_t pad = 0;
    std::vector<int64_t> off;                   // partition of x (= of the rows for a square operator), world+1
    int32_t local_rows = 0;
    int64_t nnz_own = 0, nnz_rem = 0;
    g4s_csr_t A_own = nullptr, A_rem = nullptr;
    int32_t n_ref = 0;
    std::vector<int64_t> recv_cut;              // nseg+1 (= world+1 outside loopback): segment [recv_cut[k], recv_cut[k+1]) of x_rem belongs to owner k
    std::vector<int64_t> give_cut;              // nseg+1: segment of the send buffer that goes to peer k
    std::vector<char> give_set;
    int32_t *d_src_own = nullptr, *d_src_rem = nullptr;   // G4S_SPMV_UPDATABLE: where in the local CSR arrays the entries of the two parts came from
    double *d_stage = nullptr;                  // … and one staging array for their new values (max(nnz_own, nnz_rem) doubles)
    int32_t *d_want = nullptr;                  // n_ref indices, local to their owner's slab (what this rank asks for), in x_rem order
    int32_t *d_give = nullptr;                  // indices into x_local, concatenated per peer
    double *d_send = nullptr, *d_xrem = nullptr;
    // own + remote form: rows that hold remote-column entries, when they are few (the two boundary planes of a stencil slab): A_rem then has only those
    // rows, its product lands in d_rem_y and is added into y by row index — instead of a read-modify-write of every row of y for a handful of entries
    int32_t n_rem_rows = 0;
    int32_t *d_rem_rows = nullptr;
    double *d_rem_y = nullptr;
    hipStream_t cstream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_done = nullptr;
    ncclComm_t comm = nullptr;
    bool exchange_posted = false;
    bool columns = false;                       // column-partition variant (g4s_spmv_dist_create_columns): this rank holds A[:, its slab of x]; the product is a
                                                // partial y of ALL rows, summed over the ranks by an all-reduce
    bool poisoned = false;                      // a set-up exchange timed out or RCCL reported an asynchronous error: the communicator has been aborted, operations on the
                                                // side stream may never complete. Nothing of this handle is synchronised or freed on the device any more (see dist_poison)
};

namespace {

void dist_release(g4s_spmv_dist_s *h)
{
    if (!h) return;
    if (h->poisoned) { delete h; return; }      // hipFree / hipStreamDestroy synchronise with work that may never finish: the device memory is left to the process's exit
    if (h->A_own) g4s_csr_destroy(h->A_own);
    if (h->A_rem) g4s_csr_destroy(h->A_rem);
    (void)hipFree(h->d_src_own); (void)hipFree(h->d_src_rem); (void)hipFree(h->d_stage);
    (void)hipFree(h->d_want); (void)hipFree(h->d_give); (void)hipFree(h->d_send); (void)hipFree(h->d_xrem); (void)hipFree(h->d_rem_rows); (void)hipFree(h->d_rem_y);
    if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->cstream) (void)hipStreamDestroy(h->cstream);
    delete h;
}

int owner_of(const std::vector<int64_t> &off, int64_t col)
{
    return (int)(std::upper_bound(off.begin(), off.end(), col) - off.begin()) - 1;
}

// One rank's rows cut into the own-column and the remote-column part (host arrays; set-up logic, runs once per matrix).
struct Split {
    int nseg = 1;
    bool merged = false, allgather = false;
    int64_t pad = 0;
    int32_t n_ref = 0;                                              // length of the remote x (packed: referenced columns; all-gather: world·pad)
    std::vector<int32_t> orp, oci, rrp, rci, want;
    std::vector<int32_t> osrc, rsrc;                                // position in the local CSR arrays of every entry of the two parts (g4s_spmv_dist_update_values)
    std::vector<double> ova, rva;
    std::vector<int64_t> recv_cut;
};

// Everything is validated before it is used as an index (a crash here would be a host crash across the C boundary): rowptr[0] == 0,
// non-decreasing, nnz inside int32, every column inside [0, n_cols).
int split_rows(int rank, int world, const std::vector<int64_t> &off, int64_t n_cols, int32_t m, const int32_t *rp, const int32_t *ci, const double *va,
               bool loopback, bool allgather, Split &S)
{
    if (rp[0] != 0) return g4s::set_error(G4S_ERR_INVALID, "rowptr[0] != 0");
    for (int32_t i = 0; i < m; ++i)
        if (rp[i + 1] < rp[i]) return g4s::set_error(G4S_ERR_INVALID, "rowptr decreases at row %d", i);
    const int64_t nnz = rp[m];
    if (nnz < 0) return g4s::set_error(G4S_ERR_INVALID, "rowptr[rows] is negative");
    if (nnz && (!ci || !va)) return g4s::set_error(G4S_ERR_INVALID, "colids/values NULL with nnz > 0");
    for (int64_t k = 0; k < nnz; ++k)
        if (ci[k] < 0 || ci[k] >= n_cols) return g4s::set_error(G4S_ERR_INVALID, "a column index is outside [0, n_cols)");
    const int64_t r0 = off[rank], r1 = off[(size_t)rank + 1];
    // own range of columns (loopback: only the first half of the slab counts as own, the rest travels rank 0 → rank 0 through RCCL).
    // Merged form: when fewer than a quarter of the entries sit in own columns (a power-law graph cut into row slabs: ≈10 %), two products
    // cost more than the overlap buys (tools/dist_probe.py: 0.12–0.16 ms against 0.07–0.09 ms for one product on an eighth of configs[1]),
    // so the own columns are renumbered into the remote x like everybody else's (packed: filled by a local gather instead of a message).
    int64_t own_lo = r0, own_hi = loopback ? r0 + (r1 - r0) / 2 : r1;
    if (!loopback && world > 1 && !getenv("G4S_DIST_NO_MERGE")) {
        int64_t in_own = 0;
        for (int64_t k = 0; k < nnz; ++k) in_own += ci[k] >= own_lo && ci[k] < own_hi;
        if (getenv("G4S_DIST_MERGE") || 4 * in_own < nnz) { S.merged = true; own_hi = own_lo; }
    }
    S.allgather = allgather;
    S.nseg = world;
    if (loopback && !allgather) {
        const char *e = getenv("G4S_DIST_LOOPBACK_PEERS");
        S.nseg = std::max(1, std::min(64, e ? atoi(e) : 7));
    }
    const int nseg = S.nseg;
    S.recv_cut.assign((size_t)nseg + 1, 0);
    std::vector<int32_t> ref;                                       // packed mode: the referenced remote columns, ascending
    if (allgather) {
        for (int k = 0; k < world; ++k) S.pad = std::max(S.pad, off[(size_t)k + 1] - off[k]);
        if (S.pad * world > INT32_MAX) return g4s::set_error(G4S_ERR_UNSUPPORTED, "all-gather exchange: world * longest slab exceeds the int32 column type");
        S.n_ref = (int32_t)(S.pad * world);
        for (int k = 0; k <= world; ++k) S.recv_cut[k] = S.pad * k;
    } else {
        for (int64_t k = 0; k < nnz; ++k)
            if (ci[k] < own_lo || ci[k] >= own_hi) ref.push_back(ci[k]);
        std::sort(ref.begin(), ref.end());
        ref.erase(std::unique(ref.begin(), ref.end()), ref.end());
        S.n_ref = (int32_t)ref.size();
        // what this rank wants from every owner: ref is sorted, so owner k's columns are one segment
        S.want.resize(ref.size());
        for (int32_t i = 0; i < S.n_ref; ++i) {
            // loopback: the remote half [own_hi, r1) in nseg equal column ranges, every one of them "owned" by rank 0
            const int k = loopback ? (int)std::min<int64_t>(nseg - 1, (ref[i] - own_hi) * nseg / std::max<int64_t>(1, r1 - own_hi)) : owner_of(off, ref[i]);
            S.recv_cut[(size_t)k + 1]++;
            S.want[i] = (int32_t)(ref[i] - off[loopback ? 0 : k]);
        }
        for (int k = 0; k < nseg; ++k) S.recv_cut[(size_t)k + 1] += S.recv_cut[k];
        if (!loopback && !S.merged && S.recv_cut[(size_t)rank + 1] != S.recv_cut[rank]) return g4s::set_error(G4S_ERR_INVALID, "internal: own columns among the remote ones");
    }
    S.orp.assign((size_t)m + 1, 0); S.rrp.assign((size_t)m + 1, 0);
    S.oci.reserve((size_t)nnz); S.ova.reserve((size_t)nnz);
    for (int32_t i = 0; i < m; ++i) {
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
            const int32_t c = ci[k];
            if (c >= own_lo && c < own_hi) { S.oci.push_back((int32_t)(c - r0)); S.ova.push_back(va[k]); S.osrc.push_back(k); }
            else {
                int32_t rc;
                if (allgather) { const int o = loopback ? 0 : owner_of(off, c); rc = (int32_t)(S.pad * o + (c - off[o])); }
                else rc = (int32_t)(std::lower_bound(ref.begin(), ref.end(), c) - ref.begin());
                S.rci.push_back(rc); S.rva.push_back(va[k]); S.rsrc.push_back(k);
            }
        }
        S.orp[(size_t)i + 1] = (int32_t)S.oci.size(); S.rrp[(size_t)i + 1] = (int32_t)S.rci.size();
    }
    return G4S_OK;
}

int check_partition(int32_t rank, int32_t world, const int64_t *row_offsets, int64_t n_cols)
{
    G4S_REQUIRE(world >= 1 && rank >= 0 && rank < world && row_offsets, "bad partition arguments");
    for (int k = 0; k < world; ++k) G4S_REQUIRE(row_offsets[k] <= row_offsets[k + 1], "row_offsets must not decrease");
    G4S_REQUIRE(row_offsets[0] == 0 && row_offsets[world] == n_cols, "square operator expected: x is partitioned like the rows (row_offsets[world] == n_cols)");
    G4S_REQUIRE(n_cols <= INT32_MAX, "n_cols exceeds the int32 index type");
    return G4S_OK;
}

template <typename T>
T *dup_array(const std::vector<T> &v)
{
    T *p = static_cast<T *>(g4s_malloc(sizeof(T) * std::max<size_t>(v.size(), 1)));
    if (p && !v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

} // namespace

// Equal-work contiguous row partition: BIN::set_rows_offset (mm/inc/BIN.h:101-122) — prefix-sum the per-row work, average share
// avg = ceil(total / parts), boundary t = lower_bound(prefix, avg·t); the last boundary is `rows`. The reference splits rows over OpenMP
// threads with work = flop per row; here the same rule splits rows over GPUs. Host-side set-up logic (no GPU needed).
G4S_API g4s_status g4s_row_partition(int32_t rows, const int32_t *rowptr, const int64_t *row_work, int64_t row_weight, int32_t parts,
                                     int64_t *row_offsets, unsigned flags)
{
    G4S_REQUIRE(rows >= 0 && parts >= 1 && row_offsets && (rowptr || row_work || rows == 0), "bad argument");
    try {
        std::vector<int64_t> prefix((size_t)rows + 1, 0);
        if (row_work) {
            for (int32_t i = 0; i < rows; ++i) { G4S_REQUIRE(row_work[i] >= 0, "negative row work"); prefix[(size_t)i + 1] = prefix[i] + row_work[i]; }
        } else if (rows) {
            std::vector<int32_t> rp((size_t)rows + 1);
            if (flags & G4S_DEVICE_POINTERS) G4S_HIP_TRY(hipMemcpy(rp.data(), rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost));
            else std::copy(rowptr, rowptr + rows + 1, rp.begin());
            for (int32_t i = 0; i < rows; ++i) {
                G4S_REQUIRE(rp[(size_t)i + 1] >= rp[i], "rowptr decreases");
                prefix[(size_t)i + 1] = prefix[i] + (rp[(size_t)i + 1] - rp[i]) + row_weight;
            }
        }
        const int64_t total = prefix[rows], avg = (total + parts - 1) / parts;
        row_offsets[0] = 0;
        for (int32_t t = 1; t <= parts; ++t)
            row_offsets[t] = std::min<int64_t>(rows, std::lower_bound(prefix.begin(), prefix.end(), avg * t) - prefix.begin());
        row_offsets[parts] = rows;                                  // BIN.h:120
        for (int32_t t = 1; t <= parts; ++t) row_offsets[t] = std::max(row_offsets[t], row_offsets[t - 1]);
    } catch (const std::bad_alloc &) {
        return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    }
    return G4S_OK;
}

G4S_API g4s_status g4s_dist_split_rows(int32_t rank, int32_t world, const int64_t *row_offsets, int64_t n_cols,
                                       const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags, g4s_dist_split *out)
{
    G4S_REQUIRE(out && rowptr, "NULL argument");
    std::memset(out, 0, sizeof(*out));
    G4S_REQUIRE(!(flags & G4S_DEVICE_POINTERS), "g4s_dist_split_rows works on host arrays");
    G4S_TRY(check_partition(rank, world, row_offsets, n_cols));
    const bool loopback = (flags & G4S_DIST_LOOPBACK) != 0;
    G4S_REQUIRE(!loopback || world == 1, "G4S_DIST_LOOPBACK is a single-rank rehearsal mode");
    try {
        const std::vector<int64_t> off(row_offsets, row_offsets + world + 1);
        const int32_t m = (int32_t)(off[(size_t)rank + 1] - off[rank]);
        Split S;
        G4S_TRY(split_rows(rank, world, off, n_cols, m, rowptr, colids, values, loopback, (flags & G4S_DIST_ALLGATHER) != 0, S));
        out->local_rows = m; out->n_ref = S.n_ref; out->merged = S.merged; out->allgather = S.allgather; out->pad = S.pad;
        out->nnz_own = (int64_t)S.oci.size(); out->nnz_rem = (int64_t)S.rci.size();
        out->own_rowptr = dup_array(S.orp); out->own_colids = dup_array(S.oci); out->own_values = dup_array(S.ova);
        out->rem_rowptr = dup_array(S.rrp); out->rem_colids = dup_array(S.rci); out->rem_values = dup_array(S.rva);
        out->want = dup_array(S.want); out->recv_cut = dup_array(S.recv_cut);   // (nseg + 1 entries: world + 1 outside the loopback rehearsal)
        if (!out->own_rowptr || !out->own_colids || !out->own_values || !out->rem_rowptr || !out->rem_colids || !out->rem_values || !out->want || !out->recv_cut) {
            g4s_dist_split_free(out);
            return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
        }
    } catch (const std::bad_alloc &) {
        g4s_dist_split_free(out);
        return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    }
    return G4S_OK;
}

G4S_API void g4s_dist_split_free(g4s_dist_split *s)
{
    if (!s) return;
    g4s_free(s->own_rowptr); g4s_free(s->own_colids); g4s_free(s->own_values);
    g4s_free(s->rem_rowptr); g4s_free(s->rem_colids); g4s_free(s->rem_values);
    g4s_free(s->want); g4s_free(s->recv_cut);
    std::memset(s, 0, sizeof(*s));
}

// Rows partitioned by row_offsets, x (the columns) by col_offsets: the square operator has both alike; a rectangular one (the discrete
// divergence / gradient of the Stokes iteration: elements × equations and back) has two different partitions.
G4S_API g4s_status g4s_spmv_dist_create_rect(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *row_offsets, const int64_t *col_offsets,
                                             const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_REQUIRE(rowptr && row_offsets && col_offsets && world >= 1, "NULL argument");
    const int64_t n_cols = col_offsets[world];
    G4S_TRY(check_partition(rank, world, col_offsets, n_cols));
    for (int k = 0; k < world; ++k) G4S_REQUIRE(row_offsets[k] <= row_offsets[k + 1], "row_offsets must not decrease");
    G4S_REQUIRE(row_offsets[0] == 0 && row_offsets[world] <= INT32_MAX, "bad row_offsets");
    const int64_t r0 = col_offsets[rank], r1 = col_offsets[rank + 1];      // this rank's slab of x
    const int32_t m = (int32_t)(row_offsets[rank + 1] - row_offsets[rank]);
    auto h = new (std::nothrow) g4s_spmv_dist_s();
    if (!h) return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    auto fail = [&](int code) { dist_release(h); return code; };
    try {
        h->rank = rank; h->world = world; h->local_rows = m;
        h->off.assign(col_offsets, col_offsets + world + 1);        // the partition of x: what the exchange is about
        h->loopback = (flags & G4S_DIST_LOOPBACK) != 0;
        if (h->loopback && world != 1) return fail(g4s::set_error(G4S_ERR_INVALID, "G4S_DIST_LOOPBACK is a single-rank rehearsal mode"));
        const bool allgather = (flags & G4S_DIST_ALLGATHER) != 0 || (getenv("G4S_DIST_EXCHANGE") && !strcmp(getenv("G4S_DIST_EXCHANGE"), "allgather"));

        // ---- the local rows on the host (set-up runs once per matrix)
        std::vector<int32_t> rp((size_t)m + 1);
        const bool dp = (flags & G4S_DEVICE_POINTERS) != 0;
        if (dp) { if (hipMemcpy(rp.data(), rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(g4s::set_error(G4S_ERR_HIP, "D2H copy of rowptr failed")); }
        else std::copy(rowptr, rowptr + m + 1, rp.begin());
        const int64_t nnz = rp[m];
        if (nnz < 0) return fail(g4s::set_error(G4S_ERR_INVALID, "rowptr[rows] is negative"));
        if (nnz && !(colids && values)) return fail(g4s::set_error(G4S_ERR_INVALID, "colids/values NULL with nnz > 0"));
        std::vector<int32_t> ci((size_t)nnz);
        std::vector<double> va((size_t)nnz);
        if (nnz) {
            if (dp) {
                if (hipMemcpy(ci.data(), colids, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(va.data(), values, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail(g4s::set_error(G4S_ERR_HIP, "D2H copy of the matrix failed"));
            } else { std::copy(colids, colids + nnz, ci.begin()); std::copy(values, values + nnz, va.begin()); }
        }
        Split S;
        int st = split_rows(rank, world, h->off, n_cols, m, rp.data(), ci.data(), va.data(), h->loopback, allgather, S);
        if (st != G4S_OK) return fail(st);
        h->merged = S.merged; h->allgather = S.allgather; h->pad = S.pad; h->n_ref = S.n_ref; h->nseg = S.nseg;
        h->nnz_own = (int64_t)S.oci.size(); h->nnz_rem = (int64_t)S.rci.size();
        h->recv_cut = S.recv_cut;
        const unsigned path_flags = flags & (G4S_SPMV_NO_NT | G4S_SPMV_BLOCKED | G4S_SPMV_STREAM | G4S_SPMV_UPDATABLE);
        if (flags & G4S_SPMV_UPDATABLE) {
            const size_t stage = std::max<size_t>(std::max(S.osrc.size(), S.rsrc.size()), 1);
            if (g4s::device_malloc((void **)&h->d_src_own, sizeof(int32_t) * std::max<size_t>(S.osrc.size(), 1)) != hipSuccess ||
                g4s::device_malloc((void **)&h->d_src_rem, sizeof(int32_t) * std::max<size_t>(S.rsrc.size(), 1)) != hipSuccess ||
                g4s::device_malloc((void **)&h->d_stage, sizeof(double) * stage) != hipSuccess)
                return fail(g4s::set_error(G4S_ERR_NOMEM, "device allocation failed"));
            if ((!S.osrc.empty() && hipMemcpy(h->d_src_own, S.osrc.data(), sizeof(int32_t) * S.osrc.size(), hipMemcpyHostToDevice) != hipSuccess) ||
                (!S.rsrc.empty() && hipMemcpy(h->d_src_rem, S.rsrc.data(), sizeof(int32_t) * S.rsrc.size(), hipMemcpyHostToDevice) != hipSuccess))
                return fail(g4s::set_error(G4S_ERR_HIP, "H2D copy failed"));
        }
        st = g4s_csr_create(&h->A_own, m, (int32_t)(r1 - r0), h->nnz_own, S.orp.data(), S.oci.data(), S.ova.data(), G4S_HOST_POINTERS | path_flags);
        if (st != G4S_OK) return fail(st);
        // remote-column part: only the rows that have one, when they are under a quarter of the slab (never in the merged form, whose one product writes all of y)
        std::vector<int32_t> rem_rows;
        if (!h->merged && h->nnz_rem > 0) {
            for (int32_t i = 0; i < m; ++i)
                if (S.rrp[(size_t)i + 1] > S.rrp[i]) rem_rows.push_back(i);
            if ((int64_t)rem_rows.size() * 4 >= m) rem_rows.clear();
        }
        if (!rem_rows.empty()) {
            std::vector<int32_t> crp(rem_rows.size() + 1, 0);
            for (size_t i = 0; i < rem_rows.size(); ++i) crp[i + 1] = S.rrp[(size_t)rem_rows[i] + 1];   // the entries stay where they are: empty rows hold none
            h->n_rem_rows = (int32_t)rem_rows.size();
            st = g4s_csr_create(&h->A_rem, h->n_rem_rows, std::max(h->n_ref, 1), h->nnz_rem, crp.data(), S.rci.data(), S.rva.data(), G4S_HOST_POINTERS | path_flags);
            if (st != G4S_OK) return fail(st);
            if (g4s::device_malloc((void **)&h->d_rem_rows, sizeof(int32_t) * rem_rows.size()) != hipSuccess ||
                g4s::device_malloc((void **)&h->d_rem_y, sizeof(double) * rem_rows.size()) != hipSuccess)
                return fail(g4s::set_error(G4S_ERR_NOMEM, "device allocation failed"));
            if (hipMemcpy(h->d_rem_rows, rem_rows.data(), sizeof(int32_t) * rem_rows.size(), hipMemcpyHostToDevice) != hipSuccess)
                return fail(g4s::set_error(G4S_ERR_HIP, "H2D copy failed"));
        } else {
            st = g4s_csr_create(&h->A_rem, m, std::max(h->n_ref, 1), h->nnz_rem, S.rrp.data(), S.rci.data(), S.rva.data(), G4S_HOST_POINTERS | path_flags);
            if (st != G4S_OK) return fail(st);
        }
        h->give_cut.assign((size_t)h->nseg + 1, 0);
        h->give_set.assign((size_t)h->nseg, 0);
        if (g4s::device_malloc((void **)&h->d_want, sizeof(int32_t) * std::max<size_t>(S.want.size(), 1)) != hipSuccess ||
            g4s::device_malloc((void **)&h->d_xrem, sizeof(double) * (size_t)std::max(h->n_ref, 1)) != hipSuccess)
            return fail(g4s::set_error(G4S_ERR_NOMEM, "device allocation failed"));
        if (!S.want.empty() && hipMemcpy(h->d_want, S.want.data(), sizeof(int32_t) * S.want.size(), hipMemcpyHostToDevice) != hipSuccess)
            return fail(g4s::set_error(G4S_ERR_HIP, "H2D copy failed"));
        if (hipMemset(h->d_xrem, 0, sizeof(double) * (size_t)std::max(h->n_ref, 1)) != hipSuccess) return fail(g4s::set_error(G4S_ERR_HIP, "memset failed"));
        if (hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess)
            return fail(g4s::set_error(G4S_ERR_HIP, "stream / event creation failed"));
        if (h->allgather) {                                        // nothing to wire: every rank sends its slab, every rank knows where each slab lands
            std::fill(h->give_set.begin(), h->give_set.end(), 1);
            h->give_cut[(size_t)h->nseg] = h->pad;                   // the one send segment: this rank's slot of the gathered vector, for every peer alike
        } else if (world == 1 && !h->loopback) h->give_set[0] = 1;  // nothing to exchange
    } catch (const std::bad_alloc &) {
        return fail(g4s::set_error(G4S_ERR_NOMEM, "host allocation failed"));
    }
    *out = h;
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_dist_create(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *row_offsets, int64_t n_cols,
                                        const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_TRY(check_partition(rank, world, row_offsets, n_cols));    // square: x is partitioned like the rows
    return g4s_spmv_dist_create_rect(out, rank, world, row_offsets, row_offsets, rowptr, colids, values, flags);
}

// Column-partition variant (north_star: "all-reduce of partial products … only where the problem shards naturally"; SURVEY §8e: "implement only as a correctness
// variant"): rank g holds the COLUMNS [col_offsets[g], col_offsets[g+1]) of A — every row, global column ids inside its slab — and the matching slab of x;
// y = Σ_g A[:, g]·x_g: every rank forms a partial y of all n_rows entries and an all-reduce (ncclAllReduce on the handle's communicator, or the caller's own
// between _begin and _finish) sums them, so every rank ends with the WHOLE y. For a square n × n operator that moves 2·(N−1)/N·n entries per rank and product
// against the (N−1)/N·n of the all-gather of x in the row partition — and n instead of the few per cent of n the packed halo exchange moves —, and every rank
// then holds (and has reduced) all of y: the row partition is the product's form, this one exists to be checked against it (closest reference pattern: the
// rank-strided loop + MPI_Allreduce of cantera/src/thermo/RedlichKwongMFTP.cpp:1014-1015).
G4S_API g4s_status g4s_spmv_dist_create_columns(g4s_spmv_dist_t *out, int32_t rank, int32_t world, const int64_t *col_offsets, int32_t n_rows,
                                                const int32_t *rowptr, const int32_t *colids, const double *values, unsigned flags)
{
    G4S_REQUIRE(out, "out is NULL");
    *out = nullptr;
    G4S_REQUIRE(rowptr && col_offsets && world >= 1 && rank >= 0 && rank < world && n_rows >= 0, "bad argument");
    for (int k = 0; k < world; ++k) G4S_REQUIRE(col_offsets[k] <= col_offsets[k + 1], "col_offsets must not decrease");
    G4S_REQUIRE(col_offsets[0] == 0 && col_offsets[world] <= INT32_MAX, "bad col_offsets");
    const int64_t c0 = col_offsets[rank], c1 = col_offsets[rank + 1];
    auto h = new (std::nothrow) g4s_spmv_dist_s();
    if (!h) return g4s::set_error(G4S_ERR_NOMEM, "host allocation failed");
    auto fail = [&](int code) { dist_release(h); return code; };
    try {
        h->rank = rank; h->world = world; h->local_rows = n_rows; h->columns = true; h->nseg = world;
        h->off.assign(col_offsets, col_offsets + world + 1);
        const bool dp = (flags & G4S_DEVICE_POINTERS) != 0;
        std::vector<int32_t> rp((size_t)n_rows + 1);
        if (dp) { if (hipMemcpy(rp.data(), rowptr, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(g4s::set_error(G4S_ERR_HIP, "D2H copy of rowptr failed")); }
        else std::copy(rowptr, rowptr + n_rows + 1, rp.begin());
        if (rp[0] != 0) return fail(g4s::set_error(G4S_ERR_INVALID, "rowptr[0] != 0"));
        for (int32_t i = 0; i < n_rows; ++i)
            if (rp[(size_t)i + 1] < rp[i]) return fail(g4s::set_error(G4S_ERR_INVALID, "rowptr decreases at row %d", i));
        const int64_t nnz = rp[n_rows];
        if (nnz && !(colids && values)) return fail(g4s::set_error(G4S_ERR_INVALID, "colids/values NULL with nnz > 0"));
        std::vector<int32_t> ci((size_t)nnz);
        std::vector<double> va((size_t)nnz);
        if (nnz) {
            if (dp) {
                if (hipMemcpy(ci.data(), colids, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(va.data(), values, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail(g4s::set_error(G4S_ERR_HIP, "D2H copy of the matrix failed"));
            } else { std::copy(colids, colids + nnz, ci.begin()); std::copy(values, values + nnz, va.begin()); }
        }
        for (int64_t k = 0; k < nnz; ++k) {
            if (ci[(size_t)k] < c0 || ci[(size_t)k] >= c1) return fail(g4s::set_error(G4S_ERR_INVALID, "a column index is outside this rank's slab [%lld, %lld)", (long long)c0, (long long)c1));
            ci[(size_t)k] -= (int32_t)c0;
        }
        h->nnz_own = nnz;
        const unsigned path_flags = flags & (G4S_SPMV_NO_NT | G4S_SPMV_BLOCKED | G4S_SPMV_STREAM | G4S_SPMV_UPDATABLE);
        int st = g4s_csr_create(&h->A_own, n_rows, (int32_t)std::max<int64_t>(c1 - c0, 1), nnz, rp.data(), ci.data(), va.data(), G4S_HOST_POINTERS | path_flags);
        if (st != G4S_OK) return fail(st);
        h->recv_cut.assign((size_t)world + 1, 0);
        h->give_cut.assign((size_t)world + 1, 0);
        h->give_set.assign((size_t)world, 1);                       // nothing to wire
        if (hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess)
            return fail(g4s::set_error(G4S_ERR_HIP, "stream / event creation failed"));
    } catch (const std::bad_alloc &) {
        return fail(g4s::set_error(G4S_ERR_NOMEM, "host allocation failed"));
    }
    *out = h;
    return G4S_OK;
}

// New values for this rank's rows (the order of the arrays the handle was created from): each of the two parts gathers its entries through its source map
// and hands them to its own CSR handle (g4s_csr_update_values). No communication: every rank updates its slab.
G4S_API g4s_status g4s_spmv_dist_update_values(g4s_spmv_dist_t h, const double *values_local, unsigned flags, void *stream)
{
    G4S_REQUIRE(h && !h->poisoned, "bad handle");
    if (h->columns) return g4s_csr_update_values(h->A_own, values_local, flags, stream);   // one part, in the caller's order
    G4S_REQUIRE(h->d_src_own, "the handle was created without G4S_SPMV_UPDATABLE");
    G4S_REQUIRE(values_local || h->nnz_own + h->nnz_rem == 0, "values is NULL");
    G4S_REQUIRE(flags & G4S_DEVICE_POINTERS, "g4s_spmv_dist_update_values takes a device array");
    hipStream_t s = g4s::as_stream(stream);
    for (int part = 0; part < 2; ++part) {
        const int64_t n = part ? h->nnz_rem : h->nnz_own;
        if (!n) continue;
        const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(dist_pack_kernel, dim3(grid), dim3(256), 0, s, (long long)n, part ? h->d_src_rem : h->d_src_own, values_local, h->d_stage);
        G4S_HIP_TRY(hipGetLastError());
        G4S_TRY(g4s_csr_update_values(part ? h->A_rem : h->A_own, h->d_stage, G4S_DEVICE_POINTERS, stream));   // (the parts own their arrays: copied in, in stream order)
    }
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_dist_destroy(g4s_spmv_dist_t h)
{
    if (h && !h->poisoned) (void)hipDeviceSynchronize();
    dist_release(h);
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_dist_get_info(g4s_spmv_dist_t h, g4s_spmv_dist_info *info)
{
    G4S_REQUIRE(h && info, "NULL argument");
    info->rank = h->rank; info->world = h->world; info->local_rows = h->local_rows; info->n_ref = h->n_ref;
    info->nnz_own = h->nnz_own; info->nnz_rem = h->nnz_rem;
    if (h->allgather) {                                             // every slab, padded, from every other rank; this rank's slot to every other rank
        info->recv_bytes = 8 * h->pad * (h->world - 1);
        info->send_bytes = 8 * h->pad * (h->world - 1);
    } else {
        info->recv_bytes = 8 * (h->recv_cut[h->nseg] - (h->merged ? h->recv_cut[(size_t)h->rank + 1] - h->recv_cut[h->rank] : 0));
        info->send_bytes = 8 * h->give_cut[h->nseg];
    }
    info->reserved = (h->merged ? 1 : 0) | (h->allgather ? 2 : 0) | (h->poisoned ? 4 : 0) | (h->columns ? 8 : 0);
    if (h->columns && !h->poisoned) {                              // every rank sends and receives its partial y in the all-reduce (ring: 2·(N−1)/N of it)
        info->send_bytes = info->recv_bytes = h->world > 1 ? (int64_t)(16.0 * h->local_rows * (h->world - 1) / h->world) : 0;
        g4s_csr_info ci2;
        G4S_TRY(g4s_csr_get_info(h->A_own, &ci2));
        info->own_path = ci2.spmv_path; info->rem_path = 0; info->connected = 1;
        return G4S_OK;
    }
    g4s_csr_info ci;
    if (h->poisoned) { info->own_path = info->rem_path = 0; info->connected = 0; return G4S_OK; }
    G4S_TRY(g4s_csr_get_info(h->A_own, &ci)); info->own_path = ci.spmv_path;
    G4S_TRY(g4s_csr_get_info(h->A_rem, &ci)); info->rem_path = ci.spmv_path;
    int ready = 1;
    for (int k = 0; k < h->nseg; ++k) ready &= h->give_set[k] || (k == h->rank && !h->loopback);
    info->connected = ready;
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_dist_want(g4s_spmv_dist_t h, int32_t peer, int64_t *count, const int32_t **idx_dev)
{
    G4S_REQUIRE(h && peer >= 0 && peer < h->nseg && count, "bad argument");
    G4S_REQUIRE(!h->poisoned, "the handle was poisoned by a failed set-up exchange: destroy it and exit the process");
    *count = h->recv_cut[(size_t)peer + 1] - h->recv_cut[peer];
    if (idx_dev) *idx_dev = h->d_want + h->recv_cut[peer];
    return G4S_OK;
}

// The give lists arrive peer by peer, in any order; the send buffer is laid out in peer order once all of them are known.
G4S_API g4s_status g4s_spmv_dist_set_give(g4s_spmv_dist_t h, int32_t peer, int64_t count, const int32_t *idx, unsigned flags)
{
    G4S_REQUIRE(h && peer >= 0 && peer < h->nseg && count >= 0 && (idx || count == 0), "bad argument");
    G4S_REQUIRE(!h->poisoned, "the handle was poisoned by a failed set-up exchange: destroy it and exit the process");
    G4S_REQUIRE(!h->allgather, "the all-gather exchange has no give lists");
    G4S_REQUIRE(!h->d_send, "the give lists are final once a product has run");
    std::vector<int32_t> list((size_t)count);
    if (count) {
        if (flags & G4S_DEVICE_POINTERS) G4S_HIP_TRY(hipMemcpy(list.data(), idx, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost));
        else std::copy(idx, idx + count, list.begin());
        const int32_t slab = (int32_t)(h->off[(size_t)h->rank + 1] - h->off[h->rank]);
        for (int32_t v : list) G4S_REQUIRE(v >= 0 && v < slab, "a requested index is outside this rank's slab");
    }
    // append to the device list: rebuild it in peer order from what is known so far
    std::vector<int32_t> all((size_t)h->give_cut[h->nseg]);
    if (!all.empty()) G4S_HIP_TRY(hipMemcpy(all.data(), h->d_give, sizeof(int32_t) * all.size(), hipMemcpyDeviceToHost));
    std::vector<int32_t> next;
    std::vector<int64_t> cut((size_t)h->nseg + 1, 0);
    for (int k = 0; k < h->nseg; ++k) {
        cut[k] = (int64_t)next.size();
        if (k == peer) next.insert(next.end(), list.begin(), list.end());
        else next.insert(next.end(), all.begin() + h->give_cut[k], all.begin() + h->give_cut[(size_t)k + 1]);
    }
    cut[h->nseg] = (int64_t)next.size();
    (void)hipFree(h->d_give);
    h->d_give = nullptr;
    if (g4s::device_malloc((void **)&h->d_give, sizeof(int32_t) * std::max<size_t>(next.size(), 1)) != hipSuccess) return g4s::set_error(G4S_ERR_NOMEM, "device allocation failed");
    if (!next.empty()) G4S_HIP_TRY(hipMemcpy(h->d_give, next.data(), sizeof(int32_t) * next.size(), hipMemcpyHostToDevice));
    h->give_cut = cut;
    h->give_set[peer] = 1;
    return G4S_OK;
}

namespace {

// Waits for `stream` with a deadline instead of an unbounded hipStreamSynchronize: a peer that failed before entering the same
// collective would otherwise leave this rank blocked for ever (G4S_DIST_TIMEOUT_S, default 120). Also surfaces asynchronous RCCL errors.
int wait_stream(hipStream_t stream, ncclComm_t comm, const char *what)
{
    const char *e = getenv("G4S_DIST_TIMEOUT_S");
    const double limit = e ? atof(e) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return G4S_OK;
        if (q != hipErrorNotReady) return g4s::set_error(G4S_ERR_HIP, "%s: %s", what, hipGetErrorString(q));
        if (comm && g_rccl.CommGetAsyncError) {
            ncclResult_t ar = ncclSuccess;
            if (g_rccl.CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                return g4s::set_error(G4S_ERR_HIP, "%s: RCCL reports %s", what, g_rccl.GetErrorString(ar));
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            return g4s::set_error(G4S_ERR_HIP, "%s: no completion within %.3g s (a peer that never entered the exchange?); the communicator has been aborted, the handle is "
                                               "unusable: report and exit the process", what, limit);
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

// One grouped point-to-point round. ncclGroupEnd is ALWAYS called once ncclGroupStart has succeeded — returning from inside an open group
// would leave the thread's group open and every later RCCL call of the process (the host framework's too) queued behind it.
template <typename Body>
int rccl_group(Body body, const char *what)
{
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return g4s::set_error(G4S_ERR_HIP, "%s: ncclGroupStart failed: %s", what, g_rccl.GetErrorString(r));
    const ncclResult_t rb = body();
    const ncclResult_t re = g_rccl.GroupEnd();
    if (rb != ncclSuccess) return g4s::set_error(G4S_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(rb));
    if (re != ncclSuccess) return g4s::set_error(G4S_ERR_HIP, "%s: ncclGroupEnd failed: %s", what, g_rccl.GetErrorString(re));
    return G4S_OK;
}

struct DevFree {
    void *p = nullptr;
    ~DevFree() { if (p) (void)hipFree(p); }
};

// A set-up exchange did not complete (deadline, or an asynchronous RCCL error): the grouped send / receive is still queued on the side stream and hipFree,
// hipStreamDestroy and hipDeviceSynchronize all wait for it — the time-out would only move the hang. So: abort the communicator (ncclCommAbort makes RCCL's
// kernels leave), mark the handle, and from here on free and synchronise nothing that belongs to it (g4s_spmv_dist_destroy then only drops the host side).
// The communicator is gone after this; the process is expected to report the error and EXIT — a supervisor starts a fresh process (never a re-exec of one
// that has touched the GPU).
// The communicator belongs to the CALLER (g4s_comm_create → g4s_spmv_dist_connect_rccl → g4s_comm_destroy): once aborted here it is remembered, so that the
// caller's g4s_comm_destroy — and any further library call on it — finds it gone instead of destroying or using it a second time.
std::mutex g_aborted_mu;
std::vector<void *> g_aborted_comms;
bool comm_was_aborted(void *comm)
{
    std::lock_guard<std::mutex> lk(g_aborted_mu);
    return std::find(g_aborted_comms.begin(), g_aborted_comms.end(), comm) != g_aborted_comms.end();
}
int dist_poison(g4s_spmv_dist_s *h, DevFree &in_flight, int st)
{
    h->poisoned = true;
    in_flight.p = nullptr;                                         // deliberately leaked: the stuck operation may still read or write it
    if (h->comm && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort(h->comm);
        std::lock_guard<std::mutex> lk(g_aborted_mu);
        g_aborted_comms.push_back(h->comm);
    }
    h->comm = nullptr;
    return st;
}
#define G4S_REQUIRE_LIVE(h) G4S_REQUIRE(!(h)->poisoned, "the handle was poisoned by a failed set-up exchange (g4s_spmv_dist_connect_rccl aborted its communicator): destroy it and exit the process")

} // namespace

G4S_API g4s_status g4s_spmv_dist_connect_rccl(g4s_spmv_dist_t h, void *comm)
{
    G4S_REQUIRE(h && comm, "NULL argument");
    G4S_REQUIRE_LIVE(h);
    G4S_REQUIRE(!comm_was_aborted(comm), "this communicator was aborted by a failed set-up exchange: it cannot be used again");
    G4S_TRY(rccl_load());
    h->comm = reinterpret_cast<ncclComm_t>(comm);
    if (h->allgather || h->columns) return G4S_OK;                 // no lists to exchange: the collective itself is the wiring
    // a send buffer sized from earlier give lists (g4s_spmv_dist_buffers, or a product that ran before the wiring) would be too small for the new ones
    if (h->d_send) { (void)hipFree(h->d_send); h->d_send = nullptr; }
    const int W = h->nseg;                                         // segments; segment k talks to rank k (loopback: every segment to rank 0)
    auto peer = [&](int k) { return h->loopback ? 0 : k; };
    // 1. how many entries does every peer want from me? one 8-byte exchange per pair
    std::vector<long long> want_n((size_t)W), give_n((size_t)W, 0);
    for (int k = 0; k < W; ++k) want_n[k] = h->recv_cut[(size_t)k + 1] - h->recv_cut[k];
    DevFree cnt;
    G4S_HIP_TRY(g4s::device_malloc(&cnt.p, sizeof(long long) * 2 * (size_t)W));
    long long *d_cnt = static_cast<long long *>(cnt.p);
    G4S_HIP_TRY(hipMemcpy(d_cnt, want_n.data(), sizeof(long long) * (size_t)W, hipMemcpyHostToDevice));
    if (h->loopback && getenv("G4S_DIST_TEST_STALL")) {
        // test hook (tests/test_dist_capi_gpu.py): the side stream is held up for a BOUNDED time (seconds; the kernel leaves on its own) in front of the exchange —
        // what a peer that enters the exchange late, or never, looks like from here
        const double sec = std::min(10.0, std::max(0.1, atof(getenv("G4S_DIST_TEST_STALL"))));
        hipLaunchKernelGGL(dist_stall_kernel, dim3(1), dim3(1), 0, h->cstream, (long long)(sec * 1e8));
    }
    G4S_TRY(rccl_group([&]() {
        for (int k = 0; k < W; ++k) {
            if (k == h->rank && !h->loopback) continue;
            ncclResult_t r = g_rccl.Send(d_cnt + k, 1, ncclInt64, peer(k), h->comm, h->cstream);
            if (r == ncclSuccess) r = g_rccl.Recv(d_cnt + W + k, 1, ncclInt64, peer(k), h->comm, h->cstream);
            if (r != ncclSuccess) return r;
        }
        return ncclSuccess;
    }, "exchange of the want counts"));
    if (int st = wait_stream(h->cstream, h->comm, "exchange of the want counts"); st != G4S_OK) return dist_poison(h, cnt, st);
    G4S_HIP_TRY(hipMemcpy(give_n.data(), d_cnt + W, sizeof(long long) * (size_t)W, hipMemcpyDeviceToHost));
    if (!h->loopback) give_n[h->rank] = 0;
    const int64_t slab = h->off[(size_t)h->rank + 1] - h->off[h->rank];
    for (int k = 0; k < W; ++k) G4S_REQUIRE(give_n[k] >= 0 && give_n[k] <= slab, "a peer asks for more entries than this slab holds");
    // 2. the index lists themselves, received straight into the give list in peer order
    std::vector<int64_t> cut((size_t)W + 1, 0);
    for (int k = 0; k < W; ++k) cut[(size_t)k + 1] = cut[k] + give_n[k];
    (void)hipFree(h->d_give);
    h->d_give = nullptr;
    G4S_HIP_TRY(g4s::device_malloc((void **)&h->d_give, sizeof(int32_t) * (size_t)std::max<int64_t>(cut[W], 1)));
    G4S_TRY(rccl_group([&]() {
        for (int k = 0; k < W; ++k) {
            if (k == h->rank && !h->loopback) continue;
            ncclResult_t r = ncclSuccess;
            if (want_n[k]) r = g_rccl.Send(h->d_want + h->recv_cut[k], (size_t)want_n[k], ncclInt32, peer(k), h->comm, h->cstream);
            if (r == ncclSuccess && give_n[k]) r = g_rccl.Recv(h->d_give + cut[k], (size_t)give_n[k], ncclInt32, peer(k), h->comm, h->cstream);
            if (r != ncclSuccess) return r;
        }
        return ncclSuccess;
    }, "exchange of the want lists"));
    if (int st = wait_stream(h->cstream, h->comm, "exchange of the want lists"); st != G4S_OK) return dist_poison(h, cnt, st);   // (d_give, d_want stay with the poisoned handle)
    h->give_cut = cut;
    std::fill(h->give_set.begin(), h->give_set.end(), 1);
    return G4S_OK;
}

G4S_API g4s_status g4s_spmv_dist_buffers(g4s_spmv_dist_t h, double **send_dev, const int64_t **send_cut, double **recv_dev, const int64_t **recv_cut)
{
    G4S_REQUIRE(h, "NULL handle");
    G4S_REQUIRE_LIVE(h);
    if (h->columns) {                                              // nothing of x travels: the caller's transport all-reduces y itself between _begin and _finish
        if (send_dev) *send_dev = nullptr;
        if (recv_dev) *recv_dev = nullptr;
        if (send_cut) *send_cut = h->give_cut.data();
        if (recv_cut) *recv_cut = h->recv_cut.data();
        return G4S_OK;
    }
    if (h->allgather) {
        // send = this rank's slot of the gathered vector (pad entries, the same for every peer: send_cut = {0, …, 0, pad});
        // recv = the gathered vector, slot k = [k·pad, (k+1)·pad) from rank k
        if (send_dev) *send_dev = h->d_xrem + h->pad * h->rank;
        if (send_cut) *send_cut = h->give_cut.data();
        if (recv_dev) *recv_dev = h->d_xrem;
        if (recv_cut) *recv_cut = h->recv_cut.data();
        return G4S_OK;
    }
    if (!h->d_send) G4S_HIP_TRY(g4s::device_malloc((void **)&h->d_send, sizeof(double) * (size_t)std::max<int64_t>(h->give_cut[h->nseg], 1)));
    if (send_dev) *send_dev = h->d_send;
    if (send_cut) *send_cut = h->give_cut.data();
    if (recv_dev) *recv_dev = h->d_xrem;
    if (recv_cut) *recv_cut = h->recv_cut.data();
    return G4S_OK;
}

// Pack the send segments, start the exchange (RCCL mode) on the side stream, run the own-column product on the caller's stream.
G4S_API g4s_status g4s_spmv_dist_begin(g4s_spmv_dist_t h, const double *x_local_dev, double *y_local_dev, void *stream)
{
    G4S_REQUIRE(h && (y_local_dev || h->local_rows == 0) && (x_local_dev || h->off[(size_t)h->rank + 1] == h->off[h->rank]), "NULL argument");   // (a rectangular operator may own rows but no x entries, or the reverse)
    for (int k = 0; k < h->nseg; ++k)
        if (!(h->give_set[k] || (k == h->rank && !h->loopback)))
            return g4s::set_error(G4S_ERR_INVALID, "g4s_spmv_dist_begin: the give list of peer %d is not set (g4s_spmv_dist_connect_rccl or g4s_spmv_dist_set_give)", k);
    G4S_REQUIRE(!h->poisoned, "the handle was poisoned by a failed set-up exchange (g4s_spmv_dist_connect_rccl): destroy it and exit the process");
    hipStream_t s = g4s::as_stream(stream);
    h->exchange_posted = false;
    if (h->columns) {                                              // the partial y of all rows; its sum over the ranks follows (_finish, or the caller's transport)
        if (h->local_rows == 0) return G4S_OK;
        return g4s_spmv(h->A_own, x_local_dev, y_local_dev, 1.0, 0.0, stream);
    }
    if (h->allgather) {
        // own slab into its slot of the gathered vector, then ONE in-place all-gather on the side stream while the own-column product runs
        const int64_t slab = h->off[(size_t)h->rank + 1] - h->off[h->rank];
        if (slab) G4S_HIP_TRY(hipMemcpyAsync(h->d_xrem + h->pad * h->rank, x_local_dev, sizeof(double) * (size_t)slab, hipMemcpyDeviceToDevice, s));
        if (h->comm && (h->world > 1 || h->loopback)) {
            G4S_HIP_TRY(hipEventRecord(h->ev_packed, s));
            G4S_HIP_TRY(hipStreamWaitEvent(h->cstream, h->ev_packed, 0));
            G4S_RCCL_TRY(g_rccl.AllGather(h->d_xrem + h->pad * h->rank, h->d_xrem, (size_t)h->pad, ncclDouble, h->comm, h->cstream));
            G4S_HIP_TRY(hipEventRecord(h->ev_done, h->cstream));
            h->exchange_posted = true;
        }
        if (h->merged || h->local_rows == 0) return G4S_OK;        // (a rank without rows has posted its part of the exchange and is done: y may be NULL)
        return g4s_spmv(h->A_own, x_local_dev, y_local_dev, 1.0, 0.0, stream);
    }
    G4S_TRY(g4s_spmv_dist_buffers(h, nullptr, nullptr, nullptr, nullptr));
    const int64_t n_send = h->give_cut[h->nseg];
    if (n_send) {
        const int grid = (int)std::min<int64_t>((n_send + 255) / 256, 4096);
        hipLaunchKernelGGL(dist_pack_kernel, dim3(grid), dim3(256), 0, s, n_send, h->d_give, x_local_dev, h->d_send);
        G4S_HIP_TRY(hipGetLastError());
    }
    if (h->comm && (n_send || h->n_ref)) {
        G4S_HIP_TRY(hipEventRecord(h->ev_packed, s));
        G4S_HIP_TRY(hipStreamWaitEvent(h->cstream, h->ev_packed, 0));
        G4S_TRY(rccl_group([&]() {
            for (int k = 0; k < h->nseg; ++k) {
                if (k == h->rank && !h->loopback) continue;
                const int pk = h->loopback ? 0 : k;
                const int64_t ns = h->give_cut[(size_t)k + 1] - h->give_cut[k], nr = h->recv_cut[(size_t)k + 1] - h->recv_cut[k];
                ncclResult_t r = ncclSuccess;
                if (ns) r = g_rccl.Send(h->d_send + h->give_cut[k], (size_t)ns, ncclDouble, pk, h->comm, h->cstream);
                if (r == ncclSuccess && nr) r = g_rccl.Recv(h->d_xrem + h->recv_cut[k], (size_t)nr, ncclDouble, pk, h->comm, h->cstream);
                if (r != ncclSuccess) return r;
            }
            return ncclSuccess;
        }, "exchange of the x entries"));
        G4S_HIP_TRY(hipEventRecord(h->ev_done, h->cstream));
        h->exchange_posted = true;
    }
    if (h->merged) {                                               // own entries of the compact x: a local gather, no message
        const int64_t n_self = h->recv_cut[(size_t)h->rank + 1] - h->recv_cut[h->rank];
        if (n_self) {
            const int grid = (int)std::min<int64_t>((n_self + 255) / 256, 4096);
            hipLaunchKernelGGL(dist_pack_kernel, dim3(grid), dim3(256), 0, s, n_self, h->d_want + h->recv_cut[h->rank], x_local_dev, h->d_xrem + h->recv_cut[h->rank]);
            G4S_HIP_TRY(hipGetLastError());
        }
        return G4S_OK;
    }
    // the part of the product that needs nothing from anybody runs while the entries travel
    if (h->local_rows == 0) return G4S_OK;                         // (posted its sends, owns no rows: y may be NULL)
    return g4s_spmv(h->A_own, x_local_dev, y_local_dev, 1.0, 0.0, stream);
}

G4S_API g4s_status g4s_spmv_dist_finish(g4s_spmv_dist_t h, double *y_local_dev, void *stream)
{
    G4S_REQUIRE(h && (y_local_dev || h->local_rows == 0), "NULL argument");   // (a rank of a rectangular operator may own x entries but no rows)
    G4S_REQUIRE_LIVE(h);
    hipStream_t s = g4s::as_stream(stream);
    if (h->columns) {
        if (h->comm && h->world > 1 && h->local_rows) G4S_RCCL_TRY(g_rccl.AllReduce(y_local_dev, y_local_dev, (size_t)h->local_rows, ncclDouble, ncclSum, h->comm, s));
        return G4S_OK;
    }
    if (h->exchange_posted) G4S_HIP_TRY(hipStreamWaitEvent(s, h->ev_done, 0));
    h->exchange_posted = false;
    if (h->local_rows == 0) return G4S_OK;
    if (h->merged) return g4s_spmv(h->A_rem, h->d_xrem, y_local_dev, 1.0, 0.0, stream);
    if (h->nnz_rem == 0) return G4S_OK;
    if (h->n_rem_rows) {
        G4S_TRY(g4s_spmv(h->A_rem, h->d_xrem, h->d_rem_y, 1.0, 0.0, stream));
        hipLaunchKernelGGL(dist_add_rows_kernel, dim3((h->n_rem_rows + 255) / 256), dim3(256), 0, s, h->n_rem_rows, h->d_rem_rows, h->d_rem_y, y_local_dev);
        G4S_HIP_TRY(hipGetLastError());
        return G4S_OK;
    }
    return g4s_spmv(h->A_rem, h->d_xrem, y_local_dev, 1.0, 1.0, stream);
}

namespace g4s {
// the smallest slab of the partition (the same on every rank): the solvers on a partitioned operator need at least one row everywhere, and can refuse a
// partition that has an empty rank on ALL ranks alike, before their first collective
int64_t dist_smallest_slab(g4s_spmv_dist_t h)
{
    int64_t m = INT64_MAX;
    for (size_t k = 0; k + 1 < h->off.size(); ++k) m = std::min(m, h->off[k + 1] - h->off[k]);
    return h->off.size() < 2 ? 0 : m;
}
} // namespace g4s

G4S_API g4s_status g4s_spmv_dist_apply(g4s_spmv_dist_t h, const double *x_local_dev, double *y_local_dev, void *stream)
{
    G4S_REQUIRE(h, "NULL handle");
    if (h->columns && h->world > 1 && !h->comm)
        return g4s::set_error(G4S_ERR_INVALID, "g4s_spmv_dist_apply on a column partition needs g4s_spmv_dist_connect_rccl; with another transport all-reduce y between _begin and _finish");
    if ((h->world > 1 || (h->loopback && !h->allgather)) && !h->comm && (h->n_ref || h->give_cut[h->nseg]))
        return g4s::set_error(G4S_ERR_INVALID, "g4s_spmv_dist_apply needs g4s_spmv_dist_connect_rccl; with another transport use _begin / _buffers / _finish");
    G4S_TRY(g4s_spmv_dist_begin(h, x_local_dev, y_local_dev, stream));
    return g4s_spmv_dist_finish(h, y_local_dev, stream);
}

// ---------------------------------------------------------------------------------------------- communicator helpers
G4S_API g4s_status g4s_comm_unique_id(void *id128)
{
    G4S_REQUIRE(id128, "NULL argument");
    G4S_TRY(rccl_load());
    ncclUniqueId id;
    G4S_RCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return G4S_OK;
}

G4S_API g4s_status g4s_comm_create(void **comm, int32_t world, int32_t rank, const void *id128)
{
    G4S_REQUIRE(comm && id128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
    G4S_TRY(rccl_load());
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    G4S_RCCL_TRY(g_rccl.CommInitRank(&c, world, id, rank));
    *comm = c;
    return G4S_OK;
}

G4S_API g4s_status g4s_comm_destroy(void *comm)
{
    if (!comm) return G4S_OK;
    {   // a communicator that a failed set-up exchange aborted (dist_poison) is gone already: ncclCommAbort freed it — nothing to destroy a second time
        std::lock_guard<std::mutex> lk(g_aborted_mu);
        auto it = std::find(g_aborted_comms.begin(), g_aborted_comms.end(), comm);
        if (it != g_aborted_comms.end()) { g_aborted_comms.erase(it); return G4S_OK; }
    }
    G4S_TRY(rccl_load());
    G4S_RCCL_TRY(g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(comm)));
    return G4S_OK;
}

G4S_API g4s_status g4s_comm_allreduce_sum_f64(void *comm, double *buf_dev, int64_t count, void *stream)
{
    G4S_REQUIRE(comm && (buf_dev || count == 0) && count >= 0, "bad argument");
    G4S_REQUIRE(!comm_was_aborted(comm), "this communicator was aborted by a failed set-up exchange: it cannot be used again");
    G4S_TRY(rccl_load());
    if (count) G4S_RCCL_TRY(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(comm), g4s::as_stream(stream)));
    return G4S_OK;
}
