// prims.hpp — the two device-wide primitives the timed call paths need, hand-written for gfx950 (64-lane waves, DPP scans, ballots):
// an exclusive prefix sum and a stable key–value radix sort in descending key order. They replace hipcub::DeviceScan /
// DeviceRadixSort on the SpGEMM call path and in the SpMV plan (hipcub is CUB's interface re-hosted on rocPRIM); plan construction of
// the blocked SpMV path, which sorts 1e8 keys once per matrix, calls rocPRIM directly (spmv_pb.hip).
//   exclusive_scan(in, out, n)     out[i] = Σ_{j<i} in[j], i = 0 … n−1 (so out[n−1] is the sum of the first n−1 inputs; callers that want the
//                                  total pass n+1 with a zero behind the data — the reference's seq_scan has the same shape, utility.h:156-163)
//   sort_pairs_descending(keys, vals, n, key_bits)   4-bit digits, least significant first, ⌈key_bits/4⌉ passes; ties keep their input order
// Sizes here: n ≤ a few million (rows of a matrix, words of a bitmap); everything is enqueued on the caller's stream, scratch comes from
// the library's stream-ordered pool (runtime.cpp), nothing synchronises.
#pragma once
#include "common.hpp"

namespace g4s {
namespace prims {

constexpr int kScanThreads = 256, kScanPer = 8, kScanTile = kScanThreads * kScanPer;

template <typename T>
__device__ __forceinline__ T wave_inclusive(T v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T u = __shfl_up(v, off, 64);
        if ((int)(threadIdx.x & 63) >= off) v += u;
    }
    return v;
}

// phase 1: tile sums
template <typename T>
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(long long n, const T *__restrict__ in, T *__restrict__ sums)
{
    __shared__ T s[kScanThreads / 64];
    const long long base = (long long)blockIdx.x * kScanTile;
    T v = 0;
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
        const long long i = base + u * kScanThreads + threadIdx.x;
        if (i < n) v += in[i];
    }
    v = wave_inclusive(v);
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) t += s[w];
        sums[blockIdx.x] = t;
    }
}

// phase 2: one workgroup turns the tile sums into tile offsets (chunks of 256 with a running carry)
template <typename T>
__global__ __launch_bounds__(kScanThreads) void scan_tile_offsets_kernel(int ntiles, T *__restrict__ sums)
{
    __shared__ T s[kScanThreads / 64];
    __shared__ T carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < ntiles; b0 += kScanThreads) {
        const int i = b0 + (int)threadIdx.x;
        const T v = i < ntiles ? sums[i] : 0;
        T incl = wave_inclusive(v);
        if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
        __syncthreads();
        T before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += s[w];
        if (i < ntiles) sums[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) carry = before + incl;
        __syncthreads();
    }
}

// phase 3: every tile scans itself behind its offset. Thread t owns kScanPer CONSECUTIVE elements (a serial scan in registers, one wave
// scan of the per-thread totals): the loads of a thread are then strided by kScanPer elements, which L2 absorbs at these sizes.
template <typename T>
__global__ __launch_bounds__(kScanThreads) void scan_write_kernel(long long n, const T *__restrict__ in, const T *__restrict__ tile_off, T *__restrict__ out)
{
    __shared__ T s[kScanThreads / 64];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanPer;
    T loc[kScanPer], sum = 0;
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
        loc[u] = base + u < n ? in[base + u] : 0;
        sum += loc[u];
    }
    const T incl = wave_inclusive(sum);
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    T run = tile_off[blockIdx.x] + incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += s[w];
#pragma unroll
    for (int u = 0; u < kScanPer; ++u) {
        if (base + u < n) out[base + u] = run;
        run += loc[u];
    }
}

// a short input (round 5): one workgroup, one launch — thread t owns ceil(n / 1024) consecutive elements (the three-launch form costs 15–20 µs of launches and
// boundaries whatever n is, and a small product makes dozens of scans over a few thousand rows)
constexpr int kScanSmallThreads = 1024, kScanSmallMax = kScanSmallThreads * 8;   // (up to 64 per thread measured 124 µs for 60 K entries — a thread's consecutive elements are strided across the wave — against ≈ 20 µs for the three launches)
template <typename T>
__global__ __launch_bounds__(kScanSmallThreads) void scan_small_kernel(int n, const T *__restrict__ in, T *__restrict__ out)
{
    __shared__ T s[kScanSmallThreads / 64];
    const int per = (n + kScanSmallThreads - 1) / kScanSmallThreads, i0 = (int)threadIdx.x * per, i1 = min(n, i0 + per);
    T sum = 0;
    for (int i = i0; i < i1; ++i) sum += in[i];
    const T incl = wave_inclusive(sum);
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    T run = incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += s[w];
    for (int i = i0; i < i1; ++i) { const T v = in[i]; out[i] = run; run += v; }
}

template <typename T>
int exclusive_scan(const T *in, T *out, long long n, hipStream_t s)
{
    if (n <= 0) return G4S_OK;
    if (n <= kScanSmallMax && in != out) {
        hipLaunchKernelGGL(scan_small_kernel<T>, dim3(1), dim3(kScanSmallThreads), 0, s, (int)n, in, out);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return set_error(G4S_ERR_HIP, "exclusive_scan: %s", hipGetErrorString(e));
        return G4S_OK;
    }
    const int ntiles = (int)((n + kScanTile - 1) / kScanTile);
    void *sums = nullptr;
    G4S_TRY(scratch_alloc(&sums, sizeof(T) * (size_t)ntiles, s));
    hipLaunchKernelGGL(scan_tile_sums_kernel<T>, dim3(ntiles), dim3(kScanThreads), 0, s, n, in, static_cast<T *>(sums));
    hipLaunchKernelGGL(scan_tile_offsets_kernel<T>, dim3(1), dim3(kScanThreads), 0, s, ntiles, static_cast<T *>(sums));
    hipLaunchKernelGGL(scan_write_kernel<T>, dim3(ntiles), dim3(kScanThreads), 0, s, n, in, static_cast<const T *>(sums), out);
    const hipError_t e = hipGetLastError();
    scratch_free(sums, s);
    if (e != hipSuccess) return set_error(G4S_ERR_HIP, "exclusive_scan: %s", hipGetErrorString(e));
    return G4S_OK;
}

// ------------------------------------------------------------------------------------------------ radix sort, 4-bit digits, descending
constexpr int kSortThreads = 256, kSortPer = 8, kSortTile = kSortThreads * kSortPer, kDigits = 16;

// bucket of a key in this pass: descending order = the digit reversed
__device__ __forceinline__ int sort_bucket(int key, int shift) { return kDigits - 1 - (int)(((unsigned)key >> shift) & (kDigits - 1)); }

namespace {   // (plain kernels in a header: one copy per translation unit)
// counts[d · ntiles + tile]: digit-major, so that one exclusive scan over the table gives every (digit, tile) its first output position
__global__ __launch_bounds__(kSortThreads) void sort_count_kernel(int n, const int *__restrict__ keys, int shift, int ntiles, int *__restrict__ counts)
{
    __shared__ int h[kDigits];
    if (threadIdx.x < kDigits) h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortTile;
#pragma unroll
    for (int u = 0; u < kSortPer; ++u) {
        const int i = base + u * kSortThreads + (int)threadIdx.x;
        const int b = i < n ? sort_bucket(keys[i], shift) : -1;
#pragma unroll
        for (int d = 0; d < kDigits; ++d) {                        // one ballot per digit value: a wave adds its count with a single atomic
            const unsigned long long m = __ballot(b == d);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&h[d], __popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < kDigits) counts[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter: the tile is walked in the same order as it was counted (sub-tiles of 256 consecutive elements); inside a sub-tile an
// element's rank among the equal digits before it = lanes below it in its wave (ballot) + the earlier waves' counts + the running base.
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(int n, const int *__restrict__ keys, const int *__restrict__ vals, int shift, int ntiles,
                                                                     const int *__restrict__ first, int *__restrict__ keys_out, int *__restrict__ vals_out)
{
    __shared__ int base_of[kDigits];
    __shared__ int wave_cnt[kSortThreads / 64][kDigits];
    if (threadIdx.x < kDigits) base_of[threadIdx.x] = first[threadIdx.x * ntiles + blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int tile0 = blockIdx.x * kSortTile;
    for (int u = 0; u < kSortPer; ++u) {
        const int i = tile0 + u * kSortThreads + (int)threadIdx.x;
        const int key = i < n ? keys[i] : 0, val = i < n ? vals[i] : 0;
        const int b = i < n ? sort_bucket(key, shift) : -1;
        int rank = 0;
#pragma unroll
        for (int d = 0; d < kDigits; ++d) {
            const unsigned long long m = __ballot(b == d);
            if (b == d) rank = __popcll(m & below);
            if (lane == 0) wave_cnt[wave][d] = __popcll(m);
        }
        __syncthreads();                                           // also orders base_of's first write
        if (b >= 0) {
            int pos = base_of[b] + rank;
            for (int w = 0; w < wave; ++w) pos += wave_cnt[w][b];
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        if (threadIdx.x < kDigits) {
            int t = 0;
            for (int w = 0; w < kSortThreads / 64; ++w) t += wave_cnt[w][threadIdx.x];
            base_of[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

} // namespace

// A short list by descending key in ONE launch (round 5): one workgroup counts the keys (< 2^11: the logarithmic size keys of the row lists) in LDS, scans the
// 2 048 counters and places the pairs through per-key cursors. NOT stable — equal keys come out in any order — so only for callers to whom the order among equal
// keys means nothing (the longest-first row lists: it decides which workgroup takes a row first, never a result).
constexpr int kSortSmallMax = 1 << 16, kSortSmallBins = 2048;
namespace {
__global__ __launch_bounds__(1024) void sort_small_desc_kernel(int n, const int *__restrict__ keys, const int *__restrict__ vals, int *__restrict__ keys_out, int *__restrict__ vals_out)
{
    __shared__ int cnt[kSortSmallBins];
    __shared__ int wsum[16];
    for (int i = threadIdx.x; i < kSortSmallBins; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) atomicAdd(&cnt[kSortSmallBins - 1 - (keys[i] & (kSortSmallBins - 1))], 1);   // bin 0 = the largest key
    __syncthreads();
    // exclusive scan of the 2 048 counters: two per thread
    const int a = cnt[2 * threadIdx.x], b = cnt[2 * threadIdx.x + 1];
    const int incl = wave_inclusive(a + b);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = incl - (a + b);
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    __syncthreads();
    cnt[2 * threadIdx.x] = base;
    cnt[2 * threadIdx.x + 1] = base + a;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int k = keys[i], pos = atomicAdd(&cnt[kSortSmallBins - 1 - (k & (kSortSmallBins - 1))], 1);
        keys_out[pos] = k;
        vals_out[pos] = vals[i];
    }
}
} // namespace
inline int sort_pairs_descending_small_unstable(const int *keys_in, const int *vals_in, int *keys_out, int *vals_out, int n, hipStream_t s)
{
    if (n <= 0) return G4S_OK;
    hipLaunchKernelGGL(sort_small_desc_kernel, dim3(1), dim3(1024), 0, s, n, keys_in, vals_in, keys_out, vals_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(G4S_ERR_HIP, "sort_pairs_descending_small_unstable: %s", hipGetErrorString(e));
    return G4S_OK;
}

// keys_in / vals_in are not modified; the result lands in keys_out / vals_out. key_bits: an upper bound on the significant bits of the
// (non-negative) keys. tmp_keys / tmp_vals: n ints each (ping-pong partners of the outputs).
inline int sort_pairs_descending(const int *keys_in, const int *vals_in, int *keys_out, int *vals_out, int *tmp_keys, int *tmp_vals, int n, int key_bits, hipStream_t s)
{
    if (n <= 0) return G4S_OK;
    const int kb = key_bits < 1 ? 1 : (key_bits > 31 ? 31 : key_bits), passes = (kb + 3) / 4;
    const int ntiles = (n + kSortTile - 1) / kSortTile;
    void *cnt = nullptr, *pos = nullptr;
    const size_t tb = sizeof(int) * ((size_t)kDigits * ntiles + 1);
    G4S_TRY(scratch_alloc(&cnt, tb, s));
    if (scratch_alloc(&pos, tb, s) != G4S_OK) { scratch_free(cnt, s); return G4S_ERR_NOMEM; }
    int st = G4S_OK;
    // an odd number of passes ends in the outputs when the first pass writes there; an even number when it writes to the partners
    const int *src_k = keys_in, *src_v = vals_in;
    int *dst_k = (passes & 1) ? keys_out : tmp_keys, *dst_v = (passes & 1) ? vals_out : tmp_vals;
    for (int p = 0; p < passes && st == G4S_OK; ++p) {
        hipLaunchKernelGGL(sort_count_kernel, dim3(ntiles), dim3(kSortThreads), 0, s, n, src_k, 4 * p, ntiles, static_cast<int *>(cnt));
        st = exclusive_scan(static_cast<const int *>(cnt), static_cast<int *>(pos), (long long)kDigits * ntiles, s);
        if (st != G4S_OK) break;
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(ntiles), dim3(kSortThreads), 0, s, n, src_k, src_v, 4 * p, ntiles, static_cast<const int *>(pos), dst_k, dst_v);
        src_k = dst_k; src_v = dst_v;
        dst_k = dst_k == keys_out ? tmp_keys : keys_out;
        dst_v = dst_v == vals_out ? tmp_vals : vals_out;
    }
    const hipError_t e = hipGetLastError();
    scratch_free(pos, s);
    scratch_free(cnt, s);
    if (st == G4S_OK && e != hipSuccess) st = set_error(G4S_ERR_HIP, "sort_pairs_descending: %s", hipGetErrorString(e));
    return st;
}

} // namespace prims
} // namespace g4s
