// msq_cluster.hip -- msq_phase_cluster: reads with equal codon tuples collapsed on the device.
//
// AminoAcidCaller::PhaseVariants compares every read's codon tuple with the tuples seen so far and either adds the
// read's name to that observation or opens a new one (src/AminoAcidCaller.cpp:162-196): a partition of the reads
// by exact tuple equality, observations in first-occurrence order, names inside one in read order.  On key rows
// (msq_phase_codons: two reads have equal tuples exactly when their key rows are equal) that is a sort-reduce:
//
//   hash every row (64 bit) -> stable radix sort of (hash, read) -> a segment starts where the hash or -- compared
//   in full -- the row changes -> segment index by prefix sum -> every segment's first read is its smallest (the
//   sort is stable) -> segments ranked by first read = first-occurrence order -> cluster id per read.
//
// The rows are always compared in full, so two different rows never share a cluster; what a 64-bit collision could
// do is interleave two different rows inside one run of equal hashes (A B A), which would split A.  That case is
// detected (a segment whose hash equals its predecessor's) and reported as MSQ_ERR_RANGE: the caller falls back to
// its own pass (the C++ front's host clustering); it takes ~2^-64 * n^2 to see one.
#include "msq_internal.h"

#include <hipcub/hipcub.hpp>

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// Short rows (K <= kStagedKeys): one block per 256 reads -- their key rows are one contiguous piece of memory, read
// coalesced into LDS (row stride an odd number of dwords: no bank conflicts), then one thread per row.
constexpr int kStagedKeys = 62;
__global__ __launch_bounds__(256) void hash_rows_staged_kernel(const uint16_t* __restrict__ keys, uint64_t n, int K, uint64_t* __restrict__ hash,
                                                               uint32_t* __restrict__ idx)
{
    __shared__ uint16_t stage[256 * (kStagedKeys + 2)];
    const uint32_t tid = threadIdx.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * 256;
    const uint32_t rows = (uint32_t)min((uint64_t)256, n - r0);
    const uint32_t Ku = (uint32_t)K, S = (((Ku + 1u) >> 1) | 1u) * 2u;  // u16 per staged row; K >= 1
    const uint16_t* base = keys + r0 * (uint64_t)K;
    const uint32_t total = rows * Ku, dr = 256u / Ku, dk = 256u % Ku;
    uint32_t r = tid / Ku, k = tid % Ku;  // element tid + 256 j = key k of row r
    for (uint32_t i = tid; i < total; i += 256) {
        stage[r * S + k] = base[i];
        r += dr;
        k += dk;
        if (k >= Ku) {
            k -= Ku;
            ++r;
        }
    }
    __syncthreads();
    if (tid < rows) {
        uint64_t h = 0;
        for (uint32_t j = 0; j < Ku; ++j) h += mix64(((uint64_t)(j + 1) << 16) | stage[tid * S + j]);
        hash[r0 + tid] = mix64(h);
        idx[r0 + tid] = (uint32_t)(r0 + tid);
    }
}

// Longer rows: one wave per 64 reads, lane k walks the rows' k-th, k + 64-th ... keys (coalesced over the row); the
// lanes' partial hashes are position-dependent, so that their sum does not depend on the order they are added in
__global__ __launch_bounds__(256) void hash_rows_kernel(const uint16_t* __restrict__ keys, uint64_t n, int K, uint64_t* __restrict__ hash,
                                                        uint32_t* __restrict__ idx)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t r0 = wave * 64;
    for (uint64_t r = r0; r < r0 + 64 && r < n; ++r) {
        uint64_t h = 0;
        for (int k = lane; k < K; k += 64) h += mix64(((uint64_t)(k + 1) << 16) | keys[r * (uint64_t)K + k]);
        for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d);
        if (lane == 0) {
            hash[r] = mix64(h);
            idx[r] = (uint32_t)r;
        }
    }
}

// flag[i] = 1 where sorted element i starts a segment; err: a segment that starts inside a run of equal hashes
__global__ __launch_bounds__(256) void segment_flags_kernel(const uint16_t* __restrict__ keys, int K, const uint64_t* __restrict__ hash,
                                                            const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ flag,
                                                            uint32_t* __restrict__ collision)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 1;
    if (i > 0 && hash[i] == hash[i - 1]) {
        const uint16_t* a = keys + (uint64_t)idx[i] * K;
        const uint16_t* b = keys + (uint64_t)idx[i - 1] * K;
        f = 0;
        const uintptr_t base = reinterpret_cast<uintptr_t>(keys);
        if ((K & 3) == 0 && (base & 7) == 0) {  // rows of a multiple of four keys start on 8-byte boundaries: four keys per compare
            const uint64_t* a8 = reinterpret_cast<const uint64_t*>(a);
            const uint64_t* b8 = reinterpret_cast<const uint64_t*>(b);
            uint64_t diff = 0;
            for (int k = 0; k < (K >> 2); ++k) diff |= a8[k] ^ b8[k];
            f = diff != 0;
        } else if ((K & 1) == 0 && (base & 3) == 0) {
            const uint32_t* a4 = reinterpret_cast<const uint32_t*>(a);
            const uint32_t* b4 = reinterpret_cast<const uint32_t*>(b);
            uint32_t diff = 0;
            for (int k = 0; k < (K >> 1); ++k) diff |= a4[k] ^ b4[k];
            f = diff != 0;
        } else {
            for (int k = 0; k < K; ++k)
                if (a[k] != b[k]) {
                    f = 1;
                    break;
                }
        }
        if (f) atomicOr(collision, 1u);
    }
    flag[i] = f;
}

// seg[i] (inclusive scan of the flags) - 1 = segment of sorted element i; its first element carries the segment's
// first read; the length follows from the next segment's start
__global__ __launch_bounds__(256) void segment_heads_kernel(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ seg,
                                                            const uint32_t* __restrict__ idx, uint64_t n, uint32_t* __restrict__ first_read,
                                                            uint32_t* __restrict__ start, uint32_t* __restrict__ seg_id)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint32_t s = seg[i] - 1;
    first_read[s] = idx[i];
    start[s] = (uint32_t)i;
    seg_id[s] = s;
}

// segments sorted by first read: rank j <- segment order[j]
__global__ __launch_bounds__(256) void rank_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ start, uint64_t n_seg,
                                                   uint64_t n, uint32_t* __restrict__ rank_of_seg, uint32_t* __restrict__ count_out)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    const uint32_t s = order[j];
    rank_of_seg[s] = (uint32_t)j;
    const uint32_t end = s + 1 < n_seg ? start[s + 1] : (uint32_t)n;
    count_out[j] = end - start[s];
}

__global__ __launch_bounds__(256) void assign_kernel(const uint32_t* __restrict__ seg, const uint32_t* __restrict__ idx,
                                                     const uint32_t* __restrict__ rank_of_seg, uint64_t n, uint32_t* __restrict__ cluster_of)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cluster_of[idx[i]] = rank_of_seg[seg[i] - 1];
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ keys, int K, const uint32_t* __restrict__ first_read,
                                                          uint64_t n_clusters, uint16_t* __restrict__ rep)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_clusters * (uint64_t)K) return;
    const uint64_t j = t / (uint64_t)K, k = t - j * (uint64_t)K;
    rep[t] = keys[(uint64_t)first_read[j] * K + k];
}

inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int msq_phase_cluster(msq_ctx* ctx, const uint16_t* d_keys, uint64_t n_reads, int32_t n_pos, uint32_t* d_cluster_of,
                                 uint32_t* d_first_read, uint32_t* d_count, uint16_t* d_rep_keys, uint64_t rep_capacity,
                                 uint64_t* n_clusters)
{
    if (!ctx || !n_clusters) return msq_fail(MSQ_ERR_ARG, "msq_phase_cluster: NULL argument");
    *n_clusters = 0;
    if (n_pos < 0 || n_pos > (1 << 20)) return msq_fail(MSQ_ERR_ARG, "msq_phase_cluster: row length %d out of range (0 .. 2^20)", n_pos);
    if (n_reads == 0) return MSQ_OK;
    if (n_reads > 0x7fffffffull) return msq_fail(MSQ_ERR_ARG, "msq_phase_cluster: too many reads (hipCUB counts in int)");
    if ((n_pos > 0 && !d_keys) || !d_cluster_of || !d_first_read || !d_count) return msq_fail(MSQ_ERR_ARG, "msq_phase_cluster: NULL device buffer");
    MSQ_HIP(hipSetDevice(ctx->device));
    const uint64_t n = n_reads;
    const int K = n_pos;
    // scratch: hash, hash', idx, idx', flag, seg, first, start, seg_id, order, first', rank, collision + hipcub's own
    size_t sort_bytes = 0, scan_bytes = 0, sort2_bytes = 0;
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                               (uint32_t*)nullptr, (int)n, 0, 64, ctx->stream));
    MSQ_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, ctx->stream));
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort2_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                               (uint32_t*)nullptr, (int)n, 0, 32, ctx->stream));
    const size_t tmp_bytes = std::max(sort_bytes, std::max(scan_bytes, sort2_bytes));
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t total = 2 * pad(n * 8) + 10 * pad(n * 4) + 256 + pad(tmp_bytes);
    void* s = nullptr;
    int rc = msq_scratch(ctx, total, &s);
    if (rc != MSQ_OK) return rc;
    char* p = (char*)s;
    auto take = [&](size_t b) {
        char* q = p;
        p += pad(b);
        return q;
    };
    uint64_t* hash = (uint64_t*)take(n * 8);
    uint64_t* hash_s = (uint64_t*)take(n * 8);
    uint32_t* idx = (uint32_t*)take(n * 4);
    uint32_t* idx_s = (uint32_t*)take(n * 4);
    uint32_t* flag = (uint32_t*)take(n * 4);
    uint32_t* seg = (uint32_t*)take(n * 4);
    uint32_t* first = (uint32_t*)take(n * 4);
    uint32_t* start = (uint32_t*)take(n * 4);
    uint32_t* seg_id = (uint32_t*)take(n * 4);
    uint32_t* order = (uint32_t*)take(n * 4);
    uint32_t* first_sorted = (uint32_t*)take(n * 4);
    uint32_t* rank = (uint32_t*)take(n * 4);
    uint32_t* collision = (uint32_t*)take(256);
    void* tmp = take(tmp_bytes);
    size_t tb = tmp_bytes;

    MSQ_HIP(hipMemsetAsync(collision, 0, 4, ctx->stream));
    if (K >= 1 && K <= kStagedKeys)
        hipLaunchKernelGGL(hash_rows_staged_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_keys, n, K, hash, idx);
    else
        hipLaunchKernelGGL(hash_rows_kernel, dim3(blocks_for(((n + 63) / 64) * 64)), dim3(256), 0, ctx->stream, d_keys, n, K, hash, idx);
    MSQ_HIP(hipGetLastError());
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const uint64_t*)hash, hash_s, (const uint32_t*)idx, idx_s, (int)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(segment_flags_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, d_keys, K, (const uint64_t*)hash_s,
                       (const uint32_t*)idx_s, n, flag, collision);
    MSQ_HIP(hipGetLastError());
    tb = tmp_bytes;
    MSQ_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tb, (const uint32_t*)flag, seg, (int)n, ctx->stream));
    uint32_t host[2] = {0, 0};  // segments, collision
    MSQ_HIP(hipMemcpyAsync(&host[0], seg + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipMemcpyAsync(&host[1], collision, 4, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (host[1]) return msq_fail(MSQ_ERR_RANGE, "msq_phase_cluster: two different key rows share a 64-bit hash; cluster on the host");
    const uint64_t n_seg = host[0];
    *n_clusters = n_seg;
    hipLaunchKernelGGL(segment_heads_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, (const uint32_t*)flag, (const uint32_t*)seg,
                       (const uint32_t*)idx_s, n, first, start, seg_id);
    MSQ_HIP(hipGetLastError());
    tb = tmp_bytes;
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const uint32_t*)first, first_sorted, (const uint32_t*)seg_id, order, (int)n_seg, 0, 32,
                                               ctx->stream));
    hipLaunchKernelGGL(rank_kernel, dim3(blocks_for(n_seg)), dim3(256), 0, ctx->stream, (const uint32_t*)order, (const uint32_t*)start, n_seg, n,
                       rank, d_count);
    MSQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(assign_kernel, dim3(blocks_for(n)), dim3(256), 0, ctx->stream, (const uint32_t*)seg, (const uint32_t*)idx_s,
                       (const uint32_t*)rank, n, d_cluster_of);
    MSQ_HIP(hipGetLastError());
    MSQ_HIP(hipMemcpyAsync(d_first_read, first_sorted, n_seg * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_rep_keys && K > 0) {
        if (n_seg > rep_capacity) {
            MSQ_HIP(hipStreamSynchronize(ctx->stream));
            return msq_fail(MSQ_ERR_RANGE, "msq_phase_cluster: %llu clusters, room for %llu representative rows", (unsigned long long)n_seg,
                            (unsigned long long)rep_capacity);
        }
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for(n_seg * (uint64_t)K)), dim3(256), 0, ctx->stream, d_keys, K,
                           (const uint32_t*)d_first_read, n_seg, d_rep_keys);
        MSQ_HIP(hipGetLastError());
    }
    MSQ_HIP(hipStreamSynchronize(ctx->stream));  // the scratch is the context's: the next call may regrow it
    return MSQ_OK;
}
