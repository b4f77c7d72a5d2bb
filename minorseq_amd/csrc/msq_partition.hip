// msq_partition.hip -- column-slab partition of a packed read set, for windows wider than one slab.
//
// The streaming pileup keeps one batch of rows in LDS, which bounds a launch to a slab of <= 4096 window columns;
// a wider window (the 9.7 kb HIV genome, 20 kb amplicons) takes one launch per slab.  Run over the whole read set,
// every such launch streams and decodes every read again only to drop the bases outside its slab.  The partition
// is the HBM layout that avoids it: per slab a packed read set of its own that holds, of every read, just the
// 32-base groups whose columns fall into the slab (plus the two columns to its right that the slab's last codons
// reach into).  Groups are copied as they are -- a sub-read starts on a group boundary like any read, its
// ReferenceStart is the column of its first group -- so the slab's launch sees ordinary reads and the counts are
// the same (MSAByRow::AddRead places a base by its column alone, src/MSA.cpp:158-171).  It is a pack-time step:
// one scan of the bit planes (per-group column offsets: a wave prefix sum of popcounts) and one gather.
#include "msq_internal.h"

#include <hipcub/hipcub.hpp>

#include <new>
#include <vector>

void msq_slab_layout(int32_t W, int max_words, int* n_slabs, int* Gslab);
int msq_pileup_stream_slab_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                                  int32_t W, int slab, int slab_words, uint32_t* d_nuc, uint32_t* d_codon, bool* handled);
int msq_pileup_stream_slabs_launch(msq_ctx* ctx, const msq_packed_reads* const* sets, int n_sets, const msq_qv_thresholds& thr,
                                   int32_t win_begin0, int32_t W, int slab_words, uint32_t* d_nuc, uint32_t* d_codon, bool* handled);

struct msq_partition {
    int32_t win_begin0 = 0, W = 0;
    int slab_words = 128;  // the layout's parameter: both the split and the launches use it
    std::vector<msq_packed_dev*> sets;
};

namespace {
constexpr int kMaxSlabs = 64;

// One wave per read: the column (relative to ReferenceStart) at which every group starts, and from it, per slab,
// the run of groups [j0, j1) that holds a base in window columns [lo, hi + 2).
__global__ __launch_bounds__(256) void slab_ranges_kernel(const int32_t* __restrict__ ref_start, const int32_t* __restrict__ ref_end,
                                                          const uint64_t* __restrict__ group_off, const uint4* __restrict__ bop,
                                                          uint64_t n_reads, int32_t win_begin0, int32_t W, int n_slabs, int slab_cols,
                                                          uint32_t* __restrict__ first_group /* [n_slabs][n_reads] */,
                                                          uint64_t* __restrict__ n_groups /* [n_slabs][n_reads + 1] */,
                                                          int32_t* __restrict__ sub_start, int32_t* __restrict__ sub_end /* [n_slabs][n_reads] */)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t g_begin = group_off[r], g_end = group_off[r + 1];
        const int32_t col_base = ref_start[r] - win_begin0;
        // per slab: first / one-past-last group seen so far, and the columns at which they start
        int32_t j0 = 0, j1 = 0, c0 = 0, c1 = 0;  // lane s holds slab s
        bool found = false;
        int32_t carry = 0;
        for (uint64_t blk = g_begin; blk < g_end; blk += 64) {
            const uint64_t g = blk + (uint64_t)lane;
            const bool valid = g < g_end;
            const int ncons = valid ? __popc(~bop[g].w) : 0;  // ops 0 and 1 consume a column
            int incl = ncons;
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const int32_t cs = col_base + carry + incl - ncons, ce = cs + ncons;  // the group's columns [cs, ce)
            for (int s = 0; s < n_slabs; ++s) {
                const int32_t lo = s * slab_cols, hi = min(W, (s + 1) * slab_cols) + 2;
                const uint64_t hit = __ballot(valid && ncons > 0 && cs < hi && ce > lo);
                if (!hit) continue;
                const int first = __ffsll((long long)hit) - 1, last = 63 - __clzll((long long)hit);
                const int32_t cs_first = __shfl(cs, first), ce_last = __shfl(ce, last);
                if (lane == s) {
                    if (!found) {
                        j0 = (int32_t)(blk - g_begin) + first;
                        c0 = cs_first;
                        found = true;
                    }
                    j1 = (int32_t)(blk - g_begin) + last + 1;
                    c1 = ce_last;
                }
            }
            carry += __shfl(incl, 63);
        }
        if (lane < n_slabs) {
            const uint64_t at = (uint64_t)lane * n_reads + r;
            first_group[at] = found ? (uint32_t)j0 : 0u;
            n_groups[(uint64_t)lane * (n_reads + 1) + r] = found ? (uint64_t)(j1 - j0) : 0ull;
            // the sub-read's reference interval: the columns of its groups (an empty one sits at the read's start)
            sub_start[at] = win_begin0 + (found ? c0 : col_base);
            sub_end[at] = win_begin0 + (found ? c1 : col_base);
        }
        (void)ref_end;
    }
}

// The groups of one slab's sub-reads, copied as they are: one lane per (group, 16-byte piece).
__global__ __launch_bounds__(256) void slab_gather_kernel(const uint64_t* __restrict__ src_group_off, const uint32_t* __restrict__ first_group,
                                                          const uint64_t* __restrict__ dst_group_off, uint64_t n_reads, uint64_t n_dst_groups,
                                                          const uint4* __restrict__ src_bop, uint4* __restrict__ dst_bop,
                                                          const uint4* __restrict__ s0, uint4* __restrict__ d0, const uint4* __restrict__ s1,
                                                          uint4* __restrict__ d1, const uint4* __restrict__ s2, uint4* __restrict__ d2,
                                                          const uint4* __restrict__ s3, uint4* __restrict__ d3)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_dst_groups) return;
    uint64_t lo = 0, hi = n_reads;  // the last read r with dst_group_off[r] <= g
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (dst_group_off[mid] <= g)
            lo = mid;
        else
            hi = mid;
    }
    const uint64_t src = src_group_off[lo] + first_group[lo] + (g - dst_group_off[lo]);
    dst_bop[g] = src_bop[src];
    if (s0) { d0[2 * g] = s0[2 * src]; d0[2 * g + 1] = s0[2 * src + 1]; }
    if (s1) { d1[2 * g] = s1[2 * src]; d1[2 * g + 1] = s1[2 * src + 1]; }
    if (s2) { d2[2 * g] = s2[2 * src]; d2[2 * g + 1] = s2[2 * src + 1]; }
    if (s3) { d3[2 * g] = s3[2 * src]; d3[2 * g + 1] = s3[2 * src + 1]; }
}

void free_set(msq_packed_dev* d)
{
    if (!d) return;
    for (int j = 0; j < 12; ++j)
        if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
    delete d;
}
}  // namespace

extern "C" int msq_partition_free(msq_ctx* ctx, msq_partition* p)
{
    if (!p) return MSQ_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (msq_packed_dev* d : p->sets) free_set(d);
    delete p;
    return MSQ_OK;
}

extern "C" int msq_partition_slabs(const msq_partition* p) { return p ? (int)p->sets.size() : 0; }
extern "C" const msq_packed_reads* msq_partition_view(const msq_partition* p, int slab)
{
    return (p && slab >= 0 && slab < (int)p->sets.size()) ? &p->sets[slab]->view : nullptr;
}

extern "C" int msq_reads_partition(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W,
                                   int32_t max_slab_columns, msq_partition** out)
{
    if (!ctx || !d_reads || !out) return msq_fail(MSQ_ERR_ARG, "msq_reads_partition: NULL argument");
    *out = nullptr;
    if (W < 0) return msq_fail(MSQ_ERR_ARG, "msq_reads_partition: negative window width %d", W);
    const uint64_t n = d_reads->n_reads;
    if (n && (!d_reads->ref_start || !d_reads->ref_end || !d_reads->group_off || !d_reads->bop)) return msq_fail(MSQ_ERR_ARG, "msq_reads_partition: NULL read arrays");
    const int slab_words = max_slab_columns <= 0 ? 128 : std::min(128, std::max(1, max_slab_columns / 32));
    int n_slabs = 0, Gslab = 0;
    msq_slab_layout(W, slab_words, &n_slabs, &Gslab);
    if (n_slabs > kMaxSlabs) return msq_fail(MSQ_ERR_ARG, "msq_reads_partition: %d slabs (window of %d columns), at most %d", n_slabs, W, kMaxSlabs);
    MSQ_HIP(hipSetDevice(ctx->device));
    msq_partition* p = new (std::nothrow) msq_partition;
    if (!p) return msq_fail(MSQ_ERR_NOMEM, "msq_reads_partition: out of memory");
    p->win_begin0 = win_begin0;
    p->W = W;
    p->slab_words = slab_words;
    void *d_first = nullptr, *d_cnt = nullptr, *d_off = nullptr, *d_rs = nullptr, *d_re = nullptr, *d_tmp = nullptr;
    auto fail = [&](int code, const char* what, hipError_t e) {
        (void)hipStreamSynchronize(ctx->stream);
        for (void* q : {d_first, d_cnt, d_off, d_rs, d_re, d_tmp})
            if (q) (void)hipFree(q);
        for (msq_packed_dev* d : p->sets) free_set(d);
        delete p;
        return msq_fail(code, "msq_reads_partition: %s: %s", what, hipGetErrorString(e));
    };
    if (n == 0 || n_slabs == 0) {  // nothing to split: as many empty sets as there are slabs
        for (int s = 0; s < n_slabs; ++s) {
            msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
            if (!d) return fail(MSQ_ERR_NOMEM, "out of memory", hipSuccess);
            d->view = *d_reads;
            d->view.n_reads = d->view.n_groups = 0;
            d->view.gcol = nullptr;
            p->sets.push_back(d);
        }
        *out = p;
        return MSQ_OK;
    }
    const size_t cells = (size_t)n_slabs * n;
    hipError_t e = hipMalloc(&d_first, cells * 4);
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (size_t)n_slabs * (n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&d_off, (size_t)n_slabs * (n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&d_rs, cells * 4);
    if (e == hipSuccess) e = hipMalloc(&d_re, cells * 4);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, (size_t)n_slabs * (n + 1) * 8, ctx->stream);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "tables", e);
    {
        const uint64_t blocks = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(slab_ranges_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_reads->ref_start, d_reads->ref_end,
                           d_reads->group_off, (const uint4*)d_reads->bop, n, win_begin0, W, n_slabs, Gslab * 32, (uint32_t*)d_first,
                           (uint64_t*)d_cnt, (int32_t*)d_rs, (int32_t*)d_re);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "range kernel", e);
    }
    // per slab: group_off' = exclusive prefix sum of the group counts (n + 1 entries, the last one the total)
    size_t tmp_bytes = 0;
    e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const uint64_t*)d_cnt, (uint64_t*)d_off, (int)(n + 1), ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
    if (e != hipSuccess) return fail(MSQ_ERR_HIP, "scan storage", e);
    for (int s = 0; s < n_slabs; ++s) {
        e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, (const uint64_t*)d_cnt + (size_t)s * (n + 1), (uint64_t*)d_off + (size_t)s * (n + 1),
                                             (int)(n + 1), ctx->stream);
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "scan", e);
    }
    std::vector<uint64_t> totals((size_t)n_slabs);
    for (int s = 0; s < n_slabs; ++s) {
        e = hipMemcpyAsync(&totals[s], (const uint64_t*)d_off + (size_t)s * (n + 1) + n, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "totals", e);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(MSQ_ERR_HIP, "synchronise", e);
    const void* src_qv[4] = {d_reads->qual, d_reads->delqv, d_reads->subqv, d_reads->insqv};
    for (int s = 0; s < n_slabs; ++s) {
        msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
        if (!d) return fail(MSQ_ERR_NOMEM, "out of memory", hipSuccess);
        p->sets.push_back(d);
        const uint64_t tot = totals[s];
        const size_t bytes[8] = {n * 4, n * 4, (n + 1) * 8, tot * 16, tot * 32, tot * 32, tot * 32, tot * 32};
        for (int k = 0; k < 8; ++k) {
            if (k >= 4 && !src_qv[k - 4]) continue;
            e = hipMalloc(&d->ptrs[k], bytes[k] ? bytes[k] : 16);
            if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "slab set", e);
        }
        e = hipMemcpyAsync(d->ptrs[0], (const int32_t*)d_rs + (size_t)s * n, n * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d->ptrs[1], (const int32_t*)d_re + (size_t)s * n, n * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d->ptrs[2], (const uint64_t*)d_off + (size_t)s * (n + 1), (n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "slab tables", e);
        if (tot) {
            hipLaunchKernelGGL(slab_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_reads->group_off,
                               (const uint32_t*)d_first + (size_t)s * n, (const uint64_t*)d->ptrs[2], n, tot, (const uint4*)d_reads->bop,
                               (uint4*)d->ptrs[3], (const uint4*)src_qv[0], (uint4*)d->ptrs[4], (const uint4*)src_qv[1], (uint4*)d->ptrs[5],
                               (const uint4*)src_qv[2], (uint4*)d->ptrs[6], (const uint4*)src_qv[3], (uint4*)d->ptrs[7]);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(MSQ_ERR_HIP, "gather kernel", e);
        }
        msq_packed_reads& v = d->view;
        v.n_reads = n;
        v.n_groups = tot;
        v.ref_start = (const int32_t*)d->ptrs[0];
        v.ref_end = (const int32_t*)d->ptrs[1];
        v.group_off = (const uint64_t*)d->ptrs[2];
        v.bop = (const uint32_t*)d->ptrs[3];
        v.qual = (const uint8_t*)d->ptrs[4];
        v.delqv = (const uint8_t*)d->ptrs[5];
        v.subqv = (const uint8_t*)d->ptrs[6];
        v.insqv = (const uint8_t*)d->ptrs[7];
        v.flags = d_reads->flags & ~MSQ_READS_ALIGNED;  // a slab's pieces are plain groups to every kernel
        v.align_origin = 0;
        v.gcol = nullptr;
        if (msq_packed_dev_index(ctx, d) != MSQ_OK) return fail(MSQ_ERR_HIP, "group column index", hipGetLastError());  // a sub-read is a read
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(MSQ_ERR_HIP, "synchronise", e);
    for (void* q : {d_first, d_cnt, d_off, d_rs, d_re, d_tmp})
        if (q) (void)hipFree(q);
    *out = p;
    return MSQ_OK;
}

extern "C" int msq_pileup_partition(msq_ctx* ctx, const msq_partition* p, const msq_qv_thresholds* thr, uint32_t* d_nuc, uint32_t* d_codon)
{
    if (!ctx || !p || !thr) return msq_fail(MSQ_ERR_ARG, "msq_pileup_partition: NULL argument");
    if (p->W == 0 || p->sets.empty()) return MSQ_OK;
    if (!d_nuc || !d_codon) return msq_fail(MSQ_ERR_ARG, "msq_pileup_partition: NULL count matrix");
    for (int k = 0; k < 4; ++k) {
        const int32_t t = (&thr->qual)[k];
        if (t < -1 || t > 255) return msq_fail(MSQ_ERR_ARG, "msq_pileup_partition: QV threshold %d out of range", t);
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    {  // all slabs in one launch where that applies (the per-launch costs are paid once)
        std::vector<const msq_packed_reads*> views;
        for (const msq_packed_dev* d : p->sets) views.push_back(&d->view);
        bool handled = false;
        int rc = msq_pileup_stream_slabs_launch(ctx, views.data(), (int)views.size(), *thr, p->win_begin0, p->W, p->slab_words, d_nuc, d_codon, &handled);
        if (rc != MSQ_OK || handled) return rc;
    }
    for (int s = 0; s < (int)p->sets.size(); ++s) {
        const msq_packed_reads& v = p->sets[s]->view;
        if (v.n_reads == 0) continue;
        bool handled = false;
        int rc = msq_pileup_stream_slab_launch(ctx, v, *thr, p->win_begin0, p->W, s, p->slab_words, d_nuc, d_codon, &handled);
        if (rc != MSQ_OK) return rc;
        if (!handled) return msq_fail(MSQ_ERR_ARG, "msq_pileup_partition: needs the streaming kernel (flags=%u)", v.flags);
    }
    return MSQ_OK;
}
