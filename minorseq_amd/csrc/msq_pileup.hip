// msq_pileup.hip -- window bounds + the SIMPLE pileup kernel (one wave per read, byte rows in LDS).
//
// The simple kernel is the obviously-correct device restatement of MSAByRow::AddRead + MSAByColumn +
// CodonsAt (src/MSA.cpp:55-92,126-136,144-219): it handles any window width that fits one LDS row and
// every op of the packed format.  The streaming bit-plane kernel (msq_pileup_stream.hip) is the fast
// path; msq_pileup() picks between them.
#include "msq_internal.h"
#include "msq_rowbuild.h"

#include <climits>

int msq_pileup_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr,
                             int32_t win_begin0, int32_t W, const int32_t* d_bounds, uint32_t* d_nuc, uint32_t* d_codon,
                             bool* handled);
int msq_pileup_rows_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                           uint32_t* d_nuc, uint32_t* d_codon, uint32_t* fixed_partials, size_t fixed_bytes, int32_t* d_step_bounds,
                           uint32_t* d_dev, bool* handled);
int msq_pileup_stream_step_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                                  int32_t* d_bounds, uint32_t* d_counts, bool* handled);

// ------------------------------------------------------------------------------------------ window bounds
// MSAByRow::UpdateBoundaries over all reads (src/MSA.cpp:138-142).  acc[0] = max over reads of ~key(start),
// acc[1] = max of key(end), with key(x) = x ^ 0x80000000, the order-preserving map of int32 onto uint32: both
// maxima start from 0.  acc lives in the context's small accumulators, which are all zero between kernels: the
// workgroup that draws the last ticket (acc[2]) reads the maxima back through atomics, writes {-begin0, end0}
// (endPos_ never drops below its initial 0) to `out` and clears acc -- no memset before, no kernel after.
// zero / zero_words: msq_pileup_begin's count matrices, cleared by the same launch (MSARow(size) rows start blank,
// MSAColumn counts start at 0); NULL / 0 for the plain bounds call.
__global__ __launch_bounds__(256) void window_bounds_kernel(const int32_t* __restrict__ ref_start, const int32_t* __restrict__ ref_end,
                                                            uint64_t n, uint32_t* __restrict__ acc, int32_t* __restrict__ out,
                                                            uint32_t* __restrict__ zero, uint64_t zero_words)
{
    int32_t lo = INT_MAX, hi = INT_MIN;
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, nthr = (uint64_t)gridDim.x * blockDim.x;
    if (zero_words) {  // stores first: they retire while the loads below are in flight
        const uint64_t z4 = ((uintptr_t)zero & 15u) ? 0 : zero_words / 4;
        uint4* zq = reinterpret_cast<uint4*>(zero);
        for (uint64_t i = tid; i < z4; i += nthr) zq[i] = make_uint4(0, 0, 0, 0);
        for (uint64_t i = z4 * 4 + tid; i < zero_words; i += nthr) zero[i] = 0;
    }
    const uint64_t n4 = (((uintptr_t)ref_start | (uintptr_t)ref_end) & 15u) ? 0 : n / 4;  // 16-byte loads when aligned
    const int4* s4 = reinterpret_cast<const int4*>(ref_start);
    const int4* e4 = reinterpret_cast<const int4*>(ref_end);
    uint64_t i = tid;
    for (; i + 3 * nthr < n4; i += 4 * nthr) {  // eight independent loads in flight per thread: the kernel is latency-bound
        int4 s[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[k] = s4[i + k * nthr];
            e[k] = e4[i + k * nthr];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo = min(min(lo, s[k].x), min(s[k].y, min(s[k].z, s[k].w)));
            hi = max(max(hi, e[k].x), max(e[k].y, max(e[k].z, e[k].w)));
        }
    }
    for (; i < n4; i += nthr) {
        const int4 s = s4[i], e = e4[i];
        lo = min(min(lo, s.x), min(s.y, min(s.z, s.w)));
        hi = max(max(hi, e.x), max(e.y, max(e.z, e.w)));
    }
    for (uint64_t i = n4 * 4 + tid; i < n; i += nthr) {
        lo = min(lo, ref_start[i]);
        hi = max(hi, ref_end[i]);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d));
        hi = max(hi, __shfl_xor(hi, d));
    }
    __shared__ int32_t s_lo[4], s_hi[4];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic pair per block: same-address atomics serialise
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            lo = min(lo, s_lo[k]);
            hi = max(hi, s_hi[k]);
        }
        atomicMax(&acc[0], ~((uint32_t)lo ^ 0x80000000u));
        atomicMax(&acc[1], (uint32_t)hi ^ 0x80000000u);
        __threadfence();
        if (atomicAdd(&acc[2], 1u) == gridDim.x - 1) {  // every other workgroup's maxima are in
            __threadfence();
            const uint32_t k_lo = atomicExch(&acc[0], 0u), k_hi = atomicExch(&acc[1], 0u);
            atomicExch(&acc[2], 0u);
            const int32_t begin0 = (int32_t)(~k_lo ^ 0x80000000u), end0 = (int32_t)(k_hi ^ 0x80000000u);
            out[0] = -begin0;
            out[1] = end0 > 0 ? end0 : 0;
        }
    }
}

static int launch_window_bounds(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* d_out, uint32_t* d_zero = nullptr,
                                uint64_t zero_words = 0)
{
    uint64_t blocks = (d_reads->n_reads / 4 + 255) / 256;
    if (zero_words) blocks = 128;
    blocks = blocks < 1 ? 1 : blocks > 128 ? 128 : blocks;  // few blocks: their same-address atomics serialise
    hipLaunchKernelGGL(window_bounds_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_reads->ref_start,
                       d_reads->ref_end, d_reads->n_reads, ctx->d_small, d_out, d_zero, zero_words);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

extern "C" int msq_window_bounds_device(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* d_bounds)
{
    if (!ctx || !d_reads || !d_bounds) return msq_fail(MSQ_ERR_ARG, "msq_window_bounds_device: NULL argument");
    if (d_reads->n_reads && (!d_reads->ref_start || !d_reads->ref_end)) return msq_fail(MSQ_ERR_ARG, "msq_window_bounds_device: NULL ref arrays");
    MSQ_HIP(hipSetDevice(ctx->device));
    if (d_reads->n_reads == 0) {  // MSAByRow defaults (include/pacbio/data/MSA.h:83-84): begin INT_MAX, end 0
        const int32_t none[2] = {-INT_MAX, 0};
        MSQ_HIP(hipMemcpyAsync(d_bounds, none, sizeof(none), hipMemcpyHostToDevice, ctx->stream));
        return MSQ_OK;
    }
    return launch_window_bounds(ctx, d_reads, d_bounds);
}

extern "C" int msq_pileup_begin(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* d_bounds, uint32_t* d_counts, size_t n_words)
{
    if (!ctx || !d_reads || !d_bounds || (!d_counts && n_words)) return msq_fail(MSQ_ERR_ARG, "msq_pileup_begin: NULL argument");
    if (d_reads->n_reads && (!d_reads->ref_start || !d_reads->ref_end)) return msq_fail(MSQ_ERR_ARG, "msq_pileup_begin: NULL ref arrays");
    MSQ_HIP(hipSetDevice(ctx->device));
    if (d_reads->n_reads == 0) {
        const int32_t none[2] = {-INT_MAX, 0};
        MSQ_HIP(hipMemcpyAsync(d_bounds, none, sizeof(none), hipMemcpyHostToDevice, ctx->stream));
        if (n_words) MSQ_HIP(hipMemsetAsync(d_counts, 0, n_words * sizeof(uint32_t), ctx->stream));
        return MSQ_OK;
    }
    return launch_window_bounds(ctx, d_reads, d_bounds, d_counts, n_words);
}

extern "C" int msq_window_bounds(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* begin0, int32_t* end0)
{
    if (!ctx || !d_reads || !begin0 || !end0) return msq_fail(MSQ_ERR_ARG, "msq_window_bounds: NULL argument");
    MSQ_HIP(hipSetDevice(ctx->device));
    *begin0 = INT_MAX;  // MSAByRow defaults (include/pacbio/data/MSA.h:83-84)
    *end0 = 0;
    if (d_reads->n_reads == 0) return MSQ_OK;
    if (!d_reads->ref_start || !d_reads->ref_end) return msq_fail(MSQ_ERR_ARG, "msq_window_bounds: NULL ref arrays");
    int32_t* d_res = reinterpret_cast<int32_t*>(ctx->d_small + 4);
    int rc = launch_window_bounds(ctx, d_reads, d_res);
    if (rc != MSQ_OK) return rc;
    int32_t res[2];
    MSQ_HIP(hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    *begin0 = -res[0];
    *end0 = res[1];  // UpdateBoundaries only ever raises endPos_ from 0 (src/MSA.cpp:138-142)
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------ simple kernel
__global__ __launch_bounds__(256) void pileup_simple_kernel(
    const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off, const uint4* __restrict__ bop,
    QvPlanes qv, uint64_t n_reads, int32_t win_begin0, int32_t W, int32_t row_bytes, int32_t waves_per_wg,
    uint32_t* __restrict__ nuc, uint32_t* __restrict__ codon, uint32_t* __restrict__ err)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* row = lds + (size_t)wave * row_bytes;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint64_t iters = (n_reads + total_waves - 1) / total_waves;
    uint32_t my_err = 0;

    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t slot = it * total_waves + (uint64_t)blockIdx.x * waves_per_wg + wave;
        const bool active = slot < n_reads;
        const uint64_t r = slot;
        for (int c = lane * 4; c < row_bytes; c += 256) *reinterpret_cast<uint32_t*>(row + c) = 0;  // MSARow(size): ' '
        __syncthreads();
        if (active) my_err |= msq_build_row(row, lane, ref_start, group_off, bop, qv, r, win_begin0, W);
        __syncthreads();
        if (active) {
            for (int c = lane; c < W; c += 64) {
                const int s0 = row[c];
                if (s0 == SYM_BLANK) continue;                  // MSA.cpp:81-83
                atomicAdd(&nuc[(size_t)c * 6 + (s0 - 1)], 1u);  // MSAColumn::IncCounts, MSA.inl:66
                // CodingCodonAt (MSA.cpp:197-219) + CodonAt's i > 0 rule (:193)
                if (c >= 1 && c + 2 < W) {
                    const int s1 = row[c + 1], s2 = row[c + 2];
                    if (s0 <= SYM_T && s1 >= SYM_A && s1 <= SYM_T && s2 >= SYM_A && s2 <= SYM_T)
                        atomicAdd(&codon[(size_t)c * 64 + ((s0 - 1) << 4 | (s1 - 1) << 2 | (s2 - 1))], 1u);
                }
            }
        }
        __syncthreads();
    }
    if (my_err) atomicOr(err, my_err);
}

static int launch_simple(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                         int32_t W, uint32_t* d_nuc, uint32_t* d_codon, uint64_t n_reads)
{
    const int32_t row_bytes = ((W + 2 + 15) / 16) * 16;
    int waves = 150 * 1024 / row_bytes;
    if (waves < 1) return msq_fail(MSQ_ERR_ARG, "msq_pileup: window of %d columns exceeds the simple kernel's LDS row", W);
    if (waves > 4) waves = 4;
    const size_t lds = (size_t)row_bytes * waves;
    if (lds > 48 * 1024)
        MSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pileup_simple_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint64_t blocks = (n_reads + waves - 1) / waves;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    QvPlanes q;
    q.plane[0] = R.qual; q.plane[1] = R.delqv; q.plane[2] = R.subqv; q.plane[3] = R.insqv;
    q.thr[0] = thr.qual; q.thr[1] = thr.del; q.thr[2] = thr.sub; q.thr[3] = thr.ins;
    msq_timing_mark_begin(ctx);
    hipLaunchKernelGGL(pileup_simple_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, ctx->stream, R.ref_start,
                       R.group_off, reinterpret_cast<const uint4*>(R.bop), q, n_reads, win_begin0, W, row_bytes,
                       waves, d_nuc, d_codon, ctx->d_err);
    msq_timing_mark_end(ctx);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------ per-read rows
// MSARow::Bases of reads [first, first + n) as the reference keeps them (MSARow(size): W chars, ' ' where the read
// has nothing; AddRead, src/MSA.cpp:144-186): one wave per read builds the row in LDS exactly as the simple pileup
// does and writes it out as characters (a wave stores 64 consecutive chars at a time; not a hot path).
__global__ __launch_bounds__(256) void rows_decode_kernel(
    const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off, const uint4* __restrict__ bop,
    QvPlanes qv, uint64_t first, uint64_t n, int32_t win_begin0, int32_t W, int32_t row_bytes, int32_t waves_per_wg,
    char* __restrict__ rows, uint32_t* __restrict__ err)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* row = lds + (size_t)wave * row_bytes;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint64_t iters = (n + total_waves - 1) / total_waves;
    uint32_t my_err = 0;
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t slot = it * total_waves + (uint64_t)blockIdx.x * waves_per_wg + wave;
        const bool active = slot < n;
        for (int c = lane * 4; c < row_bytes; c += 256) *reinterpret_cast<uint32_t*>(row + c) = 0;
        __syncthreads();
        if (active) my_err |= msq_build_row(row, lane, ref_start, group_off, bop, qv, first + slot, win_begin0, W);
        __syncthreads();
        if (active) {
            char* out = rows + slot * (uint64_t)W;
            for (int c = lane; c < W; c += 64) {
                // SYM_* -> the chars of MSARow::Bases; a packed symbol has no other value
                out[c] = (char)((0x004e2d5447434120ull >> (8 * row[c])) & 0xff);  // the bytes of " ACGT-N", ' ' lowest
            }
        }
        __syncthreads();
    }
    if (my_err) atomicOr(err, my_err);
}

extern "C" int msq_rows_decode(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                               int32_t W, uint64_t first_read, uint64_t n_rows, char* d_rows)
{
    if (!ctx || !d_reads || !thr) return msq_fail(MSQ_ERR_ARG, "msq_rows_decode: NULL argument");
    if (W < 0) return msq_fail(MSQ_ERR_ARG, "msq_rows_decode: negative window width %d", W);
    if (first_read > d_reads->n_reads || n_rows > d_reads->n_reads - first_read)
        return msq_fail(MSQ_ERR_ARG, "msq_rows_decode: rows [%llu, +%llu) of %llu reads", (unsigned long long)first_read,
                        (unsigned long long)n_rows, (unsigned long long)d_reads->n_reads);
    if (n_rows == 0 || W == 0) return MSQ_OK;
    if (!d_rows) return msq_fail(MSQ_ERR_ARG, "msq_rows_decode: NULL row buffer");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_rows_decode")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    const int32_t row_bytes = ((W + 2 + 15) / 16) * 16;
    int waves = 150 * 1024 / row_bytes;
    if (waves < 1) return msq_fail(MSQ_ERR_ARG, "msq_rows_decode: window of %d columns exceeds one LDS row", W);
    if (waves > 4) waves = 4;
    const size_t lds = (size_t)row_bytes * waves;
    if (lds > 48 * 1024)
        MSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rows_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    uint64_t blocks = (n_rows + waves - 1) / waves;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    QvPlanes q;
    q.plane[0] = d_reads->qual; q.plane[1] = d_reads->delqv; q.plane[2] = d_reads->subqv; q.plane[3] = d_reads->insqv;
    q.thr[0] = thr->qual; q.thr[1] = thr->del; q.thr[2] = thr->sub; q.thr[3] = thr->ins;
    hipLaunchKernelGGL(rows_decode_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, reinterpret_cast<const uint4*>(d_reads->bop), q, first_read, n_rows, win_begin0, W,
                       row_bytes, waves, d_rows, ctx->d_err);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

extern "C" int msq_pileup(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr,
                          int32_t win_begin0, int32_t W, uint32_t* d_nuc, uint32_t* d_codon, int mode)
{
    if (!ctx || !d_reads || !thr) return msq_fail(MSQ_ERR_ARG, "msq_pileup: NULL argument");
    if (W < 0) return msq_fail(MSQ_ERR_ARG, "msq_pileup: negative window width %d", W);
    if (d_reads->n_reads == 0 || W == 0) return MSQ_OK;
    if (!d_nuc || !d_codon) return msq_fail(MSQ_ERR_ARG, "msq_pileup: NULL count matrix");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_pileup")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    if (mode != MSQ_PILEUP_SIMPLE) {
        bool handled = false;
        // MSQ_PILEUP_ROWS: only the kernel over column-aligned sets (tests ask for it by name; AUTO and STREAM try it first)
        int rc = msq_pileup_rows_launch(ctx, *d_reads, *thr, win_begin0, W, d_nuc, d_codon, nullptr, 0, nullptr, nullptr, &handled);
        if (rc != MSQ_OK) return rc;
        if (handled) return MSQ_OK;
        if (mode == MSQ_PILEUP_ROWS)
            return msq_fail(MSQ_ERR_ARG, "msq_pileup: the aligned-set kernel is not applicable (flags=%u, W=%d, origin %d)", d_reads->flags, W,
                            d_reads->align_origin);
        rc = msq_pileup_stream_launch(ctx, *d_reads, *thr, win_begin0, W, nullptr, d_nuc, d_codon, &handled);
        if (rc != MSQ_OK) return rc;
        if (handled) return MSQ_OK;
        if (mode == MSQ_PILEUP_STREAM)
            return msq_fail(MSQ_ERR_ARG, "msq_pileup: streaming kernel not applicable (W=%d, flags=%u)", W, d_reads->flags);
    }
    return launch_simple(ctx, *d_reads, *thr, win_begin0, W, d_nuc, d_codon, d_reads->n_reads);
}

// msq_pileup_begin + msq_pileup(AUTO) as ONE call: the reads' bounds into d_bounds, the matrices d_counts = [W][6]
// nucleotide counts then [W][64] codon counts OVERWRITTEN with the pileup inside [win_begin0, win_begin0 + W).
// When the streaming kernel can take the whole call as one launch, it is one kernel + its fold (the bounds are a
// side product of the kernel, the fold stores instead of adding: no bounds / clearing launch at all).
extern "C" int msq_pileup_step(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                               int32_t W, int32_t* d_bounds, uint32_t* d_counts)
{
    if (!ctx || !d_reads || !thr || !d_bounds) return msq_fail(MSQ_ERR_ARG, "msq_pileup_step: NULL argument");
    if (W < 0) return msq_fail(MSQ_ERR_ARG, "msq_pileup_step: negative window width %d", W);
    if (W > 0 && !d_counts) return msq_fail(MSQ_ERR_ARG, "msq_pileup_step: NULL count matrices");
    if (d_reads->n_reads && (!d_reads->ref_start || !d_reads->ref_end)) return msq_fail(MSQ_ERR_ARG, "msq_pileup_step: NULL ref arrays");
    const size_t n_words = (size_t)W * 70;
    if (d_reads->n_reads == 0 || W == 0) return msq_pileup_begin(ctx, d_reads, d_bounds, d_counts, n_words);
    if (int rc = msq_validate_reads(d_reads, thr, "msq_pileup_step")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    bool handled = false;
    int rc = msq_pileup_stream_step_launch(ctx, *d_reads, *thr, win_begin0, W, d_bounds, d_counts, &handled);
    if (rc != MSQ_OK || handled) return rc;
    rc = msq_pileup_begin(ctx, d_reads, d_bounds, d_counts, n_words);
    if (rc != MSQ_OK) return rc;
    return msq_pileup(ctx, d_reads, thr, win_begin0, W, d_counts, d_counts + (size_t)W * 6, MSQ_PILEUP_AUTO);
}

extern "C" int msq_pileup_device_window(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr,
                                        const int32_t* d_bounds, int32_t begin0, int32_t W_max, uint32_t* d_nuc,
                                        uint32_t* d_codon)
{
    if (!ctx || !d_reads || !thr || !d_bounds) return msq_fail(MSQ_ERR_ARG, "msq_pileup_device_window: NULL argument");
    if (W_max < 0) return msq_fail(MSQ_ERR_ARG, "msq_pileup_device_window: negative capacity %d", W_max);
    if (d_reads->n_reads == 0 || W_max == 0) return MSQ_OK;
    if (!d_nuc || !d_codon) return msq_fail(MSQ_ERR_ARG, "msq_pileup_device_window: NULL count matrix");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_pileup_device_window")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    bool handled = false;
    int rc = msq_pileup_stream_launch(ctx, *d_reads, *thr, begin0, W_max, d_bounds, d_nuc, d_codon, &handled);
    if (rc != MSQ_OK) return rc;
    if (!handled) return msq_fail(MSQ_ERR_ARG, "msq_pileup_device_window: needs the streaming kernel (flags=%u)", d_reads->flags);
    return MSQ_OK;
}

extern "C" int msq_pileup_check(msq_ctx* ctx)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_pileup_check: ctx is NULL");
    uint32_t e = 0;
    MSQ_HIP(hipMemcpyAsync(&e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (e & (DEVERR_BASE | DEVERR_WINDOW | DEVERR_EXOTIC)) {
        const uint32_t keep = e & ~(DEVERR_BASE | DEVERR_WINDOW | DEVERR_EXOTIC);
        MSQ_HIP(hipMemcpyAsync(ctx->d_err, &keep, sizeof(keep), hipMemcpyHostToDevice, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (e & DEVERR_EXOTIC)
            return msq_fail(MSQ_ERR_ARG, "packed reads hold op-1 codes 2/3 but flags lack MSQ_READS_HAS_EXOTIC");
        if (e & DEVERR_WINDOW) return msq_fail(MSQ_ERR_WINDOW, "a read reaches outside the MSA window");
        return msq_fail(MSQ_ERR_BASE, "Unexpected base");
    }
    return MSQ_OK;
}
