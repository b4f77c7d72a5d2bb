// msq_phase.hip -- the per-read gather of AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:149-160):
// for every read, MSARow::CodonAt at each variant position.  The fast path is the streaming pileup kernel in its
// PHASE mode (msq_pileup_stream.hip); reads with exotic ops take the kernel below: one wave per read builds the
// read's row in LDS (msq_rowbuild.h) and reads the requested cells back.  Clustering the keys is the caller's.
#include "msq_internal.h"
#include "msq_rowbuild.h"

int msq_phase_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                            int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys, bool* handled);

namespace {

__global__ __launch_bounds__(256) void phase_codons_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                           const uint4* __restrict__ bop, QvPlanes qv, uint64_t n_reads,
                                                           int32_t win_begin0, int32_t W, int32_t row_bytes, int32_t waves_per_wg,
                                                           int32_t n_pos, const int32_t* __restrict__ pos, uint16_t* __restrict__ keys,
                                                           uint32_t* __restrict__ err)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* row = lds + (size_t)wave * row_bytes;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint64_t iters = (n_reads + total_waves - 1) / total_waves;
    uint32_t my_err = 0;
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t r = it * total_waves + (uint64_t)blockIdx.x * waves_per_wg + wave;
        const bool active = r < n_reads;
        for (int c = lane * 4; c < row_bytes; c += 256) *reinterpret_cast<uint32_t*>(row + c) = 0;  // MSARow(size): ' '
        __syncthreads();
        if (active) my_err |= msq_build_row(row, lane, ref_start, group_off, bop, qv, r, win_begin0, W);
        __syncthreads();
        if (active) {
            for (int k = lane; k < n_pos; k += 64) {
                const int p = pos[k];
                uint32_t key = 0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {  // MSARow::CodonAt: indices with i > 0 && i < length only (MSA.cpp:188-195)
                    const int i = p + j;
                    const uint32_t sym = (i > 0 && i < W) ? row[i] : 7u;
                    key |= sym << (3 * j);
                }
                keys[r * (uint64_t)n_pos + k] = (uint16_t)key;
            }
        }
        __syncthreads();
    }
    // a row the pileup accepted cannot fail here; keep the error word honest anyway
    if (my_err) atomicOr(err, my_err);
}

}  // namespace

extern "C" int msq_phase_codons(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                                int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys)
{
    if (!ctx || !d_reads || !thr) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: NULL argument");
    if (W < 0 || n_pos < 0) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: negative size");
    if (d_reads->n_reads == 0 || n_pos == 0) return MSQ_OK;
    if (!pos || !d_keys) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: NULL positions / keys");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_phase_codons")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    {  // fast path: phase 1 of the streaming pileup kernel + a gather from its LDS rows
        bool handled = false;
        int rc = msq_phase_stream_launch(ctx, *d_reads, *thr, win_begin0, W, n_pos, pos, d_keys, &handled);
        if (rc != MSQ_OK) return rc;
        if (handled) {
            MSQ_HIP(hipStreamSynchronize(ctx->stream));
            return MSQ_OK;
        }
    }
    // reads with exotic ops: one wave per read, byte rows (the simple kernel's row builder)
    const int32_t row_bytes = ((W + 2 + 15) / 16) * 16;
    int waves = 150 * 1024 / (row_bytes ? row_bytes : 16);
    if (waves < 1) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: window of %d columns exceeds one LDS row", W);
    if (waves > 4) waves = 4;
    const size_t lds = (size_t)row_bytes * waves;
    if (lds > 48 * 1024)
        MSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_codons_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    void* s = nullptr;
    int rc = msq_scratch(ctx, (size_t)n_pos * sizeof(int32_t), &s);
    if (rc != MSQ_OK) return rc;
    MSQ_HIP(hipMemcpyAsync(s, pos, (size_t)n_pos * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    uint64_t blocks = (d_reads->n_reads + waves - 1) / waves;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    QvPlanes q;
    q.plane[0] = d_reads->qual; q.plane[1] = d_reads->delqv; q.plane[2] = d_reads->subqv; q.plane[3] = d_reads->insqv;
    q.thr[0] = thr->qual; q.thr[1] = thr->del; q.thr[2] = thr->sub; q.thr[3] = thr->ins;
    hipLaunchKernelGGL(phase_codons_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, reinterpret_cast<const uint4*>(d_reads->bop), q, d_reads->n_reads, win_begin0, W,
                       row_bytes, waves, n_pos, (const int32_t*)s, d_keys, ctx->d_err);
    MSQ_HIP(hipGetLastError());
    // the positions live in the context's scratch: the next call that grows it must not race this kernel
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return MSQ_OK;
}
