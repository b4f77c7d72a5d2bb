// msq_bake.hip -- QV thresholds applied once, at pack time (msq_reads_bake_qv).
//
// MSAByRow::AddRead turns a match base whose QVs miss a threshold into the row char 'N' (src/MSA.cpp:163-166 over
// ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87).  The thresholds of a run never change after the reads are
// loaded (JulietSettings / ErrorEstimates fix them), so that decision can be made once: the kernel below rewrites
// every '=' / 'X' base that fails as a literal 'N' (op 1, code 1: the very row char AddRead produces) and drops the
// QV planes.  Every later pass over the set -- pileup, phasing gather, insertion scan -- then streams 0.5 bytes per
// base instead of 1.5 .. 4.5, and gives the results of the original set under those thresholds, bit for bit.
#include "msq_internal.h"

#include <new>

namespace {

struct BakeArgs {
    const uint4* bop;
    const uint4* qv[4];  // planes with a threshold, compacted
    int32_t thr[4];
    int np;
    uint64_t n_groups;
    uint4* out;
};

// bit k = (QV byte of base k >= thr); the plane holds the group transposed (MSQ_QV_BYTE): dword i = bases i, 8+i, ..
__device__ __forceinline__ uint32_t pass_bits(const uint4* plane, uint64_t g, uint32_t thr)
{
    const uint4 a = plane[2 * g], b = plane[2 * g + 1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t pass = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((w[i] >> (8 * j)) & 0xffu) >= thr) pass |= 1u << (8 * j + i);
    return pass;
}

__global__ __launch_bounds__(256) void bake_qv_kernel(BakeArgs a)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_groups) return;
    uint4 p = a.bop[g];
    uint32_t pass = 0xffffffffu;
    for (int k = 0; k < a.np; ++k) pass &= pass_bits(a.qv[k], g, (uint32_t)a.thr[k]);
    // what a missed threshold turns into 'N': op 0, and the exotic match bases kept under op 1 (code 2 a non-ACGT
    // letter, code 3 a literal '-': msq_rowbuild.h); 'D' gaps (code 0) and literal Ns (code 1) ignore the QVs
    const uint32_t masked = ~p.w & (~p.z | p.y) & ~pass;
    p.x |= masked;                                // code 1 ..
    p.y &= ~masked;
    p.z |= masked;                                // .. under op 1: literal 'N'
    a.out[g] = p;
}

}  // namespace

extern "C" int msq_reads_bake_qv(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr,
                                 msq_packed_dev** out)
{
    if (!ctx || !d_reads || !thr || !out) return msq_fail(MSQ_ERR_ARG, "msq_reads_bake_qv: NULL argument");
    *out = nullptr;
    const uint8_t* planes[4] = {d_reads->qual, d_reads->delqv, d_reads->subqv, d_reads->insqv};
    const int32_t thrs[4] = {thr->qual, thr->del, thr->sub, thr->ins};
    BakeArgs a{};
    for (int k = 0; k < 4; ++k) {
        if (thrs[k] < -1 || thrs[k] > 255) return msq_fail(MSQ_ERR_ARG, "msq_reads_bake_qv: QV threshold %d out of range", thrs[k]);
        if (thrs[k] >= 0 && planes[k]) {
            a.qv[a.np] = reinterpret_cast<const uint4*>(planes[k]);
            a.thr[a.np] = thrs[k];
            ++a.np;
        }
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
    if (!d) return msq_fail(MSQ_ERR_NOMEM, "msq_reads_bake_qv: out of memory");
    const void* src[4] = {d_reads->ref_start, d_reads->ref_end, d_reads->group_off, nullptr};
    const size_t bytes[4] = {d_reads->n_reads * 4, d_reads->n_reads * 4, (d_reads->n_reads + 1) * 8, d_reads->n_groups * 16};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        e = hipMalloc(&d->ptrs[k], bytes[k] ? bytes[k] : 16);
        if (e == hipSuccess && src[k] && bytes[k])
            e = hipMemcpyAsync(d->ptrs[k], src[k], bytes[k], hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess && d_reads->n_groups) {
        a.bop = reinterpret_cast<const uint4*>(d_reads->bop);
        a.n_groups = d_reads->n_groups;
        a.out = static_cast<uint4*>(d->ptrs[3]);
        hipLaunchKernelGGL(bake_qv_kernel, dim3((unsigned)((a.n_groups + 255) / 256)), dim3(256), 0, ctx->stream, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        for (int j = 0; j < 4; ++j)
            if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
        delete d;
        return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "msq_reads_bake_qv: %s", hipGetErrorString(e));
    }
    d->view = *d_reads;
    d->view.ref_start = (const int32_t*)d->ptrs[0];
    d->view.ref_end = (const int32_t*)d->ptrs[1];
    d->view.group_off = (const uint64_t*)d->ptrs[2];
    d->view.bop = (const uint32_t*)d->ptrs[3];
    d->view.qual = d->view.delqv = d->view.subqv = d->view.insqv = nullptr;
    d->view.gcol = nullptr;  // the source's index belongs to the source: this set gets its own (same groups, same columns)
    d->view.guide = nullptr;  // ... and so does its guide; the groups keep their layout (an aligned source stays aligned)
    d->view.guide_words = 0;
    d->view.hap_count = 0;
    d->view.hap_guides = nullptr;
    d->view.hap_perm = nullptr;
    for (uint32_t& x : d->view.hap_reads) x = 0;
    if (a.np) d->view.flags |= MSQ_READS_QV_BAKED;
    if (int rc = msq_packed_dev_index(ctx, d)) {
        (void)hipStreamSynchronize(ctx->stream);
        for (int j = 0; j < 12; ++j)
            if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
        delete d;
        return rc;
    }
    *out = d;
    return MSQ_OK;
}
