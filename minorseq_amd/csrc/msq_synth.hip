// msq_synth.hip -- synthetic CCS amplicon generator: host (unpacked) and device (packed) forms of one model.
#include "msq_internal.h"
#include "msq_synth_model.h"

static int check_params(const msq_synth_params* p)
{
    if (!p) return msq_fail(MSQ_ERR_ARG, "msq_synth: params is NULL");
    if (p->W <= 0 || p->n_minors < 0 || p->max_jitter < 0) return msq_fail(MSQ_ERR_ARG, "msq_synth: bad params");
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------------ host
extern "C" int msq_synth_host_lengths(const msq_synth_params* p, uint64_t first_read, uint64_t n_reads, int64_t* n_bases)
{
    int rc = check_params(p);
    if (rc != MSQ_OK) return rc;
    if (n_reads && !n_bases) return msq_fail(MSQ_ERR_ARG, "msq_synth_host_lengths: NULL output");
    for (uint64_t k = 0; k < n_reads; ++k) {
        const uint64_t read = first_read + k;
        const SynRead hdr = syn_read_header(*p, read);
        int64_t n = 0;
        for (int c = hdr.c_begin; c < hdr.c_end; ++c) n += 1 + syn_column_has_ins(*p, hdr, read, c);
        n_bases[k] = n;
    }
    return MSQ_OK;
}

extern "C" int msq_synth_host_fill(const msq_synth_params* p, const uint8_t* clone_codes, uint64_t first_read,
                                   uint64_t n_reads, const int64_t* base_off, int32_t* ref_start, int32_t* ref_end,
                                   char* cigar, char* nuc, uint8_t* qual, uint8_t* subqv, uint8_t* delqv, uint8_t* insqv)
{
    int rc = check_params(p);
    if (rc != MSQ_OK) return rc;
    if (!clone_codes || !base_off || !ref_start || !ref_end || !cigar || !nuc)
        return msq_fail(MSQ_ERR_ARG, "msq_synth_host_fill: NULL argument");
    static const char kNuc[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t k = 0; k < n_reads; ++k) {
        const uint64_t read = first_read + k;
        const SynRead hdr = syn_read_header(*p, read);
        ref_start[k] = p->win_begin0 + hdr.c_begin;
        ref_end[k] = p->win_begin0 + hdr.c_end;
        int64_t i = base_off[k];
        auto put = [&](char cg, char nt, const SynQv& q) {
            cigar[i] = cg;
            nuc[i] = nt;
            if (qual) qual[i] = q.qual;
            if (subqv) subqv[i] = q.sub;
            if (delqv) delqv[i] = q.del;
            if (insqv) insqv[i] = q.ins;
            ++i;
        };
        for (int c = hdr.c_begin; c < hdr.c_end; ++c) {
            const SynColumn col = syn_column(*p, clone_codes, hdr, read, c);
            if (col.has_ins) put('I', kNuc[col.ins_code], syn_qvs(*p, read, c, true));
            if (col.kind == 2) {
                put('D', '-', SynQv{0, 0, 0, 0});
            } else {
                put(col.kind == 1 ? 'X' : '=', kNuc[col.code], syn_qvs(*p, read, c, false));
            }
        }
        if (i != base_off[k + 1]) return msq_fail(MSQ_ERR_ARG, "msq_synth_host_fill: base_off does not match the lengths");
    }
    return MSQ_OK;
}

// ---------------------------------------------------------------------------------------------- device
__global__ __launch_bounds__(256) void synth_group_counts_kernel(msq_synth_params p, uint64_t first_read, uint64_t n_reads,
                                                                 uint32_t* __restrict__ n_groups)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n_reads) return;
    const uint64_t read = first_read + k;
    const SynRead hdr = syn_read_header(p, read);
    uint32_t n = 0;
    for (int c = hdr.c_begin; c < hdr.c_end; ++c) n += 1 + syn_column_has_ins(p, hdr, read, c);
    n_groups[k] = (n + 31) / 32;
}

// One group in registers, written in the normalized layout (MSQ_READS_NORMALIZED, include/msq_abi.h): consuming
// bases take the slots from 0 upwards, the others from 31 downwards, op-3 padding in between.
struct GroupWriter {
    uint32_t n0, n1, o0, o1;
    uint32_t q[4][8];
    int up, down;
    uint64_t g;
    __device__ void reset()
    {
        n0 = n1 = o0 = o1 = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) q[a][b] = 0;
        up = 0;
        down = 31;
    }
    __device__ void flush(uint32_t* bop, uint8_t* const* planes)
    {
        if (up == 0 && down == 31) return;
        if (up <= down) {  // padding: op 3
            const uint32_t pad = (0xffffffffu >> (31 - down)) & (0xffffffffu << up);
            o0 |= pad;
            o1 |= pad;
        }
        reinterpret_cast<uint4*>(bop)[g] = make_uint4(n0, n1, o0, o1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (planes[a]) {
                uint4* dst = reinterpret_cast<uint4*>(planes[a] + g * 32);
                dst[0] = make_uint4(q[a][0], q[a][1], q[a][2], q[a][3]);
                dst[1] = make_uint4(q[a][4], q[a][5], q[a][6], q[a][7]);
            }
        ++g;
        reset();
    }
    __device__ void push(uint32_t op, uint32_t code, const SynQv& v, uint32_t* bop, uint8_t* const* planes)
    {
        const int k = op < 2u ? up++ : down--;
        n0 |= (code & 1u) << k;
        n1 |= ((code >> 1) & 1u) << k;
        o0 |= (op & 1u) << k;
        o1 |= ((op >> 1) & 1u) << k;
        const int w = MSQ_QV_BYTE(k) >> 2, sh = (MSQ_QV_BYTE(k) & 3) * 8;
        // static indexing keeps q in registers
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b == w) {
                q[0][b] |= (uint32_t)v.qual << sh;
                q[1][b] |= (uint32_t)v.del << sh;
                q[2][b] |= (uint32_t)v.sub << sh;
                q[3][b] |= (uint32_t)v.ins << sh;
            }
        if (up > down) flush(bop, planes);
    }
};

__global__ __launch_bounds__(256) void synth_fill_kernel(msq_synth_params p, const uint8_t* __restrict__ clone_codes,
                                                         uint64_t first_read, uint64_t n_reads,
                                                         const uint64_t* __restrict__ group_off, int32_t* __restrict__ ref_start,
                                                         int32_t* __restrict__ ref_end, uint32_t* __restrict__ bop,
                                                         uint8_t* qual, uint8_t* delqv, uint8_t* subqv, uint8_t* insqv)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n_reads) return;
    const uint64_t read = first_read + k;
    const SynRead hdr = syn_read_header(p, read);
    ref_start[k] = p.win_begin0 + hdr.c_begin;
    ref_end[k] = p.win_begin0 + hdr.c_end;
    uint8_t* const planes[4] = {qual, delqv, subqv, insqv};
    GroupWriter w;
    w.reset();
    w.g = group_off[k];
    for (int c = hdr.c_begin; c < hdr.c_end; ++c) {
        const SynColumn col = syn_column(p, clone_codes, hdr, read, c);
        if (col.has_ins) w.push(2u, (uint32_t)col.ins_code, syn_qvs(p, read, c, true), bop, planes);
        if (col.kind == 2)
            w.push(1u, 0u, SynQv{0, 0, 0, 0}, bop, planes);
        else
            w.push(0u, (uint32_t)col.code, syn_qvs(p, read, c, false), bop, planes);
    }
    w.flush(bop, planes);
}

extern "C" int msq_synth_dev_group_counts(msq_ctx* ctx, const msq_synth_params* p, uint64_t first_read, uint64_t n_reads,
                                          uint32_t* d_n_groups)
{
    int rc = check_params(p);
    if (rc != MSQ_OK) return rc;
    if (!ctx || (n_reads && !d_n_groups)) return msq_fail(MSQ_ERR_ARG, "msq_synth_dev_group_counts: NULL argument");
    if (n_reads == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(synth_group_counts_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, ctx->stream, *p,
                       first_read, n_reads, d_n_groups);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

extern "C" int msq_synth_dev_fill(msq_ctx* ctx, const msq_synth_params* p, const uint8_t* d_clone_codes,
                                  uint64_t first_read, const msq_packed_reads* d)
{
    int rc = check_params(p);
    if (rc != MSQ_OK) return rc;
    if (!ctx || !d || !d_clone_codes) return msq_fail(MSQ_ERR_ARG, "msq_synth_dev_fill: NULL argument");
    if (d->n_reads == 0) return MSQ_OK;
    if (!d->ref_start || !d->ref_end || !d->group_off || !d->bop || !d->qual)
        return msq_fail(MSQ_ERR_ARG, "msq_synth_dev_fill: NULL read array");
    if (p->rich_qv && (!d->delqv || !d->subqv || !d->insqv))
        return msq_fail(MSQ_ERR_ARG, "msq_synth_dev_fill: rich_qv needs the del/sub/ins planes");
    MSQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)((d->n_reads + 255) / 256)), dim3(256), 0, ctx->stream, *p,
                       d_clone_codes, first_read, d->n_reads, d->group_off, const_cast<int32_t*>(d->ref_start),
                       const_cast<int32_t*>(d->ref_end), const_cast<uint32_t*>(d->bop), const_cast<uint8_t*>(d->qual),
                       p->rich_qv ? const_cast<uint8_t*>(d->delqv) : nullptr,
                       p->rich_qv ? const_cast<uint8_t*>(d->subqv) : nullptr,
                       p->rich_qv ? const_cast<uint8_t*>(d->insqv) : nullptr);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}
