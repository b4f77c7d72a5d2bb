// msq_pack.hip -- host packer (flat ArrayBase columns -> bit-plane groups) and upload.
// Format: include/msq_abi.h "packed read format".
#include "msq_internal.h"

#include <new>
#include <vector>

struct msq_packed_host {
    std::vector<int32_t> ref_start, ref_end;
    std::vector<uint64_t> group_off;
    std::vector<uint32_t> bop;
    std::vector<uint8_t> qual, delqv, subqv, insqv;
    msq_packed_reads view;
};

struct msq_packed_dev {
    void* ptrs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    msq_packed_reads view;
};

// Per-base translation of (ArrayBase::Cigar, ArrayBase::Nucleotide) as MSAByRow::AddRead switches on
// them (src/MSA.cpp:158-184).  Returns false on an unknown cigar char ("Unexpected cigar", :182).
static inline bool encode_base(char cigar, char nuc, unsigned* op, unsigned* code, bool* exotic)
{
    switch (cigar) {
        case '=':
        case 'X':
            switch (nuc) {
                case 'A': *op = 0; *code = 0; return true;
                case 'C': *op = 0; *code = 1; return true;
                case 'G': *op = 0; *code = 2; return true;
                case 'T': *op = 0; *code = 3; return true;
                case 'N': *op = 1; *code = 1; return true;
                case '-': *op = 1; *code = 3; *exotic = true; return true;
                default:  *op = 1; *code = 2; *exotic = true; return true;
            }
        case 'D':
            *op = 1; *code = 0; return true;
        case 'I':
            *op = 2;
            // inserted bases never reach a count matrix; keep the 2-bit code when it is one
            *code = nuc == 'C' ? 1 : nuc == 'G' ? 2 : nuc == 'T' ? 3 : 0;
            return true;
        case 'P':
        case 'S':
            *op = 3; *code = 0; return true;
        default:
            return false;
    }
}

extern "C" int msq_pack_reads(uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                              const int64_t* base_off, const char* cigar, const char* nuc, const uint8_t* qual,
                              const uint8_t* subqv, const uint8_t* delqv, const uint8_t* insqv,
                              msq_packed_host** out)
{
    if (!out) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads: out is NULL");
    *out = nullptr;
    if (n_reads && (!ref_start || !ref_end || !base_off)) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads: NULL array");
    msq_packed_host* p = new (std::nothrow) msq_packed_host;
    if (!p) return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads: out of memory");
    try {
        p->ref_start.assign(ref_start, ref_start + n_reads);
        p->ref_end.assign(ref_end, ref_end + n_reads);
        p->group_off.resize(n_reads + 1);
        uint64_t g = 0;
        for (uint64_t r = 0; r < n_reads; ++r) {
            const int64_t len = base_off[r + 1] - base_off[r];
            if (len < 0) {
                delete p;
                return msq_fail(MSQ_ERR_ARG, "msq_pack_reads: base_off not monotone at read %llu", (unsigned long long)r);
            }
            p->group_off[r] = g;
            g += (uint64_t)(len + 31) / 32;
        }
        p->group_off[n_reads] = g;
        p->bop.assign(g * 4, 0);
        const uint8_t* src[4] = {qual, delqv, subqv, insqv};
        std::vector<uint8_t>* dst[4] = {&p->qual, &p->delqv, &p->subqv, &p->insqv};
        for (int k = 0; k < 4; ++k)
            if (src[k]) dst[k]->assign(g * 32, 0);
        bool exotic = false;
        for (uint64_t r = 0; r < n_reads; ++r) {
            const int64_t b0 = base_off[r], len = base_off[r + 1] - b0;
            const uint64_t g0 = p->group_off[r], ng = p->group_off[r + 1] - g0;
            for (uint64_t gi = 0; gi < ng; ++gi) {
                // normalized layout (MSQ_READS_NORMALIZED): consuming bases from slot 0 up, the others from 31 down
                uint32_t n0 = 0, n1 = 0, o0 = 0, o1 = 0;
                int up = 0, down = 31, slot[32];
                for (int k = 0; k < 32; ++k) {
                    const int64_t i = (int64_t)gi * 32 + k;
                    slot[k] = -1;
                    if (i >= len) break;
                    unsigned op = 3, code = 0;
                    if (!encode_base(cigar[b0 + i], nuc[b0 + i], &op, &code, &exotic)) {
                        const int bad = (unsigned char)cigar[b0 + i];
                        delete p;
                        // std::runtime_error("Unexpected cigar " + std::to_string(b.Cigar)), src/MSA.cpp:182
                        return msq_fail(MSQ_ERR_CIGAR, "Unexpected cigar %d", bad);
                    }
                    const int s = op < 2 ? up++ : down--;
                    slot[k] = s;
                    n0 |= (uint32_t)(code & 1) << s;
                    n1 |= (uint32_t)((code >> 1) & 1) << s;
                    o0 |= (uint32_t)(op & 1) << s;
                    o1 |= (uint32_t)((op >> 1) & 1) << s;
                }
                for (int s = up; s <= down; ++s) {  // padding between the two runs: op 3, consumes nothing
                    o0 |= 1u << s;
                    o1 |= 1u << s;
                }
                uint32_t* w = &p->bop[(g0 + gi) * 4];
                w[0] = n0; w[1] = n1; w[2] = o0; w[3] = o1;
                for (int q = 0; q < 4; ++q)
                    if (src[q])
                        for (int k = 0; k < 32 && slot[k] >= 0; ++k)  // transposed inside each group, see MSQ_QV_BYTE
                            (*dst[q])[(g0 + gi) * 32 + MSQ_QV_BYTE(slot[k])] = src[q][b0 + (int64_t)gi * 32 + k];
            }
        }
        msq_packed_reads& v = p->view;
        v.n_reads = n_reads;
        v.n_groups = g;
        v.ref_start = p->ref_start.data();
        v.ref_end = p->ref_end.data();
        v.group_off = p->group_off.data();
        v.bop = p->bop.data();
        v.qual = qual ? p->qual.data() : nullptr;
        v.delqv = delqv ? p->delqv.data() : nullptr;
        v.subqv = subqv ? p->subqv.data() : nullptr;
        v.insqv = insqv ? p->insqv.data() : nullptr;
        v.flags = (exotic ? MSQ_READS_HAS_EXOTIC : 0u) | MSQ_READS_NORMALIZED;
        v.reserved = 0;
    } catch (const std::bad_alloc&) {
        delete p;
        return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads: out of memory");
    }
    *out = p;
    return MSQ_OK;
}

extern "C" const msq_packed_reads* msq_packed_host_view(const msq_packed_host* p) { return p ? &p->view : nullptr; }
extern "C" int msq_packed_host_free(msq_packed_host* p)
{
    delete p;
    return MSQ_OK;
}

extern "C" int msq_reads_upload(msq_ctx* ctx, const msq_packed_reads* h, msq_packed_dev** out)
{
    if (!ctx || !h || !out) return msq_fail(MSQ_ERR_ARG, "msq_reads_upload: NULL argument");
    *out = nullptr;
    MSQ_HIP(hipSetDevice(ctx->device));
    msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
    if (!d) return msq_fail(MSQ_ERR_NOMEM, "msq_reads_upload: out of memory");
    const void* src[8] = {h->ref_start, h->ref_end, h->group_off, h->bop, h->qual, h->delqv, h->subqv, h->insqv};
    const size_t bytes[8] = {h->n_reads * 4, h->n_reads * 4, (h->n_reads + 1) * 8, h->n_groups * 16,
                             h->n_groups * 32, h->n_groups * 32, h->n_groups * 32, h->n_groups * 32};
    for (int k = 0; k < 8; ++k) {
        if (!src[k]) continue;
        hipError_t e = hipMalloc(&d->ptrs[k], bytes[k] ? bytes[k] : 16);
        if (e == hipSuccess && bytes[k])
            e = hipMemcpyAsync(d->ptrs[k], src[k], bytes[k], hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            for (int j = 0; j < 8; ++j)
                if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
            delete d;
            return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "msq_reads_upload: %s",
                            hipGetErrorString(e));
        }
    }
    d->view = *h;
    d->view.ref_start = (const int32_t*)d->ptrs[0];
    d->view.ref_end = (const int32_t*)d->ptrs[1];
    d->view.group_off = (const uint64_t*)d->ptrs[2];
    d->view.bop = (const uint32_t*)d->ptrs[3];
    d->view.qual = (const uint8_t*)d->ptrs[4];
    d->view.delqv = (const uint8_t*)d->ptrs[5];
    d->view.subqv = (const uint8_t*)d->ptrs[6];
    d->view.insqv = (const uint8_t*)d->ptrs[7];
    *out = d;
    return MSQ_OK;
}
extern "C" const msq_packed_reads* msq_packed_dev_view(const msq_packed_dev* p) { return p ? &p->view : nullptr; }
extern "C" int msq_packed_dev_free(msq_ctx* ctx, msq_packed_dev* p)
{
    if (!p) return MSQ_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    for (int j = 0; j < 8; ++j)
        if (p->ptrs[j]) (void)hipFree(p->ptrs[j]);
    delete p;
    return MSQ_OK;
}
