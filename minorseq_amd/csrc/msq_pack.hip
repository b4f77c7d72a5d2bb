// msq_pack.hip -- host packer (flat ArrayBase columns -> bit-plane groups) and upload.
// Format: include/msq_abi.h "packed read format".
#include "msq_internal.h"

#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct msq_packed_host {
    std::vector<int32_t> ref_start, ref_end;
    std::vector<uint64_t> group_off;
    std::vector<uint32_t> bop;
    std::vector<uint8_t> qual, delqv, subqv, insqv;
    msq_packed_reads view = {};
};


// Per-base translation of (ArrayBase::Cigar, ArrayBase::Nucleotide) as MSAByRow::AddRead switches on
// them (src/MSA.cpp:158-184).  Returns false on an unknown cigar char ("Unexpected cigar", :182).
static inline __host__ __device__ bool encode_base(char cigar, char nuc, unsigned* op, unsigned* code, bool* exotic, bool* ins_exotic)
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
            if (*code == 0 && nuc != 'A') *ins_exotic = true;  // the insertion string keeps the literal char (MSA.cpp:172-174)
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
        bool exotic = false, ins_exotic = false;
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
                    if (!encode_base(cigar[b0 + i], nuc[b0 + i], &op, &code, &exotic, &ins_exotic)) {
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
        v.flags = (exotic ? MSQ_READS_HAS_EXOTIC : 0u) | (ins_exotic ? MSQ_READS_INS_EXOTIC : 0u) | MSQ_READS_NORMALIZED;
        v.align_origin = 0;
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
            for (int j = 0; j < 12; ++j)
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
    d->view.gcol = nullptr;
    d->view.guide = nullptr;  // a host view's guide (if any) is a host pointer: not carried over
    d->view.guide_words = 0;
    d->view.hap_count = 0;
    d->view.hap_guides = nullptr;
    d->view.hap_perm = nullptr;
    for (uint32_t& x : d->view.hap_reads) x = 0;
    if (int rc = msq_packed_dev_index(ctx, d)) {  // the group column index, derived on the device
        (void)hipStreamSynchronize(ctx->stream);
        for (int j = 0; j < 12; ++j)
            if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
        delete d;
        return rc;
    }
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
    for (int j = 0; j < 12; ++j)
        if (p->ptrs[j]) (void)hipFree(p->ptrs[j]);
    delete p;
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------ device packer
// msq_pack_reads_device: the same packed bytes as msq_pack_reads + msq_reads_upload, but the host only copies each
// read's msq_array_base records (the front's ArrayBase, 8 bytes) into pinned staging -- on several threads, chunk by
// chunk, overlapped with the previous chunk's H2D copy -- and the per-base translation (MSAByRow::AddRead's switch,
// src/MSA.cpp:158-184) and the bit-plane packing run on the GPU: one half-wave per 32-base group.
__global__ __launch_bounds__(256) void pack_groups_kernel(const uint64_t* __restrict__ stage, uint64_t chunk_base0,
                                                          const uint64_t* __restrict__ base_off, const uint64_t* __restrict__ group_off,
                                                          uint64_t r0, uint64_t r1, uint64_t g0, uint64_t g1, uint4* __restrict__ bop,
                                                          uint8_t* __restrict__ q_qual, uint8_t* __restrict__ q_del,
                                                          uint8_t* __restrict__ q_sub, uint8_t* __restrict__ q_ins,
                                                          unsigned long long* __restrict__ err, uint32_t* __restrict__ flags, int plain,
                                                          int bake, msq_qv_thresholds thr, int compact)
{
    __shared__ uint8_t nib[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, k = lane & 31;
    const uint64_t g = g0 + ((uint64_t)blockIdx.x * 4 + wave) * 2 + h;
    const bool valid_g = g < g1;
    const uint64_t gs = valid_g ? g : g1 - 1;
    uint64_t lo = r0, hi = r1;  // the read of this group: the last r in [r0, r1) with group_off[r] <= gs
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (group_off[mid] <= gs)
            lo = mid;
        else
            hi = mid;
    }
    const uint64_t b0 = base_off[lo], len = base_off[lo + 1] - b0;
    const uint64_t i = (gs - group_off[lo]) * 32 + (uint64_t)k;
    const bool has = valid_g && i < len;
    // compact staging (the baked packer): one byte per base, translated and thresholded on the host (compact_record)
    const uint8_t c8 = (has && compact) ? reinterpret_cast<const uint8_t*>(stage)[b0 - chunk_base0 + i] : (uint8_t)0;
    const uint64_t w = (has && !compact) ? stage[b0 - chunk_base0 + i] : 0;  // {Cigar, Nucleotide, HasQualQV, HasRichQVs, QualQV, DelQV, SubQV, InsQV}
    unsigned op = 3, code = 0;
    bool exotic = false, ins_exotic = false;
    if (compact) {
        if (has) {
            code = c8 & 3u;
            op = (c8 >> 2) & 3u;
            exotic = (c8 & 16u) != 0;
            ins_exotic = (c8 & 32u) != 0;
        }
    } else if (has && !encode_base((char)(w & 0xff), (char)((w >> 8) & 0xff), &op, &code, &exotic, &ins_exotic)) {
        atomicMin(err, (unsigned long long)(((b0 + i) << 8) | (w & 0xff)));  // the first one in read order is reported
        op = 3;
        code = 0;
    }
    if (exotic) atomicOr(flags, MSQ_READS_HAS_EXOTIC);
    if (ins_exotic) atomicOr(flags, MSQ_READS_INS_EXOTIC);
    if (bake && has && !compact) {
        // the thresholds applied while packing (what msq_reads_bake_qv does to a packed set): a '=' / 'X' base -- op 0,
        // or one of the exotic match bases under op 1, codes 2 and 3 -- that misses a threshold becomes the literal 'N'
        // MSAByRow::AddRead writes for it (src/MSA.cpp:163-166); an absent QV passes (ArrayRead.inl:72-87)
        const bool hq = (w >> 16) & 0xff, hr = (w >> 24) & 0xff;
        const int qq = (int)((w >> 32) & 0xff), qd = (int)((w >> 40) & 0xff), qs = (int)((w >> 48) & 0xff), qi = (int)((w >> 56) & 0xff);
        const bool pass = (thr.qual < 0 || !hq || qq >= thr.qual) && (thr.del < 0 || !hr || qd >= thr.del) &&
                          (thr.sub < 0 || !hr || qs >= thr.sub) && (thr.ins < 0 || !hr || qi >= thr.ins);
        if (!pass && (op == 0 || (op == 1 && code >= 2))) {
            op = 1;
            code = 1;
        }
    }
    // normalized layout: consuming bases from slot 0 up, the others from 31 down, both in genomic order;
    // plain layout (MSQ_PACK_PLAIN): base k in slot k, padding marked with nucleotide code 3
    const bool cons = has && op < 2, ncons = has && op >= 2;
    const uint32_t hc = (uint32_t)(__ballot(cons) >> (32 * h)), hn = (uint32_t)(__ballot(ncons) >> (32 * h));
    const uint32_t below = (1u << k) - 1u;
    const int slot = plain ? k : cons ? __popc(hc & below) : 31 - __popc(hn & below);
    nib[wave][lane] = plain ? 15 : 12;  // padding: op 3
    __syncthreads();
    if (has) nib[wave][h * 32 + slot] = (uint8_t)(code | (op << 2));
    __syncthreads();
    const unsigned v = nib[wave][lane];
    const uint64_t p0 = __ballot(v & 1u), p1 = __ballot(v & 2u), p2 = __ballot(v & 4u), p3 = __ballot(v & 8u);
    if (k == 0 && valid_g) bop[g] = make_uint4((uint32_t)(p0 >> (32 * h)), (uint32_t)(p1 >> (32 * h)), (uint32_t)(p2 >> (32 * h)), (uint32_t)(p3 >> (32 * h)));
    if (has && !compact) {  // an absent QV always passes (ArrayRead.inl:72-87): 255 passes every threshold
        const uint64_t at = g * 32 + (uint64_t)MSQ_QV_BYTE(slot);
        const bool hq = (w >> 16) & 0xff, hr = (w >> 24) & 0xff;
        if (q_qual) q_qual[at] = hq ? (uint8_t)(w >> 32) : (uint8_t)255;
        if (q_del) q_del[at] = hr ? (uint8_t)(w >> 40) : (uint8_t)255;
        if (q_sub) q_sub[at] = hr ? (uint8_t)(w >> 48) : (uint8_t)255;
        if (q_ins) q_ins[at] = hr ? (uint8_t)(w >> 56) : (uint8_t)255;
    }
}

// memcpy of n records that also ORs them together (the HasQualQV / HasRichQVs bytes decide which planes exist)
static uint64_t copy_or(uint64_t* dst, const msq_array_base* src, int64_t n)
{
    uint64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t w;
        std::memcpy(&w, src + i, 8);
        acc |= w;
        dst[i] = w;
    }
    return acc;
}

// The baked packer's staging: one byte per base instead of the 8-byte record.  With the thresholds known while packing,
// everything the kernel would derive from a record is derived here, where the record is read anyway -- translation
// (encode_base, through a table over (Cigar, Nucleotide)), the threshold test (ArrayRead.inl:67-87) and its 'N' -- and
// an eighth of the bytes crosses PCIe: byte = code | op << 2 | exotic << 4 | inserted-exotic << 5; 0x80 = unknown cigar.
extern "C" uint8_t msq_compact_fail(uint8_t c)
{
    const unsigned op = (c >> 2) & 3u, code = c & 3u;
    return (op == 0 || (op == 1 && code >= 2)) ? (uint8_t)((c & 0x30u) | 5u) : c;  // op 1, code 1
}

struct CompactTable {
    uint8_t t[65536];
    CompactTable()
    {
        for (unsigned v = 0; v < 65536; ++v) {
            unsigned op = 3, code = 0;
            bool exotic = false, ins_exotic = false;
            t[v] = encode_base((char)(v & 0xff), (char)(v >> 8), &op, &code, &exotic, &ins_exotic)
                       ? (uint8_t)(code | (op << 2) | (exotic ? 16u : 0u) | (ins_exotic ? 32u : 0u))
                       : (uint8_t)0x80;
        }
    }
};
static const CompactTable& compact_table()
{
    static const CompactTable table;
    return table;
}
// n records -> n bytes; returns the OR of the records; *first_bad = index of the first record with an unknown cigar (or n)
static uint64_t compact_records(uint8_t* dst, const void* src, int64_t n, const msq_qv_thresholds& thr, int64_t* first_bad)
{
    const uint8_t* T = compact_table().t;
    const bool any_thr = thr.qual >= 0 || thr.del >= 0 || thr.sub >= 0 || thr.ins >= 0;
    uint64_t acc = 0;
    int64_t bad = n;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t w;
        std::memcpy(&w, static_cast<const char*>(src) + 8 * i, 8);  // the caller's records need not be 8-byte aligned
        acc |= w;
        uint8_t c = T[w & 0xffff];
        if (c & 0x80) {
            if (bad == n) bad = i;
            c = 12;  // op 3, code 0: what the kernel writes for it
        } else if (any_thr) {
            const bool hq = (w >> 16) & 0xff, hr = (w >> 24) & 0xff;
            const int qq = (int)((w >> 32) & 0xff), qd = (int)((w >> 40) & 0xff), qs = (int)((w >> 48) & 0xff), qi = (int)((w >> 56) & 0xff);
            // without branches (on real QVs the outcome is as good as random per base); an unset threshold is -1: no QV is below it
            const unsigned fail = ((unsigned)hq & (unsigned)(qq < thr.qual)) |
                                  ((unsigned)hr & ((unsigned)(qd < thr.del) | (unsigned)(qs < thr.sub) | (unsigned)(qi < thr.ins)));
            const uint8_t masked = MSQ_COMPACT_FAIL(c);  // op 1, code 1: the literal 'N' (MSA.cpp:163-166)
            c = fail ? masked : c;
        }
        dst[i] = c;
    }
    *first_bad = bad;
    return acc;
}

static void free_packed_dev(msq_packed_dev* d)
{
    for (int j = 0; j < 12; ++j)
        if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
    delete d;
}

// the records of one read into staging: copied from the caller's array, or written there by the caller's callback
// (and then OR-ed together from the staging buffer, which is still in cache)
static uint64_t stage_read(uint64_t* dst, uint64_t r, const msq_array_base* const* bases, const int64_t* n_bases,
                           msq_fill_read_fn fill, void* user)
{
    if (!fill) return copy_or(dst, bases[r], n_bases[r]);
    fill(user, r, reinterpret_cast<msq_array_base*>(dst));
    uint64_t acc = 0;
    for (int64_t i = 0; i < n_bases[r]; ++i) acc |= dst[i];
    return acc;
}

// the same into compact staging; `tmp` (the callback's records) holds the longest read; bad_key: (base index << 8 | char)
// of the first unknown cigar this thread met, in read order
static uint64_t stage_read_compact(uint8_t* dst, uint64_t r, uint64_t base0, const msq_array_base* const* bases, const int64_t* n_bases,
                                   msq_fill_read_fn fill, void* user, std::vector<uint64_t>& tmp, const msq_qv_thresholds& thr,
                                   unsigned long long* bad_key)
{
    const int64_t n = n_bases[r];
    const void* src;
    if (fill) {
        if ((int64_t)tmp.size() < n) tmp.resize((size_t)n);
        fill(user, r, reinterpret_cast<msq_array_base*>(tmp.data()));
        src = tmp.data();
    } else {
        src = bases[r];
    }
    int64_t bad = n;
    const uint64_t acc = compact_records(dst, src, n, thr, &bad);
    if (bad < n) {
        uint64_t w;
        std::memcpy(&w, static_cast<const char*>(src) + 8 * bad, 8);
        const unsigned long long key = ((unsigned long long)(base0 + (uint64_t)bad) << 8) | (w & 0xff);
        if (key < *bad_key) *bad_key = key;
    }
    return acc;
}

static int pack_reads_device_impl(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                  const msq_array_base* const* bases, const int64_t* n_bases, msq_fill_read_fn fill,
                                  void* user, uint32_t pack_flags, int threads, int64_t chunk_bases, const msq_qv_thresholds* bake_thr,
                                  msq_packed_dev** out, msq_fill_read_compact_fn cfill = nullptr)
{
    static_assert(sizeof(msq_array_base) == 8, "msq_array_base is one 8-byte record");
    if (!ctx || !out) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: NULL argument");
    *out = nullptr;
    if (n_reads && (!ref_start || !ref_end || (!bases && !fill && !cfill) || !n_bases)) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: NULL array");
    if (pack_flags & ~MSQ_PACK_PLAIN) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: unknown pack flag");
    const int plain = (pack_flags & MSQ_PACK_PLAIN) ? 1 : 0;
    msq_qv_thresholds thr{-1, -1, -1, -1};
    if (bake_thr) {
        thr = *bake_thr;
        for (int k = 0; k < 4; ++k) {
            const int32_t t = (&thr.qual)[k];
            if (t < -1 || t > 255) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: QV threshold %d out of range", t);
        }
    }
    uint64_t any_all = 0;  // OR of every record of the set: which QVs any base carries
    MSQ_HIP(hipSetDevice(ctx->device));
    std::vector<uint64_t> base_off, group_off;
    int64_t longest = 0;
    try {
        base_off.resize(n_reads + 1);
        group_off.resize(n_reads + 1);
    } catch (const std::bad_alloc&) {
        return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: out of memory");
    }
    base_off[0] = group_off[0] = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (n_bases[r] < 0 || (n_bases[r] > 0 && !fill && !cfill && !bases[r])) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: bad read %llu", (unsigned long long)r);
        base_off[r + 1] = base_off[r] + (uint64_t)n_bases[r];
        group_off[r + 1] = group_off[r] + (uint64_t)(n_bases[r] + 31) / 32;
        longest = std::max(longest, n_bases[r]);
    }
    const uint64_t total_bases = base_off[n_reads], n_groups = group_off[n_reads];
    uint64_t cap = chunk_bases > 0 ? (uint64_t)chunk_bases : (uint64_t)32 << 20;  // records per chunk (256 MB)
    cap = std::max<uint64_t>(std::min(cap, total_bases), (uint64_t)longest);
    // the baked packer over the caller's own records stages one translated byte per base (compact_records: 0.040 instead
    // of 0.056 s per 10^5 reads through the front); MSQ_PACK_WIDE_STAGING=1 keeps the records.  Records a callback writes
    // (the BAM front) go into the pinned buffer as they are: decoding them is what that path waits for, a second pass
    // over them wins nothing (measured)
    const char* wide = getenv("MSQ_PACK_WIDE_STAGING");
    const bool compact = cfill != nullptr || (bake_thr != nullptr && !fill && !(wide && wide[0] == '1'));  // cfill: the caller's bytes
    const size_t stage_bytes = std::max<size_t>(compact ? cap : cap * 8, 64);  // pinned: allocating it is most of a first call
    const unsigned n_thr = threads > 0 ? std::min(64u, (unsigned)threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));

    // staging: two pinned host buffers (kept by the context), one device buffer + the error words in the scratch
    if (ctx->h_stage_bytes < stage_bytes) {
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 2; ++k) {
            if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
            ctx->h_stage[k] = nullptr;
        }
        ctx->h_stage_bytes = 0;
        const size_t want = stage_bytes + (stage_bytes >> 3);
        for (int k = 0; k < 2; ++k) {
            hipError_t e = hipHostMalloc(&ctx->h_stage[k], want, hipHostMallocDefault);
            if (e != hipSuccess) return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        ctx->h_stage_bytes = want;
    }
    for (int k = 0; k < 2; ++k)
        if (!ctx->stage_done[k]) MSQ_HIP(hipEventCreateWithFlags(&ctx->stage_done[k], hipEventDisableTiming));
    void* scratch = nullptr;
    int rc = msq_scratch(ctx, 256 + stage_bytes, &scratch);
    if (rc != MSQ_OK) return rc;
    unsigned long long* d_errword = (unsigned long long*)scratch;
    uint32_t* d_flags = (uint32_t*)((char*)scratch + 8);
    uint64_t* d_stage = (uint64_t*)((char*)scratch + 256);
    MSQ_HIP(hipMemsetAsync(scratch, 0xFF, 8, ctx->stream));
    MSQ_HIP(hipMemsetAsync(d_flags, 0, 8, ctx->stream));

    msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
    if (!d) return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: out of memory");
    void* d_base_off = nullptr;
    auto fail = [&](int code, const char* what, hipError_t e) {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_base_off) (void)hipFree(d_base_off);
        free_packed_dev(d);
        return msq_fail(code, "msq_pack_reads_device: %s: %s", what, hipGetErrorString(e));
    };
    {
        const void* src[4] = {ref_start, ref_end, group_off.data(), nullptr};
        const size_t bytes[4] = {n_reads * 4, n_reads * 4, (n_reads + 1) * 8, n_groups * 16};
        for (int k = 0; k < 4; ++k) {
            hipError_t e = hipMalloc(&d->ptrs[k], bytes[k] ? bytes[k] : 16);
            if (e == hipSuccess && bytes[k] && src[k]) e = hipMemcpyAsync(d->ptrs[k], src[k], bytes[k], hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "read tables", e);
        }
        hipError_t e = hipMalloc(&d_base_off, (n_reads + 1) * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_base_off, base_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "base offsets", e);
    }

    unsigned long long host_bad = ~0ull;  // first unknown cigar seen while compacting: (base index << 8) | char
    uint64_t r0 = 0;
    for (int chunk = 0; r0 < n_reads; ++chunk) {
        uint64_t r1 = r0 + 1;  // at least one read; cap >= the longest read
        while (r1 < n_reads && base_off[r1 + 1] - base_off[r0] <= cap) ++r1;
        const uint64_t cb = base_off[r1] - base_off[r0];
        const int b = chunk & 1;
        if (chunk >= 2) {
            hipError_t e = hipEventSynchronize(ctx->stage_done[b]);  // the copy that last read this pinned buffer
            if (e != hipSuccess) return fail(MSQ_ERR_HIP, "hipEventSynchronize", e);
        }
        uint64_t* hs = (uint64_t*)ctx->h_stage[b];
        uint8_t* hs8 = (uint8_t*)ctx->h_stage[b];
        uint64_t any = 0;
        const unsigned workers = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_thr, cb >> 17));  // >= 1 MB each
        // one read into this chunk's staging, in the mode of the call
        auto stage_one = [&](uint64_t r, std::vector<uint64_t>& tmp, unsigned long long* bad) -> uint64_t {
            const uint64_t at = base_off[r] - base_off[r0];
            if (cfill) {  // the callee translated: its bytes, what it saw of the QVs, its first unknown cigar
                uint32_t presence = 0;
                const int64_t first_bad = cfill(user, r, hs8 + at, &presence);
                if (first_bad >= 0) {
                    const unsigned long long key = ((unsigned long long)(base_off[r] + (uint64_t)(first_bad >> 8)) << 8) | (unsigned long long)(first_bad & 0xff);
                    if (key < *bad) *bad = key;
                }
                return ((presence & 1u) ? 1ull << 16 : 0ull) | ((presence & 2u) ? 1ull << 24 : 0ull);
            }
            return compact ? stage_read_compact(hs8 + at, r, base_off[r], bases, n_bases, fill, user, tmp, thr, bad)
                           : stage_read(hs + at, r, bases, n_bases, fill, user);
        };
        if (workers == 1) {
            try {
                std::vector<uint64_t> tmp;
                for (uint64_t r = r0; r < r1; ++r) any |= stage_one(r, tmp, &host_bad);
            } catch (const std::bad_alloc&) {
                (void)hipStreamSynchronize(ctx->stream);
                if (d_base_off) (void)hipFree(d_base_off);
                free_packed_dev(d);
                return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: out of memory");
            }
        } else {
            uint64_t ors[64] = {0};
            unsigned long long bads[64];
            bool oom[64] = {false};
            for (unsigned t = 0; t < 64; ++t) bads[t] = ~0ull;
            std::vector<std::thread> pool;
            bool spawned = true;
            try {  // no C++ exception may leave through the C ABI, and no joinable thread may be destroyed
                pool.reserve(workers);
                for (unsigned t = 0; t < workers; ++t) {
                    // reads whose first base falls into this worker's share of the chunk's bases
                    const uint64_t lo_b = base_off[r0] + cb * t / workers, hi_b = base_off[r0] + cb * (t + 1) / workers;
                    const uint64_t ra = std::lower_bound(base_off.begin() + r0, base_off.begin() + r1, lo_b) - base_off.begin();
                    const uint64_t rb = t + 1 == workers ? r1 : std::lower_bound(base_off.begin() + r0, base_off.begin() + r1, hi_b) - base_off.begin();
                    pool.emplace_back([&, t, ra, rb] {
                        uint64_t acc = 0;
                        try {
                            std::vector<uint64_t> tmp;
                            for (uint64_t r = ra; r < rb; ++r) acc |= stage_one(r, tmp, &bads[t]);
                        } catch (const std::bad_alloc&) {
                            oom[t] = true;
                        }
                        ors[t] = acc;
                    });
                }
            } catch (...) {
                spawned = false;
            }
            for (auto& th : pool) th.join();
            if (!spawned) {
                (void)hipStreamSynchronize(ctx->stream);
                if (d_base_off) (void)hipFree(d_base_off);
                free_packed_dev(d);
                return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: cannot start %u staging threads", workers);
            }
            bool any_oom = false;
            for (unsigned t = 0; t < workers; ++t) {
                any |= ors[t];
                host_bad = std::min(host_bad, bads[t]);
                any_oom = any_oom || oom[t];
            }
            if (any_oom) {
                (void)hipStreamSynchronize(ctx->stream);
                if (d_base_off) (void)hipFree(d_base_off);
                free_packed_dev(d);
                return msq_fail(MSQ_ERR_NOMEM, "msq_pack_reads_device: out of memory");
            }
        }
        any_all |= any;
        // QV planes appear with the first chunk that carries the QV; every base before it lacked it -> 255.  With the
        // thresholds applied here no plane is written at all.
        const bool want[4] = {((any >> 16) & 0xff) != 0, ((any >> 24) & 0xff) != 0, ((any >> 24) & 0xff) != 0, ((any >> 24) & 0xff) != 0};
        for (int q = 0; q < 4; ++q)
            if (!bake_thr && want[q] && !d->ptrs[4 + q]) {
                hipError_t e = hipMalloc(&d->ptrs[4 + q], n_groups * 32 ? n_groups * 32 : 16);
                if (e == hipSuccess && group_off[r0]) e = hipMemsetAsync(d->ptrs[4 + q], 0xFF, group_off[r0] * 32, ctx->stream);
                if (e == hipSuccess && n_groups > group_off[r0])
                    e = hipMemsetAsync((char*)d->ptrs[4 + q] + group_off[r0] * 32, 0, (n_groups - group_off[r0]) * 32, ctx->stream);
                if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "QV plane", e);
            }
        if (cb) {
            hipError_t e = hipMemcpyAsync(d_stage, hs, compact ? cb : cb * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->stage_done[b], ctx->stream);
            if (e != hipSuccess) return fail(MSQ_ERR_HIP, "staging copy", e);
            const uint64_t g0 = group_off[r0], g1 = group_off[r1];
            const uint64_t blocks = (g1 - g0 + 7) / 8;  // 4 waves x 2 groups
            hipLaunchKernelGGL(pack_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_stage, base_off[r0],
                               (const uint64_t*)d_base_off, (const uint64_t*)d->ptrs[2], r0, r1, g0, g1, (uint4*)d->ptrs[3],
                               (uint8_t*)d->ptrs[4], (uint8_t*)d->ptrs[5], (uint8_t*)d->ptrs[6], (uint8_t*)d->ptrs[7], d_errword, d_flags, plain,
                               bake_thr ? 1 : 0, thr, compact ? 1 : 0);
            e = hipGetLastError();
            if (e != hipSuccess) return fail(MSQ_ERR_HIP, "pack kernel", e);
        } else {
            hipError_t e = hipEventRecord(ctx->stage_done[b], ctx->stream);
            if (e != hipSuccess) return fail(MSQ_ERR_HIP, "hipEventRecord", e);
        }
        r0 = r1;
    }
    uint64_t res[2] = {~0ull, 0};
    {
        hipError_t e = hipMemcpyAsync(res, scratch, 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "result copy", e);
    }
    (void)hipFree(d_base_off);
    d_base_off = nullptr;
    res[0] = std::min<uint64_t>(res[0], host_bad);
    if (res[0] != ~0ull) {
        free_packed_dev(d);
        // std::runtime_error("Unexpected cigar " + std::to_string(b.Cigar)), src/MSA.cpp:182
        return msq_fail(MSQ_ERR_CIGAR, "Unexpected cigar %d", (int)(res[0] & 0xff));
    }
    msq_packed_reads& v = d->view;
    v.n_reads = n_reads;
    v.n_groups = n_groups;
    v.ref_start = (const int32_t*)d->ptrs[0];
    v.ref_end = (const int32_t*)d->ptrs[1];
    v.group_off = (const uint64_t*)d->ptrs[2];
    v.bop = (const uint32_t*)d->ptrs[3];
    v.qual = (const uint8_t*)d->ptrs[4];
    v.delqv = (const uint8_t*)d->ptrs[5];
    v.subqv = (const uint8_t*)d->ptrs[6];
    v.insqv = (const uint8_t*)d->ptrs[7];
    v.flags = ((uint32_t)res[1] & (MSQ_READS_HAS_EXOTIC | MSQ_READS_INS_EXOTIC)) | (plain ? MSQ_READS_PAD_MARKED : MSQ_READS_NORMALIZED);
    if (bake_thr) {  // as msq_reads_bake_qv flags its result: a threshold met a QV some base carries
        const bool has_q = ((any_all >> 16) & 0xff) != 0, has_r = ((any_all >> 24) & 0xff) != 0;
        if ((thr.qual >= 0 && has_q) || (has_r && (thr.del >= 0 || thr.sub >= 0 || thr.ins >= 0))) v.flags |= MSQ_READS_QV_BAKED;
    }
    v.align_origin = 0;
    v.gcol = nullptr;
    if (int rc2 = msq_packed_dev_index(ctx, d)) {
        (void)hipStreamSynchronize(ctx->stream);
        free_packed_dev(d);
        return rc2;
    }
    *out = d;
    return MSQ_OK;
}

// pack_groups_kernel's compact mode for bytes that are all in HBM already: a wave per READ walks its bases 64 at a time
// (two groups), so no group has to look for its read (the binary search over the read tables was most of that kernel's
// 3.8 ms per 10^5 reads here).  Same slots, same planes.
__global__ __launch_bounds__(256) void pack_compact_reads_kernel(const uint8_t* __restrict__ compact, const uint64_t* __restrict__ base_off,
                                                                 const uint64_t* __restrict__ group_off, uint64_t n_reads, uint4* __restrict__ bop,
                                                                 uint32_t* __restrict__ flags, int plain)
{
    __shared__ uint8_t nib[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, k = lane & 31;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (uint64_t)wave;
    if (r >= n_reads) return;
    const uint64_t b0 = base_off[r], len = base_off[r + 1] - b0, g0 = group_off[r], ng = group_off[r + 1] - g0;
    const uint32_t below = (1u << k) - 1u;
    uint32_t seen = 0;
    for (uint64_t i0 = 0; i0 < len; i0 += 64) {
        const uint64_t i = i0 + (uint64_t)lane;
        const bool has = i < len;
        const uint8_t c8 = has ? compact[b0 + i] : (uint8_t)0;
        const unsigned code = has ? (c8 & 3u) : 0u, op = has ? ((c8 >> 2) & 3u) : 3u;
        seen |= c8 & 0x30u;
        const bool cons = has && op < 2, ncons = has && op >= 2;
        const uint32_t hc = (uint32_t)(__ballot(cons) >> (32 * h)), hn = (uint32_t)(__ballot(ncons) >> (32 * h));
        const int slot = plain ? k : cons ? __popc(hc & below) : 31 - __popc(hn & below);
        nib[wave][lane] = plain ? 15 : 12;  // padding: op 3
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has) nib[wave][h * 32 + slot] = (uint8_t)(code | (op << 2));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned v = nib[wave][lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t p0 = __ballot(v & 1u), p1 = __ballot(v & 2u), p2 = __ballot(v & 4u), p3 = __ballot(v & 8u);
        const uint64_t gi = i0 / 32 + (uint64_t)h;
        if (k == 0 && gi < ng)
            bop[g0 + gi] = make_uint4((uint32_t)(p0 >> (32 * h)), (uint32_t)(p1 >> (32 * h)), (uint32_t)(p2 >> (32 * h)), (uint32_t)(p3 >> (32 * h)));
    }
    const bool exotic = __ballot(seen & 16u) != 0, ins_exotic = __ballot(seen & 32u) != 0;
    if (lane == 0 && exotic) atomicOr(flags, MSQ_READS_HAS_EXOTIC);
    if (lane == 0 && ins_exotic) atomicOr(flags, MSQ_READS_INS_EXOTIC);
}

// The device packer over translated bytes that are already in HBM (the BAM front on the device, msq_bam_pack_device):
// the tail of pack_reads_device_impl without staging -- read tables up, one pack_groups_kernel over all groups, flags,
// column index.  Same packed bytes as msq_pack_reads_device_fill_compact on the same bytes.  The kernel that writes
// d_compact and *d_presence (which QV strings the reads carry) may still be running on the context's stream: the tables
// are made and the set's buffers allocated beside it, and the presence word comes back with the pack kernel's flags.
int msq_pack_compact_resident(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end, const uint64_t* base_off,
                              const uint64_t* d_base_off, const uint8_t* d_compact, const uint32_t* d_presence, uint32_t pack_flags,
                              const msq_qv_thresholds* thr_in, msq_packed_dev** out)
{
    *out = nullptr;
    if (pack_flags & ~MSQ_PACK_PLAIN) return msq_fail(MSQ_ERR_ARG, "msq_bam_pack_device: unknown pack flag");
    const int plain = (pack_flags & MSQ_PACK_PLAIN) ? 1 : 0;
    const msq_qv_thresholds thr = *thr_in;
    std::vector<uint64_t> group_off;
    try {
        group_off.resize(n_reads + 1);
    } catch (const std::bad_alloc&) {
        return msq_fail(MSQ_ERR_NOMEM, "msq_bam_pack_device: out of memory");
    }
    group_off[0] = 0;
    for (uint64_t r = 0; r < n_reads; ++r) group_off[r + 1] = group_off[r] + (base_off[r + 1] - base_off[r] + 31) / 32;
    const uint64_t n_groups = group_off[n_reads];
    void* scratch = nullptr;
    if (int rc = msq_scratch(ctx, 256, &scratch)) return rc;
    uint32_t* d_flags = (uint32_t*)((char*)scratch + 8);
    MSQ_HIP(hipMemsetAsync(scratch, 0xFF, 8, ctx->stream));
    MSQ_HIP(hipMemsetAsync(d_flags, 0, 8, ctx->stream));
    msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
    if (!d) return msq_fail(MSQ_ERR_NOMEM, "msq_bam_pack_device: out of memory");
    auto fail = [&](int code, const char* what, hipError_t e) {
        (void)hipStreamSynchronize(ctx->stream);
        free_packed_dev(d);
        return msq_fail(code, "msq_bam_pack_device: %s: %s", what, hipGetErrorString(e));
    };
    const void* src[4] = {ref_start, ref_end, group_off.data(), nullptr};
    const size_t bytes[4] = {n_reads * 4, n_reads * 4, (n_reads + 1) * 8, n_groups * 16};
    for (int k = 0; k < 4; ++k) {
        hipError_t e = hipMalloc(&d->ptrs[k], bytes[k] ? bytes[k] : 16);
        if (e == hipSuccess && bytes[k] && src[k]) e = hipMemcpyAsync(d->ptrs[k], src[k], bytes[k], hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "read tables", e);
    }
    if (n_reads) {
        const uint64_t blocks = (n_reads + 3) / 4;  // a wave per read
        if (blocks > 0x7fffffffull) return fail(MSQ_ERR_ARG, "too many reads for one launch", hipErrorInvalidValue);
        hipLaunchKernelGGL(pack_compact_reads_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_compact, d_base_off, (const uint64_t*)d->ptrs[2],
                           n_reads, (uint4*)d->ptrs[3], d_flags, plain);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "pack kernel", e);
    }
    uint64_t res[2] = {~0ull, 0};
    uint32_t presence = 0;
    {
        hipError_t e = hipMemcpyAsync(res, scratch, 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&presence, d_presence, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(MSQ_ERR_HIP, "result copy", e);
    }
    msq_packed_reads& v = d->view;
    v.n_reads = n_reads;
    v.n_groups = n_groups;
    v.ref_start = (const int32_t*)d->ptrs[0];
    v.ref_end = (const int32_t*)d->ptrs[1];
    v.group_off = (const uint64_t*)d->ptrs[2];
    v.bop = (const uint32_t*)d->ptrs[3];
    v.qual = v.delqv = v.subqv = v.insqv = nullptr;
    v.flags = ((uint32_t)res[1] & (MSQ_READS_HAS_EXOTIC | MSQ_READS_INS_EXOTIC)) | (plain ? MSQ_READS_PAD_MARKED : MSQ_READS_NORMALIZED);
    const bool has_q = (presence & 1u) != 0, has_r = (presence & 2u) != 0;
    if ((thr.qual >= 0 && has_q) || (has_r && (thr.del >= 0 || thr.sub >= 0 || thr.ins >= 0))) v.flags |= MSQ_READS_QV_BAKED;
    v.align_origin = 0;
    v.gcol = nullptr;
    if (int rc2 = msq_packed_dev_index(ctx, d)) {
        (void)hipStreamSynchronize(ctx->stream);
        free_packed_dev(d);
        return rc2;
    }
    *out = d;
    return MSQ_OK;
}

extern "C" int msq_pack_reads_device(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                     const msq_array_base* const* bases, const int64_t* n_bases, uint32_t pack_flags,
                                     int threads, int64_t chunk_bases, msq_packed_dev** out)
{
    if (n_reads && !bases) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device: NULL array");
    return pack_reads_device_impl(ctx, n_reads, ref_start, ref_end, bases, n_bases, nullptr, nullptr, pack_flags, threads, chunk_bases, nullptr, out);
}

extern "C" int msq_pack_reads_device_baked(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                           const msq_array_base* const* bases, const int64_t* n_bases, uint32_t pack_flags,
                                           int threads, int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out)
{
    if (n_reads && !bases) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_baked: NULL array");
    if (!thr) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_baked: thr is NULL");
    return pack_reads_device_impl(ctx, n_reads, ref_start, ref_end, bases, n_bases, nullptr, nullptr, pack_flags, threads, chunk_bases, thr, out);
}

extern "C" int msq_pack_reads_device_fill(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                          const int64_t* n_bases, msq_fill_read_fn fill, void* user, uint32_t pack_flags,
                                          int threads, int64_t chunk_bases, msq_packed_dev** out)
{
    if (!fill) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_fill: fill is NULL");
    return pack_reads_device_impl(ctx, n_reads, ref_start, ref_end, nullptr, n_bases, fill, user, pack_flags, threads, chunk_bases, nullptr, out);
}

extern "C" const uint8_t* msq_compact_table(void) { return compact_table().t; }

extern "C" int msq_pack_reads_device_fill_compact(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                                  const int64_t* n_bases, msq_fill_read_compact_fn fill, void* user, uint32_t pack_flags,
                                                  int threads, int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out)
{
    if (!fill) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_fill_compact: fill is NULL");
    if (!thr) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_fill_compact: thr is NULL");
    return pack_reads_device_impl(ctx, n_reads, ref_start, ref_end, nullptr, n_bases, nullptr, user, pack_flags, threads, chunk_bases, thr, out, fill);
}

extern "C" int msq_pack_reads_device_fill_baked(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                                const int64_t* n_bases, msq_fill_read_fn fill, void* user, uint32_t pack_flags,
                                                int threads, int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out)
{
    if (!fill) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_fill_baked: fill is NULL");
    if (!thr) return msq_fail(MSQ_ERR_ARG, "msq_pack_reads_device_fill_baked: thr is NULL");
    return pack_reads_device_impl(ctx, n_reads, ref_start, ref_end, nullptr, n_bases, fill, user, pack_flags, threads, chunk_bases, thr, out);
}
