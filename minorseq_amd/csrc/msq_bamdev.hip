// msq_bamdev.hip -- the BAM front on the device, behind the inflate: the record chain, the filters and
// measurements of IO::BamUtils::BamToArrayReads (src/BamUtils.cpp:56-78) and BAMArrayRead's per-base walk
// (src/ArrayRead.cpp:53-99) over an inflated BAM stream that never leaves HBM.  The host front, cpp/msq_bam.hpp, is the
// specification: every rule below cites the function of it (Load, CountBases, DecodeBasesCompact, FindZTag) that it
// restates, and tests compare the two record by record and packed bit by packed bit.
//
//  * The chain.  A record's place is only known from its predecessor's block_size -- a serial walk of 10^5 dependent
//    HBM loads.  Instead the stream is cut into segments; a wave per segment looks for the first offset that holds a
//    plausible record header followed by another (or the end of the stream), walks its own chain to the segment's end
//    and reports (guess, records, landing point).  The host then checks that every segment's guess IS its
//    predecessor's landing point, starting from the byte behind the BAM header: by induction every start is a true one.
//    A guess that is wrong, a broken length, a chain that misses the end of the stream: the call declines
//    (MSQ_ERR_DATA) and the caller runs the host front, whose error reporting stays the only one.
//  * A wave per record: consistency, filters, the cigar's reference length and the number of ArrayBases inside the
//    region (wave prefix sums over the cigar ops), the sq / dq / iq / RG tags (a Z string is scanned 64 bytes at a time).
//  * A wave per kept record: the cigar in chunks of 64 ops (one lane each: reference / query / output offsets by wave
//    prefix sums), then run by run all lanes translate bases -- SEQ nibble through the library's (Cigar, Nucleotide)
//    table, thresholds as ArrayBase::MeetQVThresholds -- into the one-byte-per-base form the device packer reads.
#include "msq_internal.h"

#include <algorithm>
#include <chrono>
#include <new>
#include <vector>

#include "msq_bam_dev.h"

namespace {

constexpr uint64_t NONE = ~0ull;

__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline uint32_t ldu16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ inline uint32_t ldu32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// inclusive wave prefix sum (64 lanes)
__device__ inline int64_t wave_incl_scan(int64_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// Could a record start at `off` (the offset of its block_size word)?  The fields a BAM record cannot be without.
__device__ inline bool plausible(const uint8_t* d, uint64_t data_bytes, uint64_t off, int32_t n_ref, uint64_t* next)
{
    if (off + 36 > data_bytes) return false;
    const uint64_t bs = ldu32(d + off);
    if (bs < 32 || bs > (1ull << 28) || off + 4 + bs > data_bytes) return false;
    const int32_t ref_id = (int32_t)ldu32(d + off + 4), pos = (int32_t)ldu32(d + off + 8);
    const uint32_t l_read_name = d[off + 12], n_cigar = ldu16(d + off + 16);
    const uint64_t l_seq = ldu32(d + off + 20);
    const int32_t next_ref = (int32_t)ldu32(d + off + 24), next_pos = (int32_t)ldu32(d + off + 28);
    if (ref_id < -1 || ref_id >= n_ref || next_ref < -1 || next_ref >= n_ref || pos < -1 || next_pos < -1) return false;
    if (l_read_name == 0) return false;
    if (32 + (uint64_t)l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq > bs) return false;
    if (d[off + 36 + l_read_name - 1] != 0) return false;
    *next = off + 4 + bs;
    return true;
}

// a wave per segment: guess[k] = the first plausible record start in the segment (segment 0: the known first record),
// count[k] = records that start in the segment on the chain from there, land[k] = where that chain leaves the segment
// (NONE: a length that cannot be)
__global__ __launch_bounds__(256) void bam_chain_kernel(const uint8_t* __restrict__ d, uint64_t data_bytes, uint64_t first_off, uint64_t seg_bytes,
                                                        uint64_t n_seg, int32_t n_ref, uint64_t* __restrict__ guess, uint64_t* __restrict__ land,
                                                        uint32_t* __restrict__ count)
{
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t k = (uint64_t)blockIdx.x * 4 + uni(threadIdx.x >> 6);
    if (k >= n_seg) return;
    const uint64_t seg_begin = first_off + k * seg_bytes, seg_end = min(seg_begin + seg_bytes, data_bytes);
    uint64_t g = NONE;
    if (k == 0) {
        g = first_off;
    } else {
        for (uint64_t base = seg_begin; base < seg_end; base += 64) {
            const uint64_t off = base + (uint64_t)lane;
            uint64_t nx = 0, nx2 = 0;
            bool ok = off < seg_end && plausible(d, data_bytes, off, n_ref, &nx);
            if (ok && nx != data_bytes) ok = plausible(d, data_bytes, nx, n_ref, &nx2);
            const uint64_t m = __ballot(ok);
            if (m) {
                g = base + (uint64_t)__builtin_ctzll(m);
                break;
            }
        }
    }
    uint64_t cur = g;
    uint32_t n = 0;
    if (g != NONE) {
        while (cur < seg_end) {  // every step moves on by at least 36 bytes
            if (cur + 4 > data_bytes) {
                cur = NONE;
                break;
            }
            const uint64_t bs = ((uint64_t)uni(ldu32(d + cur)));
            if (bs < 32 || cur + 4 + bs > data_bytes) {
                cur = NONE;
                break;
            }
            ++n;
            cur += 4 + bs;
        }
    }
    if (lane == 0) {
        guess[k] = g;
        land[k] = cur;
        count[k] = n;
    }
}

// a thread per segment whose guess the host has confirmed: the body offsets of its records, at the segment's rank
__global__ __launch_bounds__(256) void bam_chain_write_kernel(const uint8_t* __restrict__ d, uint64_t data_bytes, const uint64_t* __restrict__ start,
                                                              const uint64_t* __restrict__ rank, const uint32_t* __restrict__ count, uint64_t n_seg,
                                                              uint64_t* __restrict__ rec_off)
{
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_seg) return;
    uint64_t cur = start[k];
    const uint32_t n = count[k];
    for (uint32_t i = 0; i < n; ++i) {
        if (cur + 4 > data_bytes) break;  // cannot happen behind the host's check; never read past the stream
        rec_off[rank[k] + i] = cur + 4;
        cur += 4 + (uint64_t)ldu32(d + cur);
    }
}

__device__ inline bool op_cons_ref(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
__device__ inline bool op_cons_query(uint32_t op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }

// a wave per record: BamUtils::Load pass 1b + CountBases + the tag lookups of DecodeBasesCompact / ChemistryOf
__global__ __launch_bounds__(256) void bam_record_kernel(const uint8_t* __restrict__ d, uint64_t data_bytes, const uint64_t* __restrict__ rec_off,
                                                         uint64_t n_rec, int32_t region_start, int32_t region_end, msq_bam_record* __restrict__ out)
{
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t r = (uint64_t)blockIdx.x * 4 + uni(threadIdx.x >> 6);
    if (r >= n_rec) return;
    const uint64_t body = rec_off[r];
    const uint8_t* rec = d + body;
    const uint64_t bs = uni(ldu32(rec - 4));
    msq_bam_record o;
    memset(&o, 0, sizeof o);
    o.body_off = body;
    o.block_size = (uint32_t)bs;
    const uint32_t l_read_name = uni(rec[8]), n_cigar = uni(ldu16(rec + 12)), flag = uni(ldu16(rec + 14));
    const uint64_t l_seq = uni(ldu32(rec + 16));
    o.ref_start = (int32_t)uni(ldu32(rec + 4));
    o.l_seq = (uint32_t)l_seq;
    o.n_cigar = (uint16_t)n_cigar;
    o.flag = (uint16_t)flag;
    o.l_read_name = (uint8_t)l_read_name;
    o.rg_len = 255;
    o.status = MSQ_BAM_REC_SKIP;
    const uint64_t fixed = 32 + (uint64_t)l_read_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
    if (fixed > bs || l_read_name == 0 || rec[32 + l_read_name - 1] != 0) {  // "malformed BAM record" (Load, pass 1b); a name without its NUL: host's call
        o.status = MSQ_BAM_REC_MALFORMED;
        if (lane == 0) out[r] = o;
        return;
    }
    const uint8_t* cigar = rec + 32 + l_read_name;
    const uint8_t* seq = cigar + 4ull * n_cigar;
    const uint8_t* qual = seq + (l_seq + 1) / 2;
    const uint8_t* tags = qual + l_seq;
    const uint8_t* end = rec + bs;
    o.has_qual = (l_seq > 0 && uni(qual[0]) != 0xFF) ? 1 : 0;
    // pass A over the cigar: reference length, query length, ops that cannot be
    int64_t ref_len = 0, qry_len = 0;
    bool bad_op = false;
    for (uint32_t c0 = 0; c0 < n_cigar; c0 += 64) {
        const uint32_t k = c0 + (uint32_t)lane;
        const uint32_t cg = k < n_cigar ? ldu32(cigar + 4ull * k) : 5u;  // a zero-length 'H' consumes nothing
        const uint32_t op = cg & 15u;
        const int64_t len = cg >> 4;
        bad_op = bad_op || __ballot(op > 8) != 0;
        ref_len += __shfl(wave_incl_scan(op_cons_ref(op) ? len : 0, lane), 63, 64);
        qry_len += __shfl(wave_incl_scan(op_cons_query(op) ? len : 0, lane), 63, 64);
    }
    const int64_t ref_start = o.ref_start, ref_end = ref_start + ref_len;
    o.ref_end = (int32_t)ref_end;
    const bool candidate = !(flag & 0x800u) && !(flag & 0x100u) && !(flag & 0x4u) && ref_start < (int64_t)region_end && ref_end > (int64_t)region_start;
    if (!candidate) {  // BamUtils.cpp:69-71
        if (lane == 0) out[r] = o;
        return;
    }
    o.status = MSQ_BAM_REC_KEEP;
    if (bad_op || qry_len > (int64_t)l_seq) o.status = MSQ_BAM_REC_CIGAR;  // "bad CIGAR op", "CIGAR longer than SEQ" (DecodeBases)
    // pass B: the ArrayBases inside the clip interval (CountBases); an 'M' or 'N' run among them is the MSA's exception
    const int64_t clip_lo = max(ref_start, (int64_t)region_start), clip_hi = min(ref_end, (int64_t)region_end);
    int64_t rp0 = ref_start, n_bases = 0;
    bool msa_throws = false;
    for (uint32_t c0 = 0; c0 < n_cigar; c0 += 64) {
        const uint32_t k = c0 + (uint32_t)lane;
        const uint32_t cg = k < n_cigar ? ldu32(cigar + 4ull * k) : 5u;
        const uint32_t op = cg & 15u;
        const int64_t len = cg >> 4;
        const bool cr = op_cons_ref(op);
        const int64_t incl = wave_incl_scan(cr ? len : 0, lane);
        const int64_t rp = rp0 + incl - (cr ? len : 0);
        int64_t cnt = 0;
        if (op == 4 || op == 5 || op > 8) {
            cnt = 0;
        } else if (cr) {
            const int64_t first = max(rp, clip_lo), last = min(rp + len, clip_hi);
            cnt = last > first ? last - first : 0;
        } else if (rp >= clip_lo && rp <= clip_hi) {
            cnt = len;
        }
        msa_throws = msa_throws || __ballot(cnt > 0 && (op == 0 || op == 3)) != 0;
        n_bases += __shfl(wave_incl_scan(cnt, lane), 63, 64);
        rp0 += __shfl(incl, 63, 64);
    }
    o.n_bases = n_bases;
    if (o.status == MSQ_BAM_REC_KEEP && msa_throws) o.status = MSQ_BAM_REC_MSA;
    // the tags (FindZTag): the first Z value of each of sq, dq, iq, RG; the walk ends at a type it does not know
    uint64_t found_off[4] = {0, 0, 0, 0}, found_len[4] = {0, 0, 0, 0};
    bool found[4] = {false, false, false, false};
    const uint8_t* p = tags;
    while (p + 3 <= end) {
        const uint32_t a = uni(p[0]), b = uni(p[1]), type = uni(p[2]);
        p += 3;
        uint64_t skip = 0;
        if (type == 'A' || type == 'c' || type == 'C') {
            skip = 1;
        } else if (type == 's' || type == 'S') {
            skip = 2;
        } else if (type == 'i' || type == 'I' || type == 'f') {
            skip = 4;
        } else if (type == 'Z' || type == 'H') {
            const uint8_t* s = p;
            uint64_t len = (uint64_t)(end - p);
            for (const uint8_t* q = p; q < end; q += 64) {
                const bool stop = q + lane >= end || q[lane] == 0;
                const uint64_t m = __ballot(stop);
                if (m) {
                    len = min((uint64_t)(q - p) + (uint64_t)__builtin_ctzll(m), (uint64_t)(end - p));
                    break;
                }
            }
            p += len;
            if (type == 'Z') {
                const int which = (a == 's' && b == 'q') ? 0 : (a == 'd' && b == 'q') ? 1 : (a == 'i' && b == 'q') ? 2 : (a == 'R' && b == 'G') ? 3 : -1;
                if (which >= 0 && !found[which]) {
                    found[which] = true;
                    found_off[which] = (uint64_t)(s - rec);
                    found_len[which] = len;
                }
            }
            skip = 1;  // the terminating NUL
        } else if (type == 'B') {
            if (p + 5 > end) break;
            const uint32_t sub = uni(p[0]);
            const uint64_t n = uni(ldu32(p + 1));
            const uint64_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            skip = 5 + n * w;
        } else {
            break;
        }
        if (skip > (uint64_t)(end - p)) break;  // the host's pointer runs past `end` and its loop ends
        p += skip;
    }
    if (found[0] && found[1] && found[2] && found_len[0] == l_seq && found_len[1] == l_seq && found_len[2] == l_seq) {
        o.sq_off = (uint32_t)found_off[0];
        o.dq_off = (uint32_t)found_off[1];
        o.iq_off = (uint32_t)found_off[2];
    }
    if (found[3]) {
        o.rg_len = found_len[3] <= 16 ? (uint8_t)found_len[3] : (uint8_t)254;
        for (uint64_t i = 0; i < found_len[3] && i < 16; ++i) o.rg[i] = (char)uni(rec[found_off[3] + i]);
    }
    if (lane == 0) out[r] = o;
}

// a wave per kept record: its name (l_read_name bytes, NUL included) to names[name_off[r] ..]
__global__ __launch_bounds__(256) void bam_names_kernel(const uint8_t* __restrict__ d, const msq_bam_record* __restrict__ kept, uint64_t n_kept,
                                                        const uint64_t* __restrict__ name_off, char* __restrict__ names)
{
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_kept) return;
    const uint8_t* src = d + kept[r].body_off + 32;
    const uint32_t n = kept[r].l_read_name;
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) names[name_off[r] + i] = (char)src[i];
}

struct CompactRows {  // per cigar op "MIDNSHP=X": the byte of each SEQ code "=ACMGRSVTWYHKDBN", and of the op's placeholder base
    uint8_t seq[9][16];
    uint8_t gap[9];
};

struct RunTable {  // a chunk of <= 64 cigar runs of one record (a wave's own): output offset within the chunk, first query base, op (0xff: writes nothing)
    uint32_t o_rel[64], q_at[64];
    uint8_t op[64];
};

// one wave owns its RunTable and the LDS performs a wave's operations in order: the compiler must keep them in order too
__device__ inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// DecodeBasesCompact, a wave per kept record
// timing-experiment switches (tools/dec_exp.sh): waves per workgroup (16: 1.34 instead of 0.86 ms), passes unrolled (8: slower)
#ifndef MSQ_DECODE_WAVES
#define MSQ_DECODE_WAVES 4
#endif
constexpr int kDecodeWaves = MSQ_DECODE_WAVES;  // reads (waves) per workgroup
__global__ __launch_bounds__(64 * kDecodeWaves) void bam_decode_kernel(const uint8_t* __restrict__ d, const msq_bam_record* __restrict__ kept, uint64_t n_reads,
                                                         const uint64_t* __restrict__ base_off, int32_t region_start, int32_t region_end,
                                                         msq_qv_thresholds thr, CompactRows rows_arg, uint8_t* __restrict__ dst_all,
                                                         uint32_t* __restrict__ presence)
{
    __shared__ CompactRows rows;
    __shared__ RunTable runs[kDecodeWaves];
    for (uint32_t i = threadIdx.x; i < sizeof(CompactRows); i += 64 * kDecodeWaves) reinterpret_cast<uint8_t*>(&rows)[i] = reinterpret_cast<const uint8_t*>(&rows_arg)[i];
    __syncthreads();
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t r = (uint64_t)blockIdx.x * kDecodeWaves + uni(threadIdx.x >> 6);
    if (r >= n_reads) return;
    RunTable& my = runs[uni(threadIdx.x >> 6)];
    const msq_bam_record k = kept[r];
    const uint8_t* rec = d + k.body_off;
    const uint32_t n_cigar = k.n_cigar;
    const uint64_t l_seq = k.l_seq;
    const uint8_t* cigar = rec + 32 + k.l_read_name;
    const uint8_t* seq = cigar + 4ull * n_cigar;
    const uint8_t* qual = seq + (l_seq + 1) / 2;
    const bool has_qual = k.has_qual != 0, rich = k.sq_off != 0, reverse = (k.flag & 0x10u) != 0;
    const uint8_t *sq = rec + k.sq_off, *dq = rec + k.dq_off, *iq = rec + k.iq_off;
    const bool any_thr = thr.qual >= 0 || thr.del >= 0 || thr.sub >= 0 || thr.ins >= 0;
    const bool need_qual = has_qual && thr.qual > 0, need_del = thr.del > 0, need_sub = thr.sub > 0, need_ins = thr.ins > 0;
    const bool no_qual_fails = !has_qual && 0 < thr.qual;  // without QUAL every base carries QualQV 0
    // a placeholder base ('-' / '*', QualQV 0, rich QVs 0) misses a threshold above 0
    const bool gap_fails = any_thr && (0 < thr.qual || (rich && (0 < thr.del || 0 < thr.sub || 0 < thr.ins)));
    const int64_t ref_start = k.ref_start, ref_end = k.ref_end;
    const int64_t clip_lo = max(ref_start, (int64_t)region_start), clip_hi = min(ref_end, (int64_t)region_end);
    uint8_t* dst = dst_all + (base_off[r] - base_off[0]);
    const int64_t room = (int64_t)(base_off[r + 1] - base_off[r]);  // = n_bases; nothing is ever written beyond it
    int64_t rp0 = ref_start, qp0 = 0, out0 = 0;
    for (uint32_t c0 = 0; c0 < n_cigar; c0 += 64) {
        const uint32_t kk = c0 + (uint32_t)lane;
        const uint32_t cg = kk < n_cigar ? ldu32(cigar + 4ull * kk) : 5u;
        const uint32_t op = cg & 15u;
        const int64_t len = cg >> 4;
        const bool cr = op_cons_ref(op), cq = op_cons_query(op);
        const int64_t ri = wave_incl_scan(cr ? len : 0, lane), qi = wave_incl_scan(cq ? len : 0, lane);
        const int64_t rp = rp0 + ri - (cr ? len : 0), qp = qp0 + qi - (cq ? len : 0);
        int64_t skip = 0, cnt = 0;
        if (op == 4 || op == 5 || op > 8) {
            cnt = 0;
        } else if (cr) {
            const int64_t first = max(rp, clip_lo), last = min(rp + len, clip_hi);
            skip = max(first - rp, (int64_t)0);
            cnt = last > first ? last - first : 0;
        } else if (rp >= clip_lo && rp <= clip_hi) {
            cnt = len;
        }
        const int64_t oi = wave_incl_scan(cnt, lane);
        const int64_t o_at = out0 + oi - cnt, q_at = qp + skip;
        // The chunk's runs laid out for a flat walk over its OUTPUT bases: a read has ~50 runs of ~60 bases, and a loop
        // over the runs is a chain of ~50 memory round trips (load the run's bases, then store them); with the runs'
        // output offsets in LDS every lane finds the run of its output base by bisection, and the bases of all runs are
        // loaded side by side.  A run that would not fit (cannot happen: n_bases is this same arithmetic) or that reaches
        // beyond SEQ ("CIGAR longer than SEQ": such a record is never packed) writes nothing, as before.
        const bool fits = o_at + cnt <= room && (!cq || q_at + cnt <= (int64_t)l_seq);
        const int64_t total_here = __shfl(oi, 63, 64);
        wave_lds_sync();  // the previous chunk's readers are done
        my.o_rel[lane] = (uint32_t)(o_at - out0);
        my.q_at[lane] = (uint32_t)q_at;
        my.op[lane] = (uint8_t)((cnt > 0 && fits) ? op : 0xffu);
        wave_lds_sync();
#ifndef MSQ_DECODE_UNROLL
#define MSQ_DECODE_UNROLL 4
#endif
#pragma unroll MSQ_DECODE_UNROLL
        for (int64_t p = lane; p < total_here; p += 64) {
            // the last run that starts at or before p (runs without output share their successor's offset)
            uint32_t j = 0;
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1)
                if (my.o_rel[j + step] <= (uint32_t)p) j += step;
            // no branch on the lane up to the store: the loads of the unrolled passes can then be in flight together
            const uint32_t op_raw = my.op[j];
            const bool live = op_raw != 0xffu;
            const uint32_t op_j = live ? op_raw : 0u;
            const bool from_seq = live && op_cons_query(op_j);
            const uint64_t q = from_seq ? (uint64_t)my.q_at[j] + ((uint64_t)p - my.o_rel[j]) : 0ull;  // a base of SEQ, or SEQ's first byte for nothing
            const uint32_t nib = (seq[q >> 1] >> ((q & 1) ? 0 : 4)) & 15u;
            uint8_t c = from_seq ? rows.seq[op_j][nib] : rows.gap[op_j];
            if (c & 0x80) c = 12;  // a cigar the MSA throws on: flagged by the record pass, never packed
            if (any_thr) {
                // ArrayBase::MeetQVThresholds (ArrayRead.inl:67-87): every base carries a QualQV (0 without QUAL,
                // ArrayRead.cpp:94-96), the rich QVs when all three tags are there (native orientation); a placeholder
                // base has QualQV 0 and rich QVs 0
                // (a byte is never below a threshold of 0 or less: such a string is not read -- with the reference's defaults,
                // SUBQV 42 and the others unset, that is QUAL and two of the three tags, two thirds of the kernel's loads)
                bool fail = no_qual_fails || (need_qual && (int)qual[q] < thr.qual);
                if (rich) {
                    const uint64_t i = reverse ? l_seq - 1 - q : q;
                    if (need_del) fail = fail || (int)(uint8_t)(dq[i] - 33) < thr.del;
                    if (need_sub) fail = fail || (int)(uint8_t)(sq[i] - 33) < thr.sub;
                    if (need_ins) fail = fail || (int)(uint8_t)(iq[i] - 33) < thr.ins;
                }
                if (from_seq ? fail : gap_fails) c = MSQ_COMPACT_FAIL(c);
            }
            if (live) dst[out0 + p] = c;
        }
        rp0 += __shfl(ri, 63, 64);
        qp0 += __shfl(qi, 63, 64);
        out0 += __shfl(oi, 63, 64);
    }
    // 10^5 waves OR into one word (same-address atomics pass the L2 one by one): the bits are only ever set, so a wave
    // that reads them set has nothing to add
    if (lane == 0 && room > 0) {
        const uint32_t want = 1u | (rich ? 2u : 0u);
        if ((__hip_atomic_load(presence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & want) != want) atomicOr(presence, want);
    }
}

}  // namespace

extern "C" int msq_bam_index_device(msq_ctx* ctx, msq_bam_dev* bam, uint64_t first_record_off, int32_t n_ref, int32_t region_start0,
                                    int32_t region_end0, uint64_t* n_records, uint64_t* n_kept)
{
    if (!ctx || !bam) return msq_fail(MSQ_ERR_ARG, "msq_bam_index_device: NULL argument");
    if (first_record_off > bam->data_bytes) return msq_fail(MSQ_ERR_ARG, "msq_bam_index_device: first record behind the stream");
    if (n_ref < 0) return msq_fail(MSQ_ERR_ARG, "msq_bam_index_device: negative n_ref");
    MSQ_HIP(hipSetDevice(ctx->device));
    for (void** p : {(void**)&bam->d_rec_off, (void**)&bam->d_rec, (void**)&bam->d_kept}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    bam->n_records = bam->n_kept = 0;
    bam->h_kept.clear();
    bam->region_start0 = region_start0;
    bam->region_end0 = region_end0;
    const bool trace = getenv("MSQ_BAM_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double t_chain = 0, t_records = 0;
    const uint64_t span = bam->data_bytes - first_record_off;
    const uint64_t seg = std::max<uint64_t>(64, ctx->bam_segment_bytes);
    const uint64_t n_seg = (span + seg - 1) / seg;
    uint64_t total = 0;
    try {
        std::vector<uint64_t> guess(n_seg), land(n_seg), start(n_seg), rank(n_seg);
        std::vector<uint32_t> count(n_seg);
        void *d_guess = nullptr, *d_land = nullptr, *d_count = nullptr, *d_start = nullptr, *d_rank = nullptr;
        auto release = [&]() {
            for (void* p : {d_guess, d_land, d_count, d_start, d_rank})
                if (p) (void)hipFree(p);
        };
        auto hip_fail = [&](const char* what, hipError_t e) {
            (void)hipStreamSynchronize(ctx->stream);
            release();
            return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "msq_bam_index_device: %s: %s", what, hipGetErrorString(e));
        };
        if (n_seg) {
            hipError_t e = hipMalloc(&d_guess, n_seg * 8);
            if (e == hipSuccess) e = hipMalloc(&d_land, n_seg * 8);
            if (e == hipSuccess) e = hipMalloc(&d_count, n_seg * 4);
            if (e == hipSuccess) e = hipMalloc(&d_start, n_seg * 8);
            if (e == hipSuccess) e = hipMalloc(&d_rank, n_seg * 8);
            if (e != hipSuccess) return hip_fail("hipMalloc", e);
            hipLaunchKernelGGL(bam_chain_kernel, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, ctx->stream, bam->d_data, bam->data_bytes, first_record_off,
                               seg, n_seg, n_ref, (uint64_t*)d_guess, (uint64_t*)d_land, (uint32_t*)d_count);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(guess.data(), d_guess, n_seg * 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(land.data(), d_land, n_seg * 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(count.data(), d_count, n_seg * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return hip_fail("chain kernel", e);
        }
        t_chain = ms_since();
        // the induction: the chain enters segment k at `cur`; the segment's own guess must be that very offset
        uint64_t cur = first_record_off;
        for (uint64_t k = 0; k < n_seg; ++k) {
            const uint64_t seg_end = std::min(first_record_off + (k + 1) * seg, bam->data_bytes);
            start[k] = cur;
            rank[k] = total;
            if (cur >= seg_end) {  // a record spans the whole segment: nothing starts here, whatever the wave guessed
                count[k] = 0;
                continue;
            }
            if (guess[k] != cur || land[k] == NONE) {
                release();
                return msq_fail(MSQ_ERR_DATA, guess[k] != cur ? "msq_bam_index_device: the record chain does not confirm segment %llu's first record"
                                                              : "truncated BAM record (segment %llu)",
                                (unsigned long long)k);
            }
            total += count[k];
            cur = land[k];
        }
        if (cur != bam->data_bytes) {
            release();
            return msq_fail(MSQ_ERR_DATA, "truncated BAM record");
        }
        if (total) {
            hipError_t e = hipMalloc((void**)&bam->d_rec_off, total * 8);
            if (e == hipSuccess) e = hipMalloc((void**)&bam->d_rec, total * sizeof(msq_bam_record));
            if (e == hipSuccess) e = hipMemcpyAsync(d_start, start.data(), n_seg * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_rank, rank.data(), n_seg * 8, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_count, count.data(), n_seg * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return hip_fail("record tables", e);
            hipLaunchKernelGGL(bam_chain_write_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, ctx->stream, bam->d_data, bam->data_bytes,
                               (const uint64_t*)d_start, (const uint64_t*)d_rank, (const uint32_t*)d_count, n_seg, bam->d_rec_off);
            hipLaunchKernelGGL(bam_record_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, ctx->stream, bam->d_data, bam->data_bytes,
                               (const uint64_t*)bam->d_rec_off, total, region_start0, region_end0, bam->d_rec);
            e = hipGetLastError();
            std::vector<msq_bam_record> all(total);
            if (e == hipSuccess) e = hipMemcpyAsync(all.data(), bam->d_rec, total * sizeof(msq_bam_record), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return hip_fail("record kernel", e);
            t_records = ms_since();
            uint64_t n_keep = 0;
            for (uint64_t r = 0; r < total; ++r) {
                const msq_bam_record& k = all[r];
                if (k.status == MSQ_BAM_REC_SKIP) continue;
                ++n_keep;
                if (k.status != MSQ_BAM_REC_KEEP) {
                    release();
                    return msq_fail(MSQ_ERR_DATA, k.status == MSQ_BAM_REC_MALFORMED ? "malformed BAM record (record %llu)"
                                                  : k.status == MSQ_BAM_REC_CIGAR   ? "bad CIGAR (record %llu)"
                                                                                    : "Unexpected cigar (record %llu)",
                                    (unsigned long long)r);
                }
            }
            if (n_keep == total) {  // the table as it came back is the kept table: no second copy of it
                bam->h_kept = std::move(all);
            } else {
                bam->h_kept.reserve(n_keep);
                for (uint64_t r = 0; r < total; ++r)
                    if (all[r].status == MSQ_BAM_REC_KEEP) bam->h_kept.push_back(all[r]);
            }
            if (bam->h_kept.size() == total) {  // every record kept: the table in HBM is the kept table
                bam->d_kept = bam->d_rec;
                bam->d_rec = nullptr;
            } else if (!bam->h_kept.empty()) {
                e = hipMalloc((void**)&bam->d_kept, bam->h_kept.size() * sizeof(msq_bam_record));
                if (e == hipSuccess) e = hipMemcpyAsync(bam->d_kept, bam->h_kept.data(), bam->h_kept.size() * sizeof(msq_bam_record), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) return hip_fail("kept table", e);
            }
        }
        release();
    } catch (const std::bad_alloc&) {
        (void)hipStreamSynchronize(ctx->stream);
        return msq_fail(MSQ_ERR_NOMEM, "msq_bam_index_device: out of memory");
    }
    bam->n_records = total;
    bam->n_kept = bam->h_kept.size();
    if (trace)
        std::fprintf(stderr, "msq_bam_index_device: %llu segments, %llu records, %llu kept: chain %.2f ms, records back at %.2f ms, done at %.2f ms\n",
                     (unsigned long long)n_seg, (unsigned long long)total, (unsigned long long)bam->n_kept, t_chain, t_records, ms_since());
    if (n_records) *n_records = total;
    if (n_kept) *n_kept = bam->n_kept;
    return MSQ_OK;
}

extern "C" int msq_bam_records(msq_ctx* ctx, const msq_bam_dev* bam, msq_bam_record* host_kept)
{
    if (!ctx || !bam || (bam->n_kept && !host_kept)) return msq_fail(MSQ_ERR_ARG, "msq_bam_records: NULL argument");
    if (bam->n_kept) std::memcpy(host_kept, bam->h_kept.data(), bam->n_kept * sizeof(msq_bam_record));
    return MSQ_OK;
}

extern "C" const msq_bam_record* msq_bam_records_view(const msq_bam_dev* bam)
{
    return bam && !bam->h_kept.empty() ? bam->h_kept.data() : nullptr;
}

extern "C" int msq_bam_names(msq_ctx* ctx, const msq_bam_dev* bam, char* host_buf, uint64_t buf_bytes)
{
    if (!ctx || !bam) return msq_fail(MSQ_ERR_ARG, "msq_bam_names: NULL argument");
    const uint64_t n = bam->n_kept;
    if (n == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    const bool trace = getenv("MSQ_BAM_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<uint64_t> off;
    try {
        off.resize(n + 1);
    } catch (const std::bad_alloc&) {
        return msq_fail(MSQ_ERR_NOMEM, "msq_bam_names: out of memory");
    }
    off[0] = 0;
    for (uint64_t r = 0; r < n; ++r) off[r + 1] = off[r] + bam->h_kept[r].l_read_name;
    if (!host_buf || buf_bytes < off[n]) return msq_fail(MSQ_ERR_ARG, "msq_bam_names: the buffer holds %llu bytes, the names take %llu", (unsigned long long)buf_bytes, (unsigned long long)off[n]);
    // offsets and names in the context's scratch (free between the inflate, whose file image it held, and the pack)
    const uint64_t off_bytes = ((n + 1) * 8 + 255) & ~255ull;
    void* scratch = nullptr;
    if (int rc = msq_scratch(ctx, off_bytes + std::max<uint64_t>(off[n], 4), &scratch)) return rc;
    void* const d_off = scratch;
    void* const d_names = (char*)scratch + off_bytes;
    const double t_off = ms_since();
    hipError_t e = hipMemcpyAsync(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bam_names_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, bam->d_data, bam->d_kept, n, (const uint64_t*)d_off, (char*)d_names);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_buf, d_names, off[n], hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "msq_bam_names: %s", hipGetErrorString(e));
    if (trace) std::fprintf(stderr, "msq_bam_names: %llu names, %.1f MB: offsets at %.2f ms, done at %.2f ms\n", (unsigned long long)n, off[n] / 1e6, t_off, ms_since());
    return MSQ_OK;
}

extern "C" int msq_bam_pack_device(msq_ctx* ctx, const msq_bam_dev* bam, uint64_t first, uint64_t count, uint32_t pack_flags,
                                   const msq_qv_thresholds* thr_in, msq_packed_dev** out)
{
    if (!ctx || !bam || !out || !thr_in) return msq_fail(MSQ_ERR_ARG, "msq_bam_pack_device: NULL argument");
    *out = nullptr;
    if (first > bam->n_kept || count > bam->n_kept - first) return msq_fail(MSQ_ERR_ARG, "msq_bam_pack_device: reads outside the file's kept records");
    const msq_qv_thresholds thr = *thr_in;
    for (int k = 0; k < 4; ++k) {
        const int32_t t = (&thr.qual)[k];
        if (t < -1 || t > 255) return msq_fail(MSQ_ERR_ARG, "msq_bam_pack_device: QV threshold %d out of range", t);
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> rs, re;
    std::vector<uint64_t> base_off;
    try {
        rs.resize(count);
        re.resize(count);
        base_off.resize(count + 1);
    } catch (const std::bad_alloc&) {
        return msq_fail(MSQ_ERR_NOMEM, "msq_bam_pack_device: out of memory");
    }
    base_off[0] = 0;
    for (uint64_t r = 0; r < count; ++r) {
        const msq_bam_record& k = bam->h_kept[first + r];
        // source.referenceStart / referenceEnd of BamToReadSource: the record's interval clipped to the region
        rs[r] = std::max(k.ref_start, bam->region_start0);
        re[r] = std::min(k.ref_end, bam->region_end0);
        base_off[r + 1] = base_off[r] + (uint64_t)k.n_bases;
    }
    const uint64_t total = base_off[count];
    CompactRows rows;
    {
        static const char kSeq[] = "=ACMGRSVTWYHKDBN";
        static const char kOps[] = "MIDNSHP=X";
        const uint8_t* table = msq_compact_table();
        for (int op = 0; op < 9; ++op) {
            for (int v = 0; v < 16; ++v) rows.seq[op][v] = table[(uint8_t)kOps[op] | ((unsigned)(uint8_t)kSeq[v] << 8)];
            rows.gap[op] = table[(uint8_t)kOps[op] | ((unsigned)(uint8_t)(op == 6 ? '*' : '-') << 8)];
        }
    }
    // the translated bytes, the base offsets and the presence word live in the context's scratch (the file image that
    // msq_bgzf_inflate_device staged there is long inflated): [0, 256) the packer's own words, then these
    const uint64_t off_bytes = ((count + 1) * 8 + 255) & ~255ull;
    void* scratch = nullptr;
    if (int rc = msq_scratch(ctx, 512 + off_bytes + std::max<uint64_t>(total, 64), &scratch)) return rc;
    char* const d_presence = (char*)scratch + 256;
    char* const d_base_off = (char*)scratch + 512;
    char* const d_compact = d_base_off + off_bytes;
    hipError_t e = hipMemcpyAsync(d_base_off, base_off.data(), (count + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_presence, 0, 4, ctx->stream);
    if (e == hipSuccess && count) {
        // (a wave per read: waves that stride over the reads, eight workgroups per CU, took 1.33 instead of 0.86 ms)
        hipLaunchKernelGGL(bam_decode_kernel, dim3((unsigned)((count + kDecodeWaves - 1) / kDecodeWaves)), dim3(64 * kDecodeWaves), 0, ctx->stream, bam->d_data, bam->d_kept + first, count,
                           (const uint64_t*)d_base_off, bam->region_start0, bam->region_end0, thr, rows, (uint8_t*)d_compact, (uint32_t*)d_presence);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return msq_fail(MSQ_ERR_HIP, "msq_bam_pack_device: %s", hipGetErrorString(e));
    }
    // no wait for the decode: the packer makes its tables and allocates the set's buffers beside it
    const int rc = msq_pack_compact_resident(ctx, count, rs.data(), re.data(), base_off.data(), (const uint64_t*)d_base_off, (const uint8_t*)d_compact,
                                             (const uint32_t*)d_presence, pack_flags, &thr, out);
    (void)hipStreamSynchronize(ctx->stream);  // the scratch may go to its next user
    return rc;
}
