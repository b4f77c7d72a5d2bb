// msq_pileup_stream.hip -- the streaming bit-plane pileup kernel (fast path of msq_pileup).
//
// What it computes is exactly MSAByRow::AddRead + MSAByColumn + CodonsAt for every column
// (src/MSA.cpp:55-92,126-136,144-219); how it computes it is built for gfx950:
//
//  * A persistent workgroup streams whole reads, each packed byte is loaded exactly once, 16-32 B per lane,
//    coalesced (1 KiB of bit planes + 2 KiB per QV plane per wave instruction).
//  * Phase 1 (read space, one wave per read, lane = one 32-base group): the four input planes become five
//    column-space planes V (QV-passing ACGT), B0, B1 (its code), D ('-'), N; insertions / pads are squeezed
//    out of the 32-bit lane words (software pext, rare), the lane's column offset is a wave prefix sum of
//    popcount(column-consuming ops), and the words are OR-ed at that bit offset into the read's LDS row.
//  * Phase 2 (column space, lane = 32 columns, 8 reads per lane and batch): rows are compared against a
//    GUIDE sequence 32 columns per instruction.  Cells that match the guide -- nearly all of them in amplicon
//    data -- are counted in bit-sliced vertical counters held in registers (Harley-Seal carry-save adders);
//    only deviating valid bases / codons (sequencing errors, minor variants) cost a global atomic.
//  * At the end each workgroup stores its match counts as a partial; a small kernel folds the partials into
//    the count matrices under the guide's symbols.
//
// The guide is only a speed hint (consensus of a strided sample of the reads): any guide gives bit-exact
// counts, a bad one only sends more cells down the atomic path.
#include "msq_internal.h"

#include <algorithm>

namespace {

constexpr int kLevels = 16;          // bit-sliced counter depth: 65535 reads per lane between flushes
constexpr int kReadsPerLane = 8;     // one Harley-Seal block per batch
constexpr int kSample = 256;         // reads sampled for the guide
constexpr int kMaxThreads = 512;

enum QvMode : int { QV_NONE = 0, QV_QUAL = 1, QV_ANY = 2 };

struct StreamArgs {
    const int32_t* ref_start;
    const uint64_t* group_off;
    const uint4* bop;
    const uint4* qv[4];  // qual, del, sub, ins planes (two uint4 per group) or nullptr
    int32_t thr[4];      // -1 = unset
    uint64_t n_reads;
    int32_t win_begin0, W;
    int32_t G;           // 32-column groups: ceil(W / 32)
    int32_t RS;          // read slots: lanes that share a column group
    int32_t RD;          // dwords per LDS row (>= G + 1, multiple of 4)
    const uint32_t* guide;  // [3][RD]: valid, code bit0, code bit1
    uint32_t* nuc;
    uint32_t* codon;
    uint32_t* partials;  // [gridDim.x][NC][G*32]
    uint32_t* err;
};

__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int k)  // bits k.. of the 64-bit hi:lo
{
    return __builtin_amdgcn_alignbit(hi, lo, k);
}

// carry-save adder: (a + b + c) -> sum (weight 1), carry (weight 2)
__device__ __forceinline__ void csa(uint32_t a, uint32_t b, uint32_t c, uint32_t& sum, uint32_t& carry)
{
    const uint32_t u = a ^ b;
    sum = u ^ c;
    carry = (a & b) | (u & c);
}

// Add eight 32-column bit masks into a bit-sliced counter (cnt[l] holds bit l of each column's count).
__device__ __forceinline__ void counter_add8(uint32_t (&cnt)[kLevels], const uint32_t (&x)[kReadsPerLane])
{
    uint32_t ones = cnt[0], twos = cnt[1], fours = cnt[2];
    uint32_t t2a, t2b, t4a, t4b, carry;
    csa(ones, x[0], x[1], ones, t2a);
    csa(ones, x[2], x[3], ones, t2b);
    csa(twos, t2a, t2b, twos, t4a);
    csa(ones, x[4], x[5], ones, t2a);
    csa(ones, x[6], x[7], ones, t2b);
    csa(twos, t2a, t2b, twos, t4b);
    csa(fours, t4a, t4b, fours, carry);
    cnt[0] = ones; cnt[1] = twos; cnt[2] = fours;
#pragma unroll
    for (int l = 3; l < kLevels; ++l) {  // ripple the weight-8 carry upwards
        const uint32_t c = cnt[l] & carry;
        cnt[l] ^= carry;
        carry = c;
    }
}

// 32-bit pass mask of one QV plane for one group: bit k = (qv of base k >= threshold).  The plane stores the
// group transposed (MSQ_QV_BYTE), so dword i carries bases i, 8+i, 16+i, 24+i and eight shift-accumulates
// leave the mask in base order.
__device__ __forceinline__ uint32_t qv_pass_mask(const uint4& a, const uint4& b, int thr)
{
    const uint32_t H = 0x80808080u;
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t acc = 0;
    if (thr < 128) {  // wave-uniform
        const uint32_t Y = 0x01010101u * (uint32_t)thr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // per byte: x >= y  <=>  bit 7 of ((x | 0x80) - y) | x   (y <= 127: no borrow crosses a byte)
            const uint32_t d = ((w[i] | H) - Y) | w[i];
            acc = (acc >> 1) | (d & H);
        }
    } else {
        const uint32_t Y = 0x01010101u * (uint32_t)(thr - 128);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // y >= 128: x >= y  <=>  x has bit 7 and (x & 0x7f) >= y - 128
            const uint32_t d = (((w[i] & ~H) | H) - Y) & w[i];
            acc = (acc >> 1) | (d & H);
        }
    }
    return acc;
}

// Remove the bits of `x` at the zero positions of `keep` below keep's top set bit (software pext for the few
// groups that hold an insertion or a pad), for five planes at once.
__device__ __forceinline__ void squeeze5(uint32_t keep, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3, uint32_t& p4)
{
    if (keep == 0) return;
    uint32_t holes = ~keep & (0xffffffffu >> __clz(keep));
    while (holes) {
        const int j = 31 - __clz(holes);  // highest hole first: lower hole positions stay valid
        const uint32_t low = (1u << j) - 1u;
        p0 = (p0 & low) | ((p0 >> 1) & ~low);
        p1 = (p1 & low) | ((p1 >> 1) & ~low);
        p2 = (p2 & low) | ((p2 >> 1) & ~low);
        p3 = (p3 & low) | ((p3 >> 1) & ~low);
        p4 = (p4 & low) | ((p4 >> 1) & ~low);
        holes &= ~(1u << j);
    }
}

__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v)
{
    if (v) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int QV, bool COUNT_N>
__global__ __launch_bounds__(kMaxThreads) void pileup_kernel(StreamArgs a)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = T >> 6;
    const int RS = a.RS, R = kReadsPerLane * RS, RD = a.RD, G = a.G, W = a.W;
    const int plane_stride = R * RD;  // rows of one plane
    uint32_t my_err = 0;

    // ---- phase-2 identity: this lane owns columns [32*cg, 32*cg+32) for reads rs, rs+RS, ...
    const bool p2 = tid < G * RS;
    const int cg = p2 ? tid % G : 0, rs = p2 ? tid / G : 0;
    const uint32_t gv_lo = a.guide[cg], gv_hi = a.guide[cg + 1];
    const uint32_t g0_lo = a.guide[RD + cg], g0_hi = a.guide[RD + cg + 1];
    const uint32_t g1_lo = a.guide[2 * RD + cg], g1_hi = a.guide[2 * RD + cg + 1];
    // columns that can start a codon: 1 <= c <= W-3 (MSARow::CodonAt's i > 0, CodingCodonAt's bound; MSA.cpp:193,207)
    uint32_t codon_cols = 0;
    {
        const int c0 = cg * 32;
        const int lo = max(1 - c0, 0), hi = min(W - 3 - c0, 31);
        if (p2 && lo <= hi) codon_cols = (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo);
    }
    uint32_t cnt_match[kLevels], cnt_gap[kLevels], cnt_codon[kLevels], cnt_n[kLevels];  // cnt_n dead unless COUNT_N
#pragma unroll
    for (int l = 0; l < kLevels; ++l) cnt_match[l] = cnt_gap[l] = cnt_codon[l] = cnt_n[l] = 0;

    const uint64_t n_batches = (a.n_reads + R - 1) / R;
    for (uint64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        // ---- zero the rows (MSARow(size): all ' ')
        for (int i = tid * 4; i < 5 * plane_stride; i += T * 4) *reinterpret_cast<uint4*>(&lds[i]) = make_uint4(0, 0, 0, 0);
        __syncthreads();

        // ---- phase 1: read space -> column-space planes in LDS (MSAByRow::AddRead, MSA.cpp:144-186)
        for (int rr = wave; rr < R; rr += NW) {
            const uint64_t r = batch * R + rr;
            if (r >= a.n_reads) break;
            const uint64_t g0 = a.group_off[r], g1 = a.group_off[r + 1];
            int col = a.ref_start[r] - a.win_begin0;  // MSA.cpp:148
            uint32_t* row = &lds[rr * RD];
            for (uint64_t gb = g0; gb < g1; gb += 64) {
                const uint64_t g = gb + lane;
                const bool have = g < g1;
                uint4 w = make_uint4(0u, 0u, ~0u, ~0u);  // beyond the read: op 3
                if (have) w = a.bop[g];
                uint32_t pass = ~0u;  // ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87
                if (QV != QV_NONE && have) {
#pragma unroll
                    for (int k = 0; k < (QV == QV_QUAL ? 1 : 4); ++k)
                        if (a.thr[k] >= 0 && a.qv[k]) {  // wave-uniform
                            const uint4 qa = a.qv[k][2 * g], qb = a.qv[k][2 * g + 1];
                            pass &= qv_pass_mask(qa, qb, a.thr[k]);
                        }
                }
                const uint32_t cons = ~w.w;          // ops 0, 1 consume a column
                const uint32_t is_m = cons & ~w.z;   // op 0: '=' / 'X'
                const uint32_t op1 = cons & w.z;     // op 1: '-', literal N, exotic
                if (op1 & w.y) my_err |= DEVERR_EXOTIC;  // exotic codes are the simple kernel's job
                uint32_t pv = is_m & pass;                          // row char = nucleotide   (MSA.cpp:163-164)
                uint32_t pn = (is_m & ~pass) | (op1 & w.x & ~w.y);  // row char = 'N'          (MSA.cpp:165-166)
                uint32_t pd = op1 & ~w.x & ~w.y;                    // row char = '-'          (MSA.cpp:168-171)
                uint32_t b0 = w.x & pv, b1 = w.y & pv;
                const int n = __popc(cons);
                if (cons != (n == 32 ? ~0u : ((1u << n) - 1u))) squeeze5(cons, pv, pn, pd, b0, b1);
                // column of this lane's first base: wave-exclusive prefix sum of n
                int incl = n;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d);
                    if (lane >= d) incl += t;
                }
                const int c0 = col + incl - n;
                col += __shfl(incl, 63);
                if (n) {
                    if (c0 < 0 || c0 + n > W) {
                        my_err |= DEVERR_WINDOW;  // assert(pos >= 0) / write past the row, MSA.cpp:149
                    } else {
                        const int word = c0 >> 5, sh = c0 & 31;
                        uint32_t* p = row + word;
                        lds_or(p, pv << sh);
                        lds_or(p + plane_stride, b0 << sh);
                        lds_or(p + 2 * plane_stride, b1 << sh);
                        lds_or(p + 3 * plane_stride, pd << sh);
                        lds_or(p + 4 * plane_stride, pn << sh);
                        if (sh) {
                            const int back = 32 - sh;
                            lds_or(p + 1, pv >> back);
                            lds_or(p + 1 + plane_stride, b0 >> back);
                            lds_or(p + 1 + 2 * plane_stride, b1 >> back);
                            lds_or(p + 1 + 3 * plane_stride, pd >> back);
                            lds_or(p + 1 + 4 * plane_stride, pn >> back);
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ---- phase 2: column space (MSAByColumn ctor MSA.cpp:55-92 + CodonsAt MSA.cpp:126-136,197-219)
        if (p2) {
            uint32_t x_match[kReadsPerLane], x_gap[kReadsPerLane], x_codon[kReadsPerLane], x_n[kReadsPerLane];
#pragma unroll
            for (int i = 0; i < kReadsPerLane; ++i) {
                const uint32_t* rw = &lds[(rs + i * RS) * RD + cg];
                const uint32_t v_lo = rw[0], v_hi = rw[1];
                const uint32_t b0_lo = rw[plane_stride], b0_hi = rw[plane_stride + 1];
                const uint32_t b1_lo = rw[2 * plane_stride], b1_hi = rw[2 * plane_stride + 1];
                const uint32_t d_lo = rw[3 * plane_stride];
                const uint32_t n_lo = rw[4 * plane_stride];
                // valid base equal to the guide base
                const uint32_t m_lo = v_lo & gv_lo & ~((b0_lo ^ g0_lo) | (b1_lo ^ g1_lo));
                const uint32_t m_hi = v_hi & gv_hi & ~((b0_hi ^ g0_hi) | (b1_hi ^ g1_hi));
                // codon at column c = columns c, c+1, c+2
                const uint32_t cm = m_lo & funnel(m_lo, m_hi, 1) & funnel(m_lo, m_hi, 2) & codon_cols;
                const uint32_t cv = v_lo & funnel(v_lo, v_hi, 1) & funnel(v_lo, v_hi, 2) & codon_cols;
                x_match[i] = m_lo;
                x_gap[i] = d_lo;
                x_codon[i] = cm;
                x_n[i] = n_lo;
                // deviations from the guide: one global atomic each (MSAColumn::IncCounts MSA.inl:66; ++codons[codon])
                uint32_t nd = v_lo & ~m_lo;
                const int cbase = cg * 32;
                while (nd) {
                    const int j = __ffs(nd) - 1;
                    nd &= nd - 1;
                    const uint32_t code = ((b0_lo >> j) & 1u) | (((b1_lo >> j) & 1u) << 1);
                    atomicAdd(&a.nuc[(size_t)(cbase + j) * 6 + code], 1u);
                }
                if (!COUNT_N) {
                    uint32_t nn = n_lo;
                    while (nn) {
                        const int j = __ffs(nn) - 1;
                        nn &= nn - 1;
                        atomicAdd(&a.nuc[(size_t)(cbase + j) * 6 + 5], 1u);
                    }
                }
                uint32_t cd = cv & ~cm;
                while (cd) {
                    const int j = __ffs(cd) - 1;
                    cd &= cd - 1;
                    const uint32_t t0 = funnel(b0_lo, b0_hi, j) & 7u, t1 = funnel(b1_lo, b1_hi, j) & 7u;
                    const uint32_t idx = ((t1 & 1u) << 5) | ((t0 & 1u) << 4) | ((t1 & 2u) << 2) | ((t0 & 2u) << 1) |
                                         ((t1 & 4u) >> 1) | ((t0 & 4u) >> 2);
                    atomicAdd(&a.codon[(size_t)(cbase + j) * 64 + idx], 1u);
                }
            }
            counter_add8(cnt_match, x_match);
            counter_add8(cnt_gap, x_gap);
            counter_add8(cnt_codon, x_codon);
            if (COUNT_N) counter_add8(cnt_n, x_n);
        }
        __syncthreads();
    }

    // ---- flush: transpose the bit-sliced counters, reduce the read slots through LDS, store the partial
    const int NC = COUNT_N ? 4 : 3;
    const int cols = G * 32;
    uint32_t* acc = lds;
    for (int k = 0; k < NC; ++k) {
        for (int i = tid; i < cols; i += T) acc[i] = 0;
        __syncthreads();
        if (p2) {
            const uint32_t* cnt = k == 0 ? cnt_match : k == 1 ? cnt_gap : k == 2 ? cnt_codon : cnt_n;
            for (int j = 0; j < 32; ++j) {
                uint32_t v = 0;
#pragma unroll
                for (int l = 0; l < kLevels; ++l) v |= ((cnt[l] >> j) & 1u) << l;
                if (v) atomicAdd(&acc[cg * 32 + j], v);
            }
        }
        __syncthreads();
        uint32_t* dst = a.partials + ((size_t)blockIdx.x * NC + k) * cols;
        for (int i = tid; i < cols; i += T) dst[i] = acc[i];
        __syncthreads();
    }
    if (my_err) atomicOr(a.err, my_err);
}

// Guide planes from the sample's nucleotide counts: the most frequent of A C G T - N per column; only an
// ACGT winner makes a guide base.
__global__ void make_guide_kernel(const uint32_t* __restrict__ sample_nuc, int W, int RD, uint32_t* __restrict__ guide)
{
    const int word = blockIdx.x * blockDim.x + threadIdx.x;
    if (word >= RD) return;
    uint32_t gv = 0, g0 = 0, g1 = 0;
    for (int j = 0; j < 32; ++j) {
        const int c = word * 32 + j;
        if (c >= W) break;
        int best = 0;
        uint32_t bestn = 0;
        for (int t = 0; t < 6; ++t) {
            const uint32_t n = sample_nuc[(size_t)c * 6 + t];
            if (n > bestn) { bestn = n; best = t; }
        }
        if (bestn > 0 && best < 4) {
            gv |= 1u << j;
            g0 |= (uint32_t)(best & 1) << j;
            g1 |= (uint32_t)(best >> 1) << j;
        }
    }
    guide[word] = gv;
    guide[RD + word] = g0;
    guide[2 * RD + word] = g1;
}

// Fold the workgroup partials into the count matrices: matches go to the guide's nucleotide / codon bin,
// gaps to '-', Ns to 'N' (bins 4 and 5, NucleotideConversion.h:69-88).
__global__ __launch_bounds__(256) void fold_partials_kernel(const uint32_t* __restrict__ partials, int n_wg, int NC, int cols,
                                                            int W, int RD, const uint32_t* __restrict__ guide,
                                                            uint32_t* __restrict__ nuc, uint32_t* __restrict__ codon)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    uint32_t s[4] = {0, 0, 0, 0};
    for (int wg = 0; wg < n_wg; ++wg)
        for (int k = 0; k < NC; ++k) s[k] += partials[((size_t)wg * NC + k) * cols + c];
    auto gbit = [&](int plane, int col) -> uint32_t { return (guide[plane * RD + (col >> 5)] >> (col & 31)) & 1u; };
    auto gcode = [&](int col) -> uint32_t { return gbit(1, col) | (gbit(2, col) << 1); };
    if (s[0]) nuc[(size_t)c * 6 + gcode(c)] += s[0];  // s[0] > 0 implies a valid guide base here
    if (s[1]) nuc[(size_t)c * 6 + 4] += s[1];
    if (NC > 3 && s[3]) nuc[(size_t)c * 6 + 5] += s[3];
    if (s[2]) codon[(size_t)c * 64 + ((gcode(c) << 4) | (gcode(c + 1) << 2) | gcode(c + 2))] += s[2];
}

}  // namespace

int msq_pileup_simple_sample(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                             int32_t W, uint64_t n_sample, uint64_t stride, uint32_t* d_nuc, uint32_t* d_codon);

int msq_pileup_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                             int32_t W, uint32_t* d_nuc, uint32_t* d_codon, bool* handled)
{
    *handled = false;
    if (R.flags & MSQ_READS_HAS_EXOTIC) return MSQ_OK;
    const int G = (W + 31) / 32;
    if (G > kMaxThreads) return MSQ_OK;  // wider windows: simple kernel (column super-tiles are future work)
    // geometry: T threads, RS read slots per column group, 8 reads per lane and batch
    int best_T = 0, best_RS = 0;
    double best_score = -1.0;
    for (int T = 128; T <= kMaxThreads; T += 64) {
        if (T < G) continue;
        const int RS = T / G;
        const int RD = ((G + 1 + 3) / 4) * 4;
        const size_t lds = (size_t)5 * kReadsPerLane * RS * RD * 4;
        if (lds > 72 * 1024 && RS > 1) continue;  // keep two workgroups per CU when there is a choice
        if (lds > 150 * 1024) continue;
        const double util = (double)(G * RS) / T;
        const double score = util + 0.02 * RS - (T > 384 ? 0.05 : 0.0);
        if (score > best_score) { best_score = score; best_T = T; best_RS = RS; }
    }
    if (!best_T) return MSQ_OK;
    const int T = best_T, RS = best_RS, RD = ((G + 1 + 3) / 4) * 4, Rb = kReadsPerLane * RS;
    const size_t lds = std::max((size_t)5 * Rb * RD * 4, (size_t)G * 32 * 4);
    const int wg_per_cu = std::max(1, std::min((int)(160 * 1024 / (lds + 512)), 2048 / T));
    const uint64_t n_batches = (R.n_reads + Rb - 1) / Rb;
    const int n_wg = (int)std::min<uint64_t>(n_batches, (uint64_t)ctx->n_cu * wg_per_cu);
    if (kReadsPerLane * ((n_batches + n_wg - 1) / n_wg) > 65535) return MSQ_OK;  // bit-sliced counters are 16 levels deep

    int mode = QV_NONE;
    const bool q_on = thr.qual >= 0 && R.qual;
    const bool rich_on = (thr.del >= 0 && R.delqv) || (thr.sub >= 0 && R.subqv) || (thr.ins >= 0 && R.insqv);
    if (rich_on) mode = QV_ANY;
    else if (q_on) mode = QV_QUAL;
    const int NC = mode == QV_NONE ? 3 : 4;

    // scratch: sample counts [W][70] | guide [3][RD] | partials [n_wg][NC][G*32]
    const size_t sample_bytes = (size_t)W * 70 * 4;
    const size_t guide_off = (sample_bytes + 255) & ~(size_t)255;
    const size_t part_off = (guide_off + (size_t)3 * RD * 4 + 255) & ~(size_t)255;
    const size_t total = part_off + (size_t)n_wg * NC * G * 32 * 4;
    void* s = nullptr;
    int rc = msq_scratch(ctx, total, &s);
    if (rc != MSQ_OK) return rc;
    uint32_t* d_sample = (uint32_t*)s;
    uint32_t* d_guide = (uint32_t*)((char*)s + guide_off);
    uint32_t* d_part = (uint32_t*)((char*)s + part_off);

    // guide from a strided sample of the reads
    MSQ_HIP(hipMemsetAsync(d_sample, 0, sample_bytes, ctx->stream));
    const uint64_t n_sample = std::min<uint64_t>(R.n_reads, kSample);
    const uint64_t stride = std::max<uint64_t>(1, R.n_reads / n_sample);
    rc = msq_pileup_simple_sample(ctx, R, thr, win_begin0, W, n_sample, stride, d_sample, d_sample + (size_t)W * 6);
    if (rc != MSQ_OK) return rc;
    hipLaunchKernelGGL(make_guide_kernel, dim3((RD + 63) / 64), dim3(64), 0, ctx->stream, d_sample, W, RD, d_guide);
    MSQ_HIP(hipGetLastError());

    StreamArgs a;
    a.ref_start = R.ref_start;
    a.group_off = R.group_off;
    a.bop = reinterpret_cast<const uint4*>(R.bop);
    a.qv[0] = reinterpret_cast<const uint4*>(R.qual);
    a.qv[1] = reinterpret_cast<const uint4*>(R.delqv);
    a.qv[2] = reinterpret_cast<const uint4*>(R.subqv);
    a.qv[3] = reinterpret_cast<const uint4*>(R.insqv);
    a.thr[0] = thr.qual; a.thr[1] = thr.del; a.thr[2] = thr.sub; a.thr[3] = thr.ins;
    a.n_reads = R.n_reads;
    a.win_begin0 = win_begin0;
    a.W = W;
    a.G = G; a.RS = RS; a.RD = RD;
    a.guide = d_guide;
    a.nuc = d_nuc;
    a.codon = d_codon;
    a.partials = d_part;
    a.err = ctx->d_err;

    void (*kern)(StreamArgs) = mode == QV_NONE ? pileup_kernel<QV_NONE, false>
                               : mode == QV_QUAL ? pileup_kernel<QV_QUAL, true> : pileup_kernel<QV_ANY, true>;
    if (lds > 48 * 1024)
        MSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    msq_timing_mark_begin(ctx);
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(T), lds, ctx->stream, a);
    msq_timing_mark_end(ctx);
    MSQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(fold_partials_kernel, dim3((W + 255) / 256), dim3(256), 0, ctx->stream, d_part, n_wg, NC, G * 32, W, RD,
                       d_guide, d_nuc, d_codon);
    MSQ_HIP(hipGetLastError());
    *handled = true;
    return MSQ_OK;
}
