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
    uint32_t* nuc;
    uint32_t* codon;
    uint32_t* partials;  // [gridDim.x][NC + 1][G*32]
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

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (no LDS traffic).
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}

// bit-sliced compare: 2 * a > b, per column (a, b: kLevels-deep vertical counters)
__device__ __forceinline__ uint32_t sliced_twice_gt(const uint32_t (&a)[kLevels], const uint32_t (&b)[kLevels])
{
    uint32_t gt = a[kLevels - 1], eq = ~a[kLevels - 1];  // level kLevels of 2a vs 0
#pragma unroll
    for (int l = kLevels - 1; l >= 0; --l) {
        const uint32_t x = l > 0 ? a[l - 1] : 0u;  // level l of 2a
        gt |= eq & x & ~b[l];
        eq &= ~(x ^ b[l]);
    }
    return gt;
}

template <int NP>
struct ChunkData {
    uint4 w;
    uint4 qa[NP ? NP : 1], qb[NP ? NP : 1];
};
struct ChunkItem {
    int rr;        // read slot in the batch; >= R: no more work
    int first;     // first chunk of its read
    uint64_t gb;   // first group of the chunk
    uint64_t g1;   // end group of the read
};

template <int QV, bool COUNT_N>
__global__ __launch_bounds__(kMaxThreads) void pileup_kernel(StreamArgs a)
{
    constexpr int NP = QV == QV_NONE ? 0 : QV == QV_QUAL ? 1 : 4;  // QV planes looked at
    constexpr int DEPTH = QV == QV_ANY ? 1 : 2;                    // chunks in flight ahead of the one being decoded
    extern __shared__ __align__(16) uint32_t lds[];
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = T >> 6;
    const int RS = a.RS, R = kReadsPerLane * RS, RD = a.RD, G = a.G, W = a.W;
    const int plane_stride = R * RD;  // rows of one plane
    uint32_t* const gl = lds + 5 * plane_stride;                               // guide words [3][RD]
    uint64_t* const go = reinterpret_cast<uint64_t*>(gl + 3 * RD + (RD & 1));  // group_off of the batch [R+1]
    int32_t* const rstart = reinterpret_cast<int32_t*>(go + R + 1);            // ref_start of the batch [R]
    uint32_t my_err = 0;

    // ---- phase-2 identity: this lane owns columns [32*cg, 32*cg+32) for reads rs, rs+RS, ...
    const bool p2 = tid < G * RS;
    const int cg = p2 ? tid % G : 0, rs = p2 ? tid / G : 0;
    uint32_t gv_lo = 0, gv_hi = 0, g0_lo = 0, g0_hi = 0, g1_lo = 0, g1_hi = 0;
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
    for (int i = tid; i < 3 * RD; i += T) gl[i] = 0;

    const uint64_t n_batches = (a.n_reads + R - 1) / R;
    bool learn = true;  // the first batch of this workgroup also teaches it its guide
    for (uint64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        // ---- zero the rows (MSARow(size): all ' '), fetch the batch's read table
        for (int i = tid * 4; i < 5 * plane_stride; i += T * 4) *reinterpret_cast<uint4*>(&lds[i]) = make_uint4(0, 0, 0, 0);
        for (int i = tid; i <= R; i += T) {
            const uint64_t r = min(batch * R + (uint64_t)i, a.n_reads);
            go[i] = a.group_off[r];
            if (i < R) rstart[i] = r < a.n_reads ? a.ref_start[r] : 0;
        }
        __syncthreads();

        // ---- phase 1: read space -> column-space planes in LDS (MSAByRow::AddRead, MSA.cpp:144-186).
        // One wave per read, one lane per 32-base group; the wave's chunks form one software-pipelined stream
        // with DEPTH chunk loads in flight ahead of the chunk being decoded.
        auto seek_read = [&](int rr) -> ChunkItem {
            ChunkItem it;
            it.rr = R; it.first = 1; it.gb = 0; it.g1 = 0;
            while (rr < R) {
                const uint64_t g0 = go[rr], g1 = go[rr + 1];
                if (g1 > g0) { it.rr = rr; it.gb = g0; it.g1 = g1; break; }
                rr += NW;
            }
            return it;
        };
        auto next_item = [&](const ChunkItem& it) -> ChunkItem {
            if (it.rr >= R) return it;
            if (it.gb + 64 < it.g1) {
                ChunkItem n = it;
                n.gb += 64;
                n.first = 0;
                return n;
            }
            return seek_read(it.rr + NW);
        };
        auto load_chunk = [&](const ChunkItem& it) -> ChunkData<NP> {
            ChunkData<NP> d;
            d.w = make_uint4(0u, 0u, ~0u, ~0u);  // beyond the read: op 3
            const uint64_t g = it.gb + lane;
            const bool have = it.rr < R && g < it.g1;
            if (have) d.w = a.bop[g];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                d.qa[k] = d.qb[k] = make_uint4(0, 0, 0, 0);
                if (have && a.thr[k] >= 0 && a.qv[k]) {
                    d.qa[k] = a.qv[k][2 * g];
                    d.qb[k] = a.qv[k][2 * g + 1];
                }
            }
            return d;
        };

        ChunkItem items[DEPTH + 1];
        ChunkData<NP> data[DEPTH + 1];
        items[0] = seek_read(wave);
        data[0] = load_chunk(items[0]);
#pragma unroll
        for (int q = 1; q <= DEPTH; ++q) {
            items[q] = next_item(items[q - 1]);
            data[q] = load_chunk(items[q]);
        }
        int col = 0;
        while (items[0].rr < R) {
            const ChunkItem it = items[0];
            const uint4 w = data[0].w;
            uint32_t pass = ~0u;  // ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87
#pragma unroll
            for (int k = 0; k < NP; ++k)
                if (a.thr[k] >= 0 && a.qv[k]) pass &= qv_pass_mask(data[0].qa[k], data[0].qb[k], a.thr[k]);  // wave-uniform
            // rotate the pipeline and put the next chunk's loads in flight before decoding this one
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                items[q] = items[q + 1];
                data[q] = data[q + 1];
            }
            items[DEPTH] = next_item(items[DEPTH - 1]);
            data[DEPTH] = load_chunk(items[DEPTH]);

            if (it.first) col = rstart[it.rr] - a.win_begin0;  // MSA.cpp:148
            uint32_t* row = &lds[it.rr * RD];
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
            const int incl = wave_inclusive_scan(n);
            const int c0 = col + incl - n;
            col += __builtin_amdgcn_readlane(incl, 63);
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
        __syncthreads();

        // ---- first batch only: learn the guide = per-column majority base of this batch
        if (learn) {
            learn = false;
            if (p2 && rs == 0) {
                uint32_t cv[kLevels], c0[kLevels], c1[kLevels];
#pragma unroll
                for (int l = 0; l < kLevels; ++l) cv[l] = c0[l] = c1[l] = 0;
                for (int i0 = 0; i0 < R; i0 += kReadsPerLane) {
                    uint32_t xv[kReadsPerLane], x0[kReadsPerLane], x1[kReadsPerLane];
#pragma unroll
                    for (int i = 0; i < kReadsPerLane; ++i) {
                        const uint32_t* rw = &lds[(i0 + i) * RD + cg];
                        xv[i] = rw[0];
                        x0[i] = rw[plane_stride];
                        x1[i] = rw[2 * plane_stride];
                    }
                    counter_add8(cv, xv);
                    counter_add8(c0, x0);
                    counter_add8(c1, x1);
                }
                uint32_t any = 0;
#pragma unroll
                for (int l = 0; l < kLevels; ++l) any |= cv[l];
                gl[cg] = any;
                gl[RD + cg] = any & sliced_twice_gt(c0, cv);
                gl[2 * RD + cg] = any & sliced_twice_gt(c1, cv);
            }
            __syncthreads();
            if (p2) {
                gv_lo = gl[cg]; gv_hi = gl[cg + 1];
                g0_lo = gl[RD + cg]; g0_hi = gl[RD + cg + 1];
                g1_lo = gl[2 * RD + cg]; g1_hi = gl[2 * RD + cg + 1];
            }
        }

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

    // ---- flush: per column the guide's nucleotide / codon bins (meta), then the bit-sliced counters transposed
    // and reduced over the read slots through LDS, stored as this workgroup's partial
    constexpr int NC = COUNT_N ? 4 : 3;
    const int cols = G * 32;
    uint32_t* const part = a.partials + (size_t)blockIdx.x * (NC + 1) * cols;
    auto gbit = [&](int plane, int c) -> uint32_t { return (gl[plane * RD + (c >> 5)] >> (c & 31)) & 1u; };
    for (int c = tid; c < cols; c += T) {
        const uint32_t v0 = gbit(0, c), v1 = gbit(0, c + 1), v2 = gbit(0, c + 2);
        const uint32_t k0 = gbit(1, c) | (gbit(2, c) << 1), k1 = gbit(1, c + 1) | (gbit(2, c + 1) << 1),
                       k2 = gbit(1, c + 2) | (gbit(2, c + 2) << 1);
        part[(size_t)NC * cols + c] = k0 | (v0 << 2) | (((k0 << 4) | (k1 << 2) | k2) << 8) | ((v0 & v1 & v2) << 14);
    }
    __syncthreads();
    uint32_t* const acc = lds;
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
        for (int i = tid; i < cols; i += T) part[(size_t)k * cols + i] = acc[i];
        __syncthreads();
    }
    if (my_err) atomicOr(a.err, my_err);
}

// Fold the workgroup partials into the count matrices: matches go to each workgroup's guide nucleotide / codon
// bin, gaps to '-', Ns to 'N' (bins 4 and 5, NucleotideConversion.h:69-88).  Block = 64 columns x 4 workgroup
// lanes; blockIdx.y slices the workgroups further.
__global__ __launch_bounds__(256) void fold_partials_kernel(const uint32_t* __restrict__ partials, int n_wg, int NC, int cols,
                                                            int W, uint32_t* __restrict__ nuc, uint32_t* __restrict__ codon)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const int per = (n_wg + gridDim.y - 1) / gridDim.y;
    const int wg_lo = blockIdx.y * per, wg_hi = min(n_wg, wg_lo + per);
    if (c >= W) return;
    uint32_t base[4] = {0, 0, 0, 0}, gap = 0, nn = 0, cod_sum = 0;
    int cod_idx = -1;
    for (int wg = wg_lo + sub; wg < wg_hi; wg += 4) {
        const uint32_t* p = partials + (size_t)wg * (NC + 1) * cols;
        const uint32_t meta = p[(size_t)NC * cols + c];
        const uint32_t m = p[c], g = p[(size_t)cols + c], cm = p[(size_t)2 * cols + c];
        if (m) {  // m > 0 implies a valid guide base
#pragma unroll
            for (int k = 0; k < 4; ++k) base[k] += (meta & 3u) == (uint32_t)k ? m : 0u;
        }
        gap += g;
        if (NC > 3) nn += p[(size_t)3 * cols + c];
        if (cm) {
            const int idx = (int)((meta >> 8) & 63u);
            if (idx != cod_idx) {
                if (cod_sum) atomicAdd(&codon[(size_t)c * 64 + cod_idx], cod_sum);
                cod_idx = idx;
                cod_sum = 0;
            }
            cod_sum += cm;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base[k]) atomicAdd(&nuc[(size_t)c * 6 + k], base[k]);
    if (gap) atomicAdd(&nuc[(size_t)c * 6 + 4], gap);
    if (nn) atomicAdd(&nuc[(size_t)c * 6 + 5], nn);
    if (cod_sum) atomicAdd(&codon[(size_t)c * 64 + cod_idx], cod_sum);
}

}  // namespace

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
    // rows [5][Rb][RD] | guide [3][RD] (+pad) | group_off [Rb+1] u64 | ref_start [Rb]
    const size_t lds = (size_t)5 * Rb * RD * 4 + (size_t)(3 * RD + 1) * 4 + (size_t)(Rb + 1) * 8 + (size_t)Rb * 4 + 16;
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

    // scratch: partials [n_wg][NC + 1][G*32] (counts per counter, then the workgroup's guide bins per column)
    void* s = nullptr;
    int rc = msq_scratch(ctx, (size_t)n_wg * (NC + 1) * G * 32 * 4, &s);
    if (rc != MSQ_OK) return rc;
    uint32_t* d_part = (uint32_t*)s;

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
    const int col_blocks = (W + 63) / 64;
    const int slices = std::max(1, std::min(n_wg / 8, (2 * ctx->n_cu + col_blocks - 1) / col_blocks));
    hipLaunchKernelGGL(fold_partials_kernel, dim3(col_blocks, slices), dim3(256), 0, ctx->stream, d_part, n_wg, NC, G * 32, W,
                       d_nuc, d_codon);
    MSQ_HIP(hipGetLastError());
    *handled = true;
    return MSQ_OK;
}
