// msq_pileup_stream.hip -- the streaming bit-plane pileup kernel (fast path of msq_pileup).
//
// What it computes is exactly MSAByRow::AddRead + MSAByColumn + CodonsAt for every column
// (src/MSA.cpp:55-92,126-136,144-219); how it computes it is built for gfx950:
//
//  * Every WAVE is an autonomous worker that streams whole reads, eight at a time (a "wave batch"), through a
//    private slice of LDS; there is no workgroup barrier in the steady state, so chunk loads stay in flight
//    across every phase.  Each packed byte is loaded exactly once, 16 B per lane and plane, coalesced
//    (1 KiB of bit planes + 2 KiB per QV plane per wave instruction), DEPTH chunks ahead of the decode.
//  * Phase 1 (read space, lane = one 32-base group): the four input planes become four column-space planes
//    V (QV-passing ACGT), X ('-' or 'N'), B0, B1 (nucleotide code; for X: B0 = 1 means 'N'); insertions / pads
//    are squeezed out of the 32-bit lane words (software pext, rare), the lane's column offset is a DPP prefix
//    sum of popcount(column-consuming ops), and the words are OR-ed at that bit offset into the read's LDS row.
//  * Phase 2 (column space, lane = 32 columns, the wave's 8 rows): rows are compared against a GUIDE sequence
//    32 columns per instruction.  Cells that match the guide -- nearly all of them in amplicon data -- are
//    counted in bit-sliced vertical counters held in registers (Harley-Seal carry-save adders); only
//    deviating valid bases / codons (sequencing errors, minor variants) cost a global atomic.
//  * At the end each workgroup reduces its waves' counters through LDS and stores one partial; a small kernel
//    folds the partials into the count matrices under each workgroup's guide symbols.
//
// The guide is only a speed hint (per-column majority base of the workgroup's first reads): any guide gives
// bit-exact counts, a bad one only sends more cells down the atomic path.
#include "msq_internal.h"

#include <algorithm>

namespace {

constexpr int kLevels = 10;       // bit-sliced counter depth: 1023 reads per wave between flushes
constexpr int kRows = 8;          // reads per wave batch = one Harley-Seal block
enum QvMode : int { QV_NONE = 0, QV_QUAL = 1, QV_ANY = 2 };

// Waves per workgroup (one workgroup per CU): 12 = 3 per SIMD where the kernel fits 168 VGPRs, else 8 = 2 per SIMD.
constexpr int max_waves_of(int qv, int passes) { return (qv == QV_NONE && passes == 1) ? 12 : 8; }
constexpr int kMaxBatchesPerWave = ((1 << kLevels) - 1) / kRows;

struct StreamArgs {
    const int32_t* ref_start;
    const uint64_t* group_off;
    const uint4* bop;
    const uint4* qv[4];  // qual, del, sub, ins planes (two uint4 per group) or nullptr
    int32_t thr[4];      // -1 = unset
    uint64_t n_reads;
    uint64_t n_batches;  // ceil(n_reads / 8)
    int32_t win_begin0, W;
    int32_t G;           // 32-column groups: ceil(W / 32)
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
template <int L>
__device__ __forceinline__ void counter_add8(uint32_t (&cnt)[L], const uint32_t (&x)[kRows])
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
    for (int l = 3; l < L; ++l) {  // ripple the weight-8 carry upwards
        const uint32_t c = cnt[l] & carry;
        cnt[l] ^= carry;
        carry = c;
    }
}

// Add four 32-column bit masks into a bit-sliced counter.
template <int L>
__device__ __forceinline__ void counter_add4(uint32_t (&cnt)[L], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    uint32_t ones = cnt[0], twos = cnt[1];
    uint32_t t2a, t2b, carry;
    csa(ones, x0, x1, ones, t2a);
    csa(ones, x2, x3, ones, t2b);
    csa(twos, t2a, t2b, twos, carry);
    cnt[0] = ones; cnt[1] = twos;
#pragma unroll
    for (int l = 2; l < L; ++l) {  // ripple the weight-4 carry upwards
        const uint32_t c = cnt[l] & carry;
        cnt[l] ^= carry;
        carry = c;
    }
}

// bit-sliced compare: 2 * a > b, per column
template <int L>
__device__ __forceinline__ uint32_t sliced_twice_gt(const uint32_t (&a)[L], const uint32_t (&b)[L])
{
    uint32_t gt = a[L - 1], eq = ~a[L - 1];  // level L of 2a vs 0
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        const uint32_t x = l > 0 ? a[l - 1] : 0u;  // level l of 2a
        gt |= eq & x & ~b[l];
        eq &= ~(x ^ b[l]);
    }
    return gt;
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

// Remove the bits at the zero positions of `keep` below keep's top set bit (software pext for the few groups
// that hold an insertion or a pad), for four planes at once.
__device__ __forceinline__ void squeeze4(uint32_t keep, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3)
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
        holes &= ~(1u << j);
    }
}

__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v)
{
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Per-column majority base over every wave's first batch -> the workgroup's guide words gl[3][RD] (valid,
// code bit 0, code bit 1).  Runs once per workgroup; kept out of line so its registers do not add to the
// streaming loop's budget.
__device__ __noinline__ void learn_guide_words(const uint32_t* lds, uint32_t* gl, int NWV, int PS, int RD, int G, int tid, int T)
{
    for (int cg = tid; cg < G; cg += T) {
        uint32_t cv[8], c0[8], c1[8];  // 8 levels: up to 255 rows (12 waves x 8 = 96)
#pragma unroll
        for (int l = 0; l < 8; ++l) cv[l] = c0[l] = c1[l] = 0;
        for (int wv = 0; wv < NWV; ++wv) {
            uint32_t xv[kRows], x0[kRows], x1[kRows];
#pragma unroll
            for (int i = 0; i < kRows; ++i) {
                const uint32_t* rw = lds + wv * 4 * PS + i * RD + cg;
                xv[i] = rw[0];
                x0[i] = rw[2 * PS] & xv[i];  // the X rows reuse B0 for 'N': only valid bases vote
                x1[i] = rw[3 * PS];
            }
            counter_add8(cv, xv);
            counter_add8(c0, x0);
            counter_add8(c1, x1);
        }
        uint32_t any = 0;
#pragma unroll
        for (int l = 0; l < 8; ++l) any |= cv[l];
        gl[cg] = any;
        gl[RD + cg] = any & sliced_twice_gt(c0, cv);
        gl[2 * RD + cg] = any & sliced_twice_gt(c1, cv);
    }
}

template <int NP>
struct ChunkData {
    uint4 w;
    uint4 qa[NP ? NP : 1], qb[NP ? NP : 1];
};

// One chunk (<= 64 groups of one read) travelling through the load pipeline; the fields are wave-uniform.
struct ChunkMeta {
    uint32_t bits;    // bit 0 valid (0: the stream has ended), bit 1 first chunk of its read, bits 2-4 read slot
                      // 0..7 in the wave batch, bits 5.. sequence number of its wave batch
    int32_t rstart;   // ref_start of the read
    __device__ uint32_t valid() const { return bits & 1u; }
    __device__ uint32_t first() const { return bits & 2u; }
    __device__ uint32_t row() const { return (bits >> 2) & 7u; }
    __device__ uint32_t bseq() const { return bits >> 5; }
};

__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

// constant address space + uniform index = s_load (the arrays are read-only for the whole launch)
typedef const uint64_t __attribute__((address_space(4))) * const_u64_ptr;
typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;

// Producer side of one wave's chunk stream: wave batches b = wid, wid + TW, ...; reads r = 8b + i, each cut
// into chunks of <= 64 groups.  Every field is wave-uniform (kept in SGPRs through readfirstlane); the
// (group_off, ref_start) entries of the NEXT read are fetched with scalar loads one read ahead, so the stream
// never waits on a vector load for its own bookkeeping.
struct ReadStream {
    const_u64_ptr group_off;
    const_i32_ptr ref_start;
    uint64_t n_batches, n_reads, TW;
    // the read being streamed
    uint64_t gb, g1;   // next group to load, end group
    int32_t rs;        // its ref_start
    uint32_t valid, first, bseq, row;
    // the prefetched next read
    uint64_t nb, n_g0, n_g1;
    int32_t n_rs;
    uint32_t n_row, n_bseq;

    __device__ __forceinline__ void fetch_next()
    {
        n_g0 = n_g1 = 0;
        n_rs = 0;
        if (nb < n_batches) {
            const uint64_t r = nb * kRows + n_row;
            if (r < n_reads) {
                n_g0 = uni64(group_off[r]);
                n_g1 = uni64(group_off[r + 1]);
                n_rs = (int32_t)uni((uint32_t)ref_start[r]);
            }
        }
    }
    __device__ __forceinline__ void seek()  // move to the next non-empty read of the stream
    {
        while (true) {
            if (nb >= n_batches) {
                valid = 0;
                break;
            }
            const uint64_t f0 = n_g0, f1 = n_g1;
            rs = n_rs;
            row = n_row;
            bseq = n_bseq;
            if (++n_row == (uint32_t)kRows) {
                n_row = 0;
                nb = uni64(nb + TW);
                ++n_bseq;
            }
            fetch_next();
            if (f1 > f0) {
                gb = f0;
                g1 = f1;
                first = 1;
                break;
            }
        }
    }
    __device__ __forceinline__ void init(const uint64_t* go, const int32_t* rstart, uint64_t nbatches, uint64_t nreads,
                                         uint64_t wid, uint64_t total_waves)
    {
        group_off = (const_u64_ptr)(uintptr_t)go;
        ref_start = (const_i32_ptr)(uintptr_t)rstart;
        n_batches = nbatches;
        n_reads = nreads;
        TW = uni64(total_waves);
        gb = g1 = 0;
        rs = 0;
        valid = 1;
        first = 1;
        bseq = row = 0;
        nb = uni64(wid);
        n_row = 0;
        n_bseq = 0;
        fetch_next();
        seek();
    }
};

template <int QV, int PASSES>
__global__ __launch_bounds__(max_waves_of(QV, PASSES) * 64) void pileup_kernel(StreamArgs a)
{
    constexpr int NP = QV == QV_NONE ? 0 : QV == QV_QUAL ? 1 : 4;  // QV planes looked at
    constexpr int DEPTH = QV == QV_ANY ? 1 : 3;                    // chunk loads in flight ahead of the decode
    constexpr bool COUNT_N = QV != QV_NONE;                        // Ns are common only under a QV mask
    constexpr int NC = COUNT_N ? 4 : 3;
    extern __shared__ __align__(16) uint32_t lds[];
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, NWV = T >> 6;
    const int wave = (int)uni((uint32_t)tid >> 6);   // tell the compiler it is wave-uniform: all stream control is scalar
    const int RD = a.RD, G = a.G, W = a.W;
    const int PS = kRows * RD;                       // one plane of one wave's rows
    uint32_t* const wrow = lds + wave * 4 * PS;      // this wave's rows: [4 planes][8 rows][RD]
    uint32_t* const gl = lds + NWV * 4 * PS;         // the workgroup's guide words [3][RD]
    uint32_t my_err = 0;

    // ---- phase-2 identity: in pass p this lane owns columns [32*cg, 32*cg+32), cg = 64*p + lane
    uint32_t gv_lo[PASSES], gv_hi[PASSES], g0_lo[PASSES], g0_hi[PASSES], g1_lo[PASSES], g1_hi[PASSES], codon_cols[PASSES];
    uint32_t cnt[PASSES][NC][kLevels];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        gv_lo[p] = gv_hi[p] = g0_lo[p] = g0_hi[p] = g1_lo[p] = g1_hi[p] = 0;
        // columns that can start a codon: 1 <= c <= W-3 (MSARow::CodonAt's i > 0, CodingCodonAt's bound; MSA.cpp:193,207)
        const int c0 = (p * 64 + lane) * 32;
        const int lo = max(1 - c0, 0), hi = min(W - 3 - c0, 31);
        codon_cols[p] = (p * 64 + lane < G && lo <= hi) ? (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo) : 0u;
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int l = 0; l < kLevels; ++l) cnt[p][k][l] = 0;
    }

    auto zero_rows = [&]() {  // MSARow(size): all ' '
        for (int i = lane * 4; i < 4 * PS; i += 256) *reinterpret_cast<uint4*>(&wrow[i]) = make_uint4(0, 0, 0, 0);
    };
    zero_rows();
    for (int i = tid; i < 3 * RD; i += T) gl[i] = 0;

    // ---- producer side of the chunk stream (all scalar): see ReadStream
    ReadStream st;
    st.init(a.group_off, a.ref_start, a.n_batches, a.n_reads, (uint64_t)blockIdx.x * NWV + wave, (uint64_t)gridDim.x * NWV);
    auto produce = [&](ChunkMeta& m, ChunkData<NP>& d) {
        m.bits = uni(st.valid | (st.first << 1) | ((st.row & 7u) << 2) | (st.bseq << 5));
        m.rstart = st.rs;
        d.w = make_uint4(0u, 0u, ~0u, ~0u);  // beyond the read: op 3
#pragma unroll
        for (int k = 0; k < NP; ++k) d.qa[k] = d.qb[k] = make_uint4(0, 0, 0, 0);
        if (st.valid) {
            const uint32_t ngroups = (uint32_t)min(st.g1 - st.gb, (uint64_t)64);
            if ((uint32_t)lane < ngroups) {
                const uint64_t g = st.gb + lane;
                d.w = a.bop[g];
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (a.thr[k] >= 0 && a.qv[k]) {  // wave-uniform
                        d.qa[k] = a.qv[k][2 * g];
                        d.qb[k] = a.qv[k][2 * g + 1];
                    }
            }
            st.first = 0;
            st.gb = uni64(st.gb + 64);
            if (st.gb >= st.g1) st.seek();
        }
    };

    // ---- phase 1 for one chunk: read space -> column-space planes in LDS (MSAByRow::AddRead, MSA.cpp:144-186)
    int col = 0;
    auto decode = [&](const ChunkMeta& m, const ChunkData<NP>& d) {
        const uint4 w = d.w;
        uint32_t pass = ~0u;  // ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (a.thr[k] >= 0 && a.qv[k]) pass &= qv_pass_mask(d.qa[k], d.qb[k], a.thr[k]);  // wave-uniform
        if (m.first()) col = m.rstart - a.win_begin0;  // MSA.cpp:148
        const uint32_t cons = ~w.w;                 // ops 0, 1 consume a column
        const uint32_t is_m = cons & ~w.z;          // op 0: '=' / 'X'
        const uint32_t op1 = cons & w.z;            // op 1: '-', literal N, exotic
        if (op1 & w.y) my_err |= DEVERR_EXOTIC;     // exotic codes are the simple kernel's job
        const uint32_t masked = is_m & ~pass;       // row char = 'N'          (MSA.cpp:165-166)
        uint32_t pv = is_m & pass;                  // row char = nucleotide   (MSA.cpp:163-164)
        uint32_t px = op1 | masked;                 // row char = '-' (MSA.cpp:168-171) or 'N'
        uint32_t b0 = (w.x & (pv | op1)) | masked;  // code bit 0; for X: 1 = 'N' (literal N is op 1 code 1)
        uint32_t b1 = w.y & pv;
        const int n = __popc(cons);
        if (cons != (n == 32 ? ~0u : ((1u << n) - 1u))) squeeze4(cons, pv, px, b0, b1);
        // column of this lane's first base: wave-exclusive prefix sum of n
        const int incl = wave_inclusive_scan(n);
        const int c0 = col + incl - n;
        col += __builtin_amdgcn_readlane(incl, 63);
        if (n) {
            if (c0 < 0 || c0 + n > W) {
                my_err |= DEVERR_WINDOW;  // assert(pos >= 0) / write past the row, MSA.cpp:149
            } else {
                const int sh = c0 & 31;
                uint32_t* p = wrow + m.row() * RD + (c0 >> 5);
                const uint64_t sv = (uint64_t)pv << sh, sx = (uint64_t)px << sh, s0 = (uint64_t)b0 << sh,
                               s1 = (uint64_t)b1 << sh;
                lds_or(p, (uint32_t)sv);
                lds_or(p + 1, (uint32_t)(sv >> 32));
                lds_or(p + PS, (uint32_t)sx);
                lds_or(p + PS + 1, (uint32_t)(sx >> 32));
                lds_or(p + 2 * PS, (uint32_t)s0);
                lds_or(p + 2 * PS + 1, (uint32_t)(s0 >> 32));
                lds_or(p + 3 * PS, (uint32_t)s1);
                lds_or(p + 3 * PS + 1, (uint32_t)(s1 >> 32));
            }
        }
    };

    // ---- phase 2 for the wave's 8 rows: column space (MSAByColumn ctor MSA.cpp:55-92 + CodonsAt MSA.cpp:126-136,197-219)
    auto phase2 = [&]() {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cg = p * 64 + lane;
            if (cg < G) {
                const int cbase = cg * 32;
#pragma unroll 1
                for (int half = 0; half < kRows; half += 4) {  // rolled: bounds the live registers
                uint32_t x_match[4], x_gap[4], x_codon[4], x_n[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t* rw = wrow + (half + i) * RD + cg;
                    const uint32_t v_lo = rw[0], v_hi = rw[1];
                    const uint32_t xx = rw[PS];
                    const uint32_t b0_lo = rw[2 * PS], b0_hi = rw[2 * PS + 1];
                    const uint32_t b1_lo = rw[3 * PS], b1_hi = rw[3 * PS + 1];
                    // valid base equal to the guide base
                    const uint32_t m_lo = v_lo & gv_lo[p] & ~((b0_lo ^ g0_lo[p]) | (b1_lo ^ g1_lo[p]));
                    const uint32_t m_hi = v_hi & gv_hi[p] & ~((b0_hi ^ g0_hi[p]) | (b1_hi ^ g1_hi[p]));
                    // codon at column c = columns c, c+1, c+2
                    const uint32_t cm = m_lo & funnel(m_lo, m_hi, 1) & funnel(m_lo, m_hi, 2) & codon_cols[p];
                    const uint32_t cv = v_lo & funnel(v_lo, v_hi, 1) & funnel(v_lo, v_hi, 2) & codon_cols[p];
                    const uint32_t n_lo = xx & b0_lo;
                    x_match[i] = m_lo;
                    x_gap[i] = xx & ~b0_lo;
                    x_codon[i] = cm;
                    x_n[i] = n_lo;
                    // deviations from the guide: one global atomic each (MSAColumn::IncCounts MSA.inl:66; ++codons[codon])
                    uint32_t nd = v_lo & ~m_lo;
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
                counter_add4(cnt[p][0], x_match[0], x_match[1], x_match[2], x_match[3]);
                counter_add4(cnt[p][1], x_gap[0], x_gap[1], x_gap[2], x_gap[3]);
                counter_add4(cnt[p][2], x_codon[0], x_codon[1], x_codon[2], x_codon[3]);
                if (COUNT_N) counter_add4(cnt[p][3], x_n[0], x_n[1], x_n[2], x_n[3]);
                }
            }
        }
    };

    // ---- the workgroup's guide = per-column majority base over every wave's first batch
    auto learn_guide = [&]() {
        __syncthreads();  // every wave's first batch is in its rows
        learn_guide_words(lds, gl, NWV, PS, RD, G, tid, T);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cg = p * 64 + lane;
            if (cg < G) {
                gv_lo[p] = gl[cg]; gv_hi[p] = gl[cg + 1];
                g0_lo[p] = gl[RD + cg]; g0_hi[p] = gl[RD + cg + 1];
                g1_lo[p] = gl[2 * RD + cg]; g1_hi[p] = gl[2 * RD + cg + 1];
            }
        }
    };

    // ---- the stream: DEPTH chunk loads stay in flight ahead of the chunk being decoded
    ChunkMeta meta[DEPTH + 1];
    ChunkData<NP> data[DEPTH + 1];
#pragma unroll
    for (int q = 0; q <= DEPTH; ++q) produce(meta[q], data[q]);
    bool guide_ready = false, have_rows = false;
    uint32_t cur_bseq = 0;
    while (true) {
        const ChunkMeta m = meta[0];
        const ChunkData<NP> d = data[0];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            meta[q] = meta[q + 1];
            data[q] = data[q + 1];
        }
        produce(meta[DEPTH], data[DEPTH]);  // next loads go out before this chunk is decoded
        if (!m.valid() || m.bseq() != cur_bseq) {  // a wave batch is complete
            if (!guide_ready) {
                learn_guide();
                guide_ready = true;
            }
            if (have_rows) {
                phase2();
                zero_rows();
                have_rows = false;
            }
            if (!m.valid()) break;
            cur_bseq = m.bseq();
        }
        decode(m, d);
        have_rows = true;
    }

    // ---- flush: per column the guide's nucleotide / codon bins (meta), then the bit-sliced counters transposed
    // and reduced over the waves through LDS, stored as this workgroup's partial
    __syncthreads();
    const int cols = G * 32;
    uint32_t* const part = a.partials + (size_t)blockIdx.x * (NC + 1) * cols;
    auto gbit = [&](int plane, int c) -> uint32_t { return (gl[plane * RD + (c >> 5)] >> (c & 31)) & 1u; };
    for (int c = tid; c < cols; c += T) {
        const uint32_t v0 = gbit(0, c), v1 = gbit(0, c + 1), v2 = gbit(0, c + 2);
        const uint32_t k0 = gbit(1, c) | (gbit(2, c) << 1), k1 = gbit(1, c + 1) | (gbit(2, c + 1) << 1),
                       k2 = gbit(1, c + 2) | (gbit(2, c + 2) << 1);
        part[(size_t)NC * cols + c] = k0 | (v0 << 2) | (((k0 << 4) | (k1 << 2) | k2) << 8) | ((v0 & v1 & v2) << 14);
    }
    uint32_t* const acc = lds;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        for (int i = tid; i < cols; i += T) acc[i] = 0;
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cg = p * 64 + lane;
            if (cg < G) {
                for (int j = 0; j < 32; ++j) {
                    uint32_t v = 0;
#pragma unroll
                    for (int l = 0; l < kLevels; ++l) v |= ((cnt[p][k][l] >> j) & 1u) << l;
                    if (v) atomicAdd(&acc[cg * 32 + j], v);
                }
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

typedef void (*pileup_kernel_fn)(StreamArgs);

pileup_kernel_fn pick_kernel(int mode, int passes)
{
    if (passes == 1) {
        return mode == QV_NONE ? pileup_kernel<QV_NONE, 1> : mode == QV_QUAL ? pileup_kernel<QV_QUAL, 1> : pileup_kernel<QV_ANY, 1>;
    }
    return mode == QV_NONE ? pileup_kernel<QV_NONE, 2> : mode == QV_QUAL ? pileup_kernel<QV_QUAL, 2> : pileup_kernel<QV_ANY, 2>;
}

}  // namespace

int msq_pileup_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                             int32_t W, uint32_t* d_nuc, uint32_t* d_codon, bool* handled)
{
    *handled = false;
    if (R.flags & MSQ_READS_HAS_EXOTIC) return MSQ_OK;
    const int G = (W + 31) / 32;
    if (G > 128) return MSQ_OK;  // wider windows: simple kernel (column super-tiles are future work)
    const int passes = G <= 64 ? 1 : 2;
    const int RD = ((G + 1 + 3) / 4) * 4;
    const size_t wave_lds = (size_t)4 * kRows * RD * 4;  // rows [4][8][RD] per wave
    const size_t shared_lds = (size_t)3 * RD * 4 + 64;   // guide [3][RD]
    // waves per workgroup: as many as fit one CU (LDS, 3 waves per SIMD), one workgroup per CU
    int mode = QV_NONE;
    const bool q_on = thr.qual >= 0 && R.qual;
    const bool rich_on = (thr.del >= 0 && R.delqv) || (thr.sub >= 0 && R.subqv) || (thr.ins >= 0 && R.insqv);
    if (rich_on) mode = QV_ANY;
    else if (q_on) mode = QV_QUAL;
    const int NC = mode == QV_NONE ? 3 : 4;
    const int max_waves = max_waves_of(mode, passes);
    int nwv = (int)std::min<size_t>(max_waves, (160 * 1024 - shared_lds - 1024) / wave_lds);
    if (nwv < 1) return MSQ_OK;
    const uint64_t n_batches = (R.n_reads + kRows - 1) / kRows;
    int n_wg = ctx->n_cu;
    if ((uint64_t)n_wg * nwv > n_batches) {  // small inputs: fewer, thinner workgroups
        nwv = (int)std::max<uint64_t>(1, std::min<uint64_t>(nwv, (n_batches + ctx->n_cu - 1) / ctx->n_cu));
        n_wg = (int)((n_batches + nwv - 1) / nwv);
    }
    while ((size_t)nwv * wave_lds < (size_t)G * 32 * 4) ++nwv;  // the flush reduces through the rows' LDS
    if (nwv > max_waves) return MSQ_OK;
    if ((n_batches + (uint64_t)n_wg * nwv - 1) / ((uint64_t)n_wg * nwv) > (uint64_t)kMaxBatchesPerWave) return MSQ_OK;
    const int T = nwv * 64;
    const size_t lds = (size_t)nwv * wave_lds + shared_lds;

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
    a.n_batches = n_batches;
    a.win_begin0 = win_begin0;
    a.W = W;
    a.G = G; a.RD = RD;
    a.nuc = d_nuc;
    a.codon = d_codon;
    a.partials = d_part;
    a.err = ctx->d_err;

    pileup_kernel_fn kern = pick_kernel(mode, passes);
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
