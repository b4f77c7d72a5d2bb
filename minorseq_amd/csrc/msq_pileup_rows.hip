// msq_pileup_rows.hip -- the pileup over COLUMN-ALIGNED sets (MSQ_READS_ALIGNED, msq_reads_align): MSAByColumn's counting
// and CodonsAt (src/MSA.cpp:55-92,126-136,197-219) with no row building at all.
//
// In an aligned set a group is a 32-column word of its read's row (AddRead's placement, src/MSA.cpp:148,158-171, was done
// when the set was made), so the word a lane counts is the word it loads:
//
//  * lane = one 32-column word of the window, pass p covers words 63 p .. 63 p + 62 in lanes 0 .. 62; lane 63 loads the
//    next word and only lends it to lane 62 (codons reach two columns into the next word): the neighbour's planes come
//    over a DPP wave shift, no LDS, no second load.
//  * a wave takes eight reads at a time (a batch, drawn from a per-workgroup LDS counter): one buffer_load_dwordx4 per
//    read and pass, the read's own buffer resource (base = its first group, records = its groups; a lane left or right
//    of the read is out of range and reads zeros), re-issued for the next batch as soon as its registers are free:
//    8 x PASSES KB in flight per wave at all times.
//  * cells that match the set's GUIDE (per column the majority base, learnt when the set was made) are counted in
//    bit-sliced carry-save counters in registers; a word with anything else goes -- planes and all, 20 bytes -- on a
//    per-wave LDS queue that is drained one lane per entry when 192 entries are waiting: what matches the set's alt
//    haplotype is counted in LDS, so are isolated substitutions (by column and rank; the fold expands them), the rest
//    pays global atomics.  Any guide gives the same counts.
//  * sets whose reads msq_reads_align grouped by haplotype: a workgroup counts ONE group, through the set's
//    group-ordered read list, against that group's guide.
//  * flush: the waves' bit-sliced counters are summed while still bit-sliced, transposed, and leave as ONE partial per
//    workgroup -- u16s, two columns to a u32; fold_rows_kernel adds the partials into the count matrices.
//
// Algorithmic bytes per unit: 0.5 B per column-consuming base + 20 B per read of tables; HBM-bound.
#include "msq_internal.h"
#include "msq_stream_common.h"

#include <algorithm>

namespace {

// MSQ_ROWS_EXP: compile-time timing experiments only (make EXTRA=-DMSQ_ROWS_EXP=n); any non-zero value gives WRONG counts.
// bit 0: the drain issues no global atomics; bit 1: nothing is pushed on the deviation queue; bit 2 (right counts):
// the kernel counts its queue entries, drains and global atomics into rows_dbg (read it with a debugger build's
// hipMemcpyFromSymbol; tools only).
#ifndef MSQ_ROWS_EXP
#define MSQ_ROWS_EXP 0
#endif
// MSQ_ROWS_PAIR: 0 = batches drawn from the role's LDS counter; 1 = a fixed sequence per wave, the same for the two roles'
// waves 2k, 2k + 1; 2 = the same, and neither wave of a pair runs more than one batch ahead of the other
#ifndef MSQ_ROWS_PAIR
#define MSQ_ROWS_PAIR 0
#endif
#if MSQ_ROWS_EXP & 4
__device__ unsigned long long rows_dbg[8];  // entries, drains, nucleotide atomics, codon atomics, LDS alt adds, LDS substitution adds
#define ROWS_DBG(i, n) dbg[i] += (n)
#else
#define ROWS_DBG(i, n) ((void)0)
#endif
constexpr int kRowsWaves = 12;     // waves per workgroup (one workgroup per CU, 3 waves per SIMD: 168 VGPRs, nothing spills)
constexpr int kRowsWavesN = 8;     // ... of the variant with an N counter (a fourth counter set: 2 waves per SIMD rather than spills)
constexpr int kQueueCap = 384;     // entries of a wave's deviation queue: 16 B of planes + a 4-byte tag each
constexpr int kQueueRoom = 128;    // two reads add at most this many entries: drained when fewer slots are left
constexpr int kQueueDrainAt = 192; // at the end of a batch: drain when at least this many entries wait (three full waves of lanes: a drain ends with vmcnt(0), a bubble in the load pipeline)
constexpr int kLanesPerPass = 63;  // productive lanes of a wave; lane 63 lends the next word

// Haplotype groups of a launch: group h is counted by the workgroups wg_first[h] .. wg_first[h + 1] - 1, which share the
// places lo[h] .. hi[h] - 1 of the set's group-ordered read list.
struct HapPlan {
    int n;
    int wg_first[MSQ_MAX_HAPS + 1];
    uint32_t lo[MSQ_MAX_HAPS], hi[MSQ_MAX_HAPS];
};

struct RowsArgs {
    const uint32_t* perm;        // NULL: no groups, the reads in their own order
    const uint32_t* hap_guides;  // [groups][guide_words][8]
    HapPlan hp;
    const int32_t* ref_start;
    const int32_t* ref_end;
    const uint64_t* group_off;
    const uint4* bop;
    const uint32_t* guide;  // [guide_words][8]
    uint32_t guide_words;
    uint64_t read_lo, read_hi, n_batches;
    int32_t win_begin0, W;
    int32_t origin;   // the set's align origin
    int32_t word0;    // grid word of window word 0: (win_begin0 - origin) / 32
    int32_t G;        // words of this launch's column slab (<= 126)
    int32_t slab_w0;  // window word of the slab's word 0
    int32_t Gtot;     // words of the window
    int32_t n_roles;  // 1: every wave counts words 0 .. 62; 2: the waves of role 1 count words 63 .. 125
    uint32_t flush_wide;
    uint32_t* nuc;    // where deviations go (the caller's matrices, or the context's deviation matrix in a fused step)
    uint32_t* codon;
    uint32_t* partials;  // [workgroups][NC + 5][G * 16]: u16 pairs, columns r and r + 16 of a word share a u32
    uint32_t* err;
    uint32_t* bounds_slots;  // msq_pileup_step: [workgroups][2], every workgroup's share of the shard's bounds (keyed maxima); the fold finishes them (NULL otherwise)
};

__device__ __forceinline__ uint32_t dpp_next_lane(uint32_t v)  // lane i <- lane i + 1, lane 63 <- 0 (wave_shl:1)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t dpp_prev_lane(uint32_t v)  // lane i <- lane i - 1, lane 0 <- 0 (wave_shr:1)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// The eight reads of a batch as the loads need them, lane i = read i: the first three words of the read's buffer
// resource (base address, records = its groups in bytes) and the byte offset of window word 0 inside the read.
struct RowTable {
    uint32_t w0, w1, nb;
    int32_t rel16;
};

template <bool MANY_N>
__global__ __launch_bounds__((MANY_N ? kRowsWavesN : kRowsWaves) * 64) void pileup_rows_kernel(RowsArgs a)
{
    constexpr int NC = MANY_N ? 4 : 3;
    constexpr bool COUNT_N = MANY_N;
    extern __shared__ __align__(16) uint32_t lds[];
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, NWV = T >> 6;
    const int wave = (int)uni((uint32_t)tid >> 6);
    const int G = a.G, W = a.W;
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    // a wave counts ONE 63-word range of the window: role 0 words 0 .. 62, role 1 words 63 .. 125 (windows of more than 63
    // words: the waves alternate).  Every role sees every batch of the workgroup.
    const int n_roles = a.n_roles;
    const int role = n_roles == 2 ? (wave & 1) : 0, wave_in_role = n_roles == 2 ? (wave >> 1) : wave, waves_per_role = NWV / n_roles;
    // LDS: per wave a deviation queue (planes, then tags); the batch counters; guide and alt haplotype words of the window
    // [G + 2][4] each; one alt counter per window column
    uint4* const q4 = reinterpret_cast<uint4*>(lds + (size_t)wave * (kQueueCap * 5));
    uint32_t* const qt = reinterpret_cast<uint32_t*>(q4 + kQueueCap);
    uint32_t* const draw_ctr = lds + (size_t)NWV * (kQueueCap * 5);  // [0], [1]: the roles' batch counters; [2], [3]: the bounds
    uint32_t* const prog = draw_ctr + 4;  // [16]: the batch each wave is at (pairs of waves keep in step)
    uint32_t* const gl = prog + 16;
    uint32_t* const gl2 = gl + 4 * (G + 2);
    uint32_t* const altc = gl2 + 4 * (G + 2);  // [G * 32]: alt nucleotide count | alt codon count << 16
    // isolated substitutions per column, by the rank r of the base read ((guide's code + 1 + r) & 3): [G * 32] r0 | r1 << 16,
    // then [G * 32] r2 (a workgroup sees < 65536 reads)
    uint32_t* const subA = altc + G * 32;
    uint32_t* const subB = subA + G * 32;
    uint32_t my_err = 0;
#if MSQ_ROWS_EXP & 4
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // ---- haplotype groups: this workgroup counts the reads of ONE group against that group's guide.  Places are read
    // indices (no groups) or indices into the group-ordered list
    uint32_t wg_eff = wg, nwg_eff = nwg;
    uint64_t pos_lo = a.read_lo, pos_hi = a.read_hi;
    uint32_t n_batches = (uint32_t)a.n_batches;
    const uint32_t* gsrc = a.guide;
    if (a.perm) {
        int hap = 0;
#pragma unroll
        for (int h = 1; h < MSQ_MAX_HAPS; ++h)
            if (h < a.hp.n && wg >= (uint32_t)a.hp.wg_first[h]) hap = h;
#pragma unroll
        for (int h = 0; h < MSQ_MAX_HAPS; ++h)
            if (hap == h) {
                wg_eff = wg - (uint32_t)a.hp.wg_first[h];
                nwg_eff = (uint32_t)(a.hp.wg_first[h + 1] - a.hp.wg_first[h]);
                pos_lo = a.hp.lo[h];
                pos_hi = a.hp.hi[h];
            }
        n_batches = (uint32_t)((pos_hi - pos_lo + kRows - 1) / kRows);
        gsrc = a.hap_guides + (size_t)hap * a.guide_words * 8;
    }

    // ---- set-up: the guide words of the window from the set's guide (grid word = window word + word0)
    if (tid < 20) draw_ctr[tid] = 0;
    for (int i = tid; i < (G + 2) * 4; i += T) {
        const int w = i >> 2, k = i & 3;
        const long long gw = (long long)a.word0 + a.slab_w0 + w;  // words G, G + 1: the next slab's first words, or nothing
        const bool in = a.slab_w0 + w < a.Gtot && gw >= 0 && gw < (long long)a.guide_words;
        gl[i] = in ? gsrc[(size_t)gw * 8 + k] : 0u;
        gl2[i] = in ? gsrc[(size_t)gw * 8 + 4 + k] : 0u;
    }
    for (int i = tid; i < 3 * G * 32; i += T) altc[i] = 0;  // alt counters and both substitution tables
    // (msq_pileup_step) this workgroup's share of the shard's reads for the window bounds
    int32_t b_lo = INT32_MAX, b_hi = INT32_MIN;
    const bool with_bounds = a.bounds_slots != nullptr;
    if (with_bounds) {
#pragma unroll 4
        for (uint64_t i = a.read_lo + (uint64_t)blockIdx.x * T + tid; i < a.read_hi; i += (uint64_t)gridDim.x * T) {
            b_lo = min(b_lo, a.ref_start[i]);
            b_hi = max(b_hi, a.ref_end[i]);
        }
    }
    __syncthreads();

    // ---- this lane's word: lanes 0 .. 62 count, lane 63 lends its word to lane 62
    const int lw = kLanesPerPass * role + lane;  // word of the slab
    const int cg = a.slab_w0 + lw;               // word of the window
    const bool act = lane < kLanesPerPass && lw < G;
    const uint32_t ccols = act ? codon_cols_of(cg, W) : 0u;
    const uint64_t actb = uni64(__ballot(act));
    // columns of this word at or beyond W: a base there is outside the window (the reference writes past the row)
    const int over = W - cg * 32;
    const uint32_t tail = act ? (over >= 32 ? 0u : over <= 0 ? 0xffffffffu : (0xffffffffu << over)) : 0u;
    const int gw_ = min(lw, G + 1);
    const uint32_t gx = gl[gw_ * 4], gy = gl[gw_ * 4 + 1], gz = gl[gw_ * 4 + 2];
    const int32_t lw16 = cg << 4;
    // the next word's columns 2, 3 go into a queue entry (isolated substitutions in the next word's first columns change
    // codons of this word) -- not at a seam: the word behind lane 62 and behind a slab's last word is counted by another
    // wave or launch, whose view of this word's last columns is empty (its lane 0 has no previous lane)
    const uint32_t mh_keep = (lane == kLanesPerPass - 1 || lw == G - 1) ? 0u : 3u;
    uint32_t seen = 0;
    uint32_t cnt[NC][kLevels];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int l = 0; l < kLevels; ++l) cnt[k][l] = 0;

    // ---- batches are drawn from the role's counter of the workgroup: the t-th draw is batch t * workgroups + this workgroup
    // (the workgroups of this workgroup's group)
    uint32_t n_taken = 0;
    auto draw = [&]() -> uint32_t {
        uint32_t t = 0xffffffffu;
        if (n_taken < (uint32_t)kMaxBatchesPerWave) {
#if MSQ_ROWS_PAIR
            t = n_taken * (uint32_t)waves_per_role + (uint32_t)wave_in_role;
#else
            if (lane == 0) t = __hip_atomic_fetch_add(&draw_ctr[role], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = uni(t);
#endif
            n_taken = uni(n_taken + 1u);
        }
        const uint32_t b = t == 0xffffffffu ? 0xffffffffu : t * nwg_eff + wg_eff;
        return b < n_batches ? b : 0xffffffffu;
    };
    // the table entries of a batch's reads, lane i = read i: which read sits at the lane's place (the place itself, or the
    // group-ordered list's entry -- fetched a batch before the tables), then its group offsets and start column; clamped
    // places -- a place past the end, and every place of "no batch", counts as a read without groups
    const uint32_t* const go32 = reinterpret_cast<const uint32_t*>(a.group_off);
    const uint64_t bop_base = (uint64_t)(uintptr_t)a.bop;
    auto fetch_read = [&](uint32_t b) -> uint64_t {
        const uint64_t p = min(pos_lo + (uint64_t)(b == 0xffffffffu ? 0u : b) * kRows + (uint64_t)min(lane, kRows - 1), pos_hi - 1);
        return a.perm ? (uint64_t)a.perm[p] : p;
    };
    auto fetch_table = [&](uint64_t r, uint32_t& go_lo, uint32_t& go_hi, uint32_t& go_nx, int32_t& rs) {
        go_lo = go32[2 * r];
        go_hi = go32[2 * r + 1];
        go_nx = go32[2 * r + 2];
        rs = a.ref_start[r];
    };
    auto make_table = [&](uint32_t b, uint32_t go_lo, uint32_t go_hi, uint32_t go_nx, int32_t rs) -> RowTable {
        const uint64_t o0 = ((uint64_t)go_hi << 32) | go_lo;
        const uint64_t p0 = pos_lo + (uint64_t)(b == 0xffffffffu ? 0u : b) * kRows;
        const bool there = b != 0xffffffffu && p0 + (uint64_t)lane < pos_hi && lane < kRows;
        const uint32_t ng = there ? go_nx - go_lo : 0u;  // a read spans < 2^32 groups
        const int32_t rel = ((rs - a.origin) >> 5) - a.word0;  // window word of the read's first group
        const uint64_t base = bop_base + (o0 << 4);
        // a read that reaches outside the window: assert(pos >= 0) / a write past the row (src/MSA.cpp:149)
        if (ng != 0u && (rel < 0 || rel + (int32_t)ng > a.Gtot)) my_err |= DEVERR_WINDOW;
        RowTable t;
        t.w0 = (uint32_t)base;
        t.w1 = (uint32_t)(base >> 32) & 0xffffu;
        t.nb = ng << 4;
        t.rel16 = rel << 4;
        return t;
    };
    // one word of one read: group (window word - rel) of the read, zeros outside it (the buffer's range check); `ins` says
    // whether this lane's word lies inside the read (a lane outside loaded zeros, which would read as 32 matching 'A's)
    auto load_word = [&](const RowTable& t, int i, uint32_t& ins) -> uint4 {
        const uint32_t nb = lane_value(t.nb, i);
        const int32_t koff = lw16 - (int32_t)lane_value((uint32_t)t.rel16, i);
        ins = (uint32_t)koff < nb ? 0xffffffffu : 0u;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(uintptr_t)(((uint64_t)lane_value(t.w1, i) << 32) | lane_value(t.w0, i)), 0, (int)nb, 0x00020000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, koff, 0, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };

    // ---- the deviation queue
    uint32_t qn = 0;  // entries waiting (wave-uniform)
    auto drain = [&]() {
        wave_sync();  // the entries other lanes wrote
        const uint32_t nq = uni(qn);
        if (lane == 0) ROWS_DBG(1, 1);
        for (uint32_t e = lane; e < nq; e += 64) {
            ROWS_DBG(0, 1);
            const uint4 lo = q4[e];  // V, X, B0, B1 of the word
            const uint32_t tag = qt[e];
            const int cgq = (int)(tag & 0xffu);       // word of the slab: guide words and LDS tables
            const int cgw = a.slab_w0 + cgq;          // word of the window: columns of the matrices
            const uint32_t hv = (tag >> 8) & 3u, hb0 = (tag >> 10) & 3u, hb1 = (tag >> 12) & 3u;  // the next word's first two columns
            const uint4 gl_lo = *reinterpret_cast<const uint4*>(gl + cgq * 4), gl_hi = *reinterpret_cast<const uint4*>(gl + cgq * 4 + 4);
            const uint32_t cc = codon_cols_of(cgw, W);
            const uint32_t m_lo = lo.x & gl_lo.x & ~((lo.z ^ gl_lo.y) | (lo.w ^ gl_lo.z));
            const uint32_t m_hi = hv & gl_hi.x & ~((hb0 ^ gl_hi.y) | (hb1 ^ gl_hi.z));
            const uint32_t cm = m_lo & funnel(m_lo, m_hi, 1) & funnel(m_lo, m_hi, 2) & cc;
            const uint32_t cv = lo.x & funnel(lo.x, hv, 1) & funnel(lo.x, hv, 2) & cc;
            const int cbase = cgw * 32, lbase = cgq * 32;
            // columns of this word inside the window: a read that reaches past column W is reported (tail, DEVERR_WINDOW)
            // and must not count beyond the caller's matrices
            const int ov = W - cbase;
            const uint32_t inw = ov >= 32 ? 0xffffffffu : ov <= 0 ? 0u : ((1u << ov) - 1u);
            const uint32_t nd = lo.x & ~m_lo & inw, cd = cv & ~cm, nn = COUNT_N ? 0u : (lo.y & lo.z & inw);
            // An ISOLATED substitution -- a base that differs from the guide's while the two columns on either side of
            // it match the guide (what a sequencing error looks like) -- changes exactly one
            // nucleotide bin and the three codons that hold it, each in one position: it is counted ONCE, in LDS, by its
            // column and the rank of the base read, and the fold turns the sums into the four bins.  Returnless global
            // atomics run at ~26e9 / s on the whole chip; at the reference's error rate (0.05 %) the four atomics per
            // substitution were a third of the kernel's time, and uniform noise of 2 % cost 20 x a clean sample.
            // Isolated substitutions come FIRST, before the alt haplotype gets its say: on noisy reads the alt that a
            // group learns carries noise of its own in many columns, a substitution that happened to agree with it
            // was booked as "alt" -- and its three codons, which agree with the alt only if the alt is clean around
            // it, went to HBM one atomic each (1 % substitutions: 22 of them per read, two thirds of the kernel's time).
            uint32_t nd_w = nd, cd_w = cd;
            // match bits of columns -2 .. 35 of the word (bit = column + 2): the neighbours' as the tag brought them
            const uint32_t mh4 = m_hi | (((tag >> 16) & 3u) << 2);
            const uint64_t M = (uint64_t)((tag >> 14) & 3u) | ((uint64_t)m_lo << 2) | ((uint64_t)mh4 << 34);
            // the next word's isolated substitutions in its columns 0 and 1 (its own entry counts them: the same bits, the
            // same test) change this word's codons at columns 30, 31 / 31: the fold adds those as well
            const uint32_t fdev = hv & ~m_hi;
            if ((fdev & 1u) && (m_lo >> 30) == 3u && (mh4 & 6u) == 6u) cd_w &= ~(3u << 30);
            if ((fdev & 2u) && (m_lo >> 31) && (mh4 & 0xdu) == 0xdu) cd_w &= ~(1u << 31);
            uint32_t bits = nd_w;
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                if (((uint32_t)(M >> j) & 0x1bu) != 0x1bu) continue;  // a neighbour deviates, is a gap or is not there
                const uint32_t code = ((lo.z >> j) & 1u) | (((lo.w >> j) & 1u) << 1);
                const uint32_t g = ((gl_lo.y >> j) & 1u) | (((gl_lo.z >> j) & 1u) << 1);
                const uint32_t r = (code - g - 1u) & 3u;  // 0 .. 2: code != g
                if (r < 2u)
                    __hip_atomic_fetch_add(&subA[lbase + j], 1u << (16u * r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_fetch_add(&subB[lbase + j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                nd_w &= ~(1u << j);
                ROWS_DBG(5, 1);
                cd_w &= ~(uint32_t)((7ull << j) >> 2);  // the codons that start at j - 2, j - 1, j (those of this word): the fold adds them where they count
            }
            // what is left and matches the set's alt haplotype -- a minor variant carried by a share of the reads -- is
            // counted in LDS too (the sum leaves with the workgroup's partial); only what matches neither pays a global atomic
            const uint4 al = *reinterpret_cast<const uint4*>(gl2 + cgq * 4), ah = *reinterpret_cast<const uint4*>(gl2 + cgq * 4 + 4);
            const uint32_t am_lo = lo.x & al.x & ~((lo.z ^ al.y) | (lo.w ^ al.z));
            const uint32_t am_hi = hv & ah.x & ~((hb0 ^ ah.y) | (hb1 ^ ah.z));
            const uint32_t nd_alt = nd_w & am_lo;
            const uint32_t cd_alt = cd_w & am_lo & funnel(am_lo, am_hi, 1) & funnel(am_lo, am_hi, 2);
            bits = nd_alt | cd_alt;
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                __hip_atomic_fetch_add(&altc[lbase + j], ((nd_alt >> j) & 1u) | (((cd_alt >> j) & 1u) << 16), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                ROWS_DBG(4, 1);
            }
            const uint32_t nd_rest = nd_w & ~nd_alt, cd_rest = cd_w & ~cd_alt;
            bits = nd_rest | nn;  // disjoint: nd lies in V, nn in X
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t code = ((nn >> j) & 1u) ? 5u : (((lo.z >> j) & 1u) | (((lo.w >> j) & 1u) << 1));
                ROWS_DBG(2, 1);
#if !(MSQ_ROWS_EXP & 1)
                atomicAdd(&a.nuc[(uint32_t)(cbase + j) * 6u + code], 1u);  // MSAColumn::IncCounts, MSA.inl:66
#endif
            }
            bits = cd_rest;
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t t0 = funnel(lo.z, hb0, j) & 7u, t1 = funnel(lo.w, hb1, j) & 7u;
                ROWS_DBG(3, 1);
                const uint32_t idx = ((t1 & 1u) << 5) | ((t0 & 1u) << 4) | ((t1 & 2u) << 2) | ((t0 & 2u) << 1) | ((t1 & 4u) >> 1) | ((t0 & 4u) >> 2);
#if !(MSQ_ROWS_EXP & 1)
                atomicAdd(&a.codon[(uint32_t)(cbase + j) * 64u + idx], 1u);  // ++codons[codon], MSA.cpp:132
#else
                if (idx == 77u) my_err |= DEVERR_RANGE;
#endif
            }
        }
        qn = 0;
        // loads and returnless atomics share vmcnt and may complete out of order: retire the atomics here, so that the
        // word loads behind are waited for by count again
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        wave_sync();
    };

    // ---- one read: the loaded word against the guide (MSAByColumn's loop, src/MSA.cpp:68-87; CodonsAt, :126-136,197-219)
    // -> its match / gap / codon / N masks; a word with anything else goes on the queue
    auto one_row = [&](const uint4 w, const uint32_t ins, uint32_t (&x)[NC]) {
        const uint32_t V = ~(w.z | w.w) & ins;   // op 0: '=' / 'X'   (w = nucleotide bit 0, bit 1, op bit 0, bit 1)
        const uint32_t X = w.z & ~w.w;           // op 1: '-' or literal N
        const uint32_t f = (w.x ^ gy) | (w.y ^ gz);
        const uint32_t m = V & gx & ~f;          // the guide's base
        const uint32_t m_hi = dpp_next_lane(m), v_hi = dpp_next_lane(V);
        // codon at column c = columns c, c + 1, c + 2.  Columns that cannot start a codon (the window's first and its
        // last two, codon_cols_of) are NOT masked here: their counters are cleared at the flush, and the drain masks
        // them where it counts a deviation -- two operations less per word for the two words they can be in
        const uint32_t cm = m & funnel(m, m_hi, 1) & funnel(m, m_hi, 2);
        const uint32_t cv = V & funnel(V, v_hi, 1) & funnel(V, v_hi, 2);
        const uint32_t nn = X & w.x;
        seen |= V | X;
        x[0] = m;
        x[1] = X & ~w.x;
        x[2] = cm;
        if (COUNT_N) x[NC - 1] = nn;
        const uint32_t dev = (V & ~m) | (cv & ~cm) | (COUNT_N ? 0u : nn);
        const uint64_t bal = __ballot(dev != 0) & actb;  // lanes that count (not the lending lane, not lanes past the window)
        if (bal && !(MSQ_ROWS_EXP & 2)) {  // wave-uniform
            const bool has = (bal >> lane) & 1u;
            // the next word's first two columns (codons reach into them), for every lane: all lanes are active here
            const uint32_t b0 = w.x & (V | X), b1 = w.y & V;
            // and the match bits around the word: the previous word's columns 30, 31, the next word's columns 2, 3
            const uint32_t tag = (uint32_t)lw | ((v_hi & 3u) << 8) | ((dpp_next_lane(b0) & 3u) << 10) | ((dpp_next_lane(b1) & 3u) << 12) |
                                 ((dpp_prev_lane(m) >> 30) << 14) | (((m_hi >> 2) & mh_keep) << 16);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, qn));
            if (has) {
                q4[pos] = make_uint4(V, X, b0, b1);
                qt[pos] = tag;
            }
            qn = uni(qn + (uint32_t)__popcll(bal));
        }
    };

    // ---- the stream: the words of the current batch sit in registers; the registers of four reads are re-loaded
    // with the next batch's words as soon as those four reads are counted; the table of the batch after next is in flight
    uint32_t b_cur = draw(), b_nxt = draw(), b_nn = draw(), b_n3 = draw();
    uint32_t f_lo, f_hi, f_nx;
    int32_t f_rs;
    fetch_table(fetch_read(b_cur), f_lo, f_hi, f_nx, f_rs);
    RowTable tab = make_table(b_cur, f_lo, f_hi, f_nx, f_rs);
    uint4 wd[kRows];
    uint32_t ins[kRows];
#pragma unroll
    for (int i = 0; i < kRows; ++i) wd[i] = load_word(tab, i, ins[i]);
    fetch_table(fetch_read(b_nxt), f_lo, f_hi, f_nx, f_rs);
    tab = make_table(b_nxt, f_lo, f_hi, f_nx, f_rs);
    fetch_table(fetch_read(b_nn), f_lo, f_hi, f_nx, f_rs);
    uint64_t r_n3 = fetch_read(b_n3);
    if (with_bounds) {  // wave maxima (order-preserving unsigned codes) meet in LDS, one atomic pair per workgroup goes out
        uint32_t k_lo = ~((uint32_t)b_lo ^ 0x80000000u), k_hi = (uint32_t)b_hi ^ 0x80000000u;
        for (int d = 32; d >= 1; d >>= 1) {
            k_lo = max(k_lo, (uint32_t)__shfl_xor((int)k_lo, d));
            k_hi = max(k_hi, (uint32_t)__shfl_xor((int)k_hi, d));
        }
        if (lane == 0) {
            atomicMax(&draw_ctr[2], k_lo);
            atomicMax(&draw_ctr[3], k_hi);
        }
    }
#if MSQ_ROWS_PAIR == 2
    uint32_t it = 0;
#endif
    while (b_cur != 0xffffffffu) {
#if MSQ_ROWS_PAIR == 2
        // The two waves of a pair load neighbouring bytes of the same rows -- the 128-byte lines at the seam of the roles'
        // ranges are fetched by both.  Kept within one batch of each other, the second finds them in the L2.
        if (n_roles == 2) {
            it = uni(it + 1u);
            if (lane == 0) __hip_atomic_store(&prog[wave], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (uni(__hip_atomic_load(&prog[wave ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + 1u < it) __builtin_amdgcn_s_sleep(1);
        }
#endif
        uint32_t c4[2][NC], c8[NC];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t even[NC], twos[NC];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t x[NC];
                one_row(wd[4 * h + i], ins[4 * h + i], x);
#pragma unroll
                for (int k = 0; k < NC; ++k) {  // streaming Harley-Seal: rows fold in pairs as they come
                    if (i == 0 || i == 2) {
                        even[k] = x[k];
                    } else if (i == 1) {
                        csa(cnt[k][0], even[k], x[k], cnt[k][0], twos[k]);
                    } else {
                        uint32_t t2;
                        csa(cnt[k][0], even[k], x[k], cnt[k][0], t2);
                        csa(cnt[k][1], twos[k], t2, cnt[k][1], c4[h][k]);
                    }
                }
                if ((i & 1) && qn > (uint32_t)(kQueueCap - kQueueRoom)) drain();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) wd[4 * h + i] = load_word(tab, 4 * h + i, ins[4 * h + i]);
        }
        {
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                csa(cnt[k][2], c4[0][k], c4[1][k], cnt[k][2], c8[k]);
                c8[k] = ripple<3, kLowLevels>(cnt[k], c8[k]);
                any |= c8[k];
            }
            if (__ballot(any != 0)) {
#pragma unroll
                for (int k = 0; k < NC; ++k) (void)ripple<kLowLevels, kLevels>(cnt[k], c8[k]);
            }
        }
        if (qn >= (uint32_t)kQueueDrainAt) drain();
        tab = make_table(b_nn, f_lo, f_hi, f_nx, f_rs);  // the entries fetched a batch ago
        b_cur = b_nxt;
        b_nxt = b_nn;
        b_nn = b_n3;
        fetch_table(r_n3, f_lo, f_hi, f_nx, f_rs);  // of the read looked up a batch ago
        b_n3 = draw();
        r_n3 = fetch_read(b_n3);
    }
    if (qn) drain();
#if MSQ_ROWS_PAIR == 2
    if (lane == 0) __hip_atomic_store(&prog[wave], 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif

    // ---- flush: per column the guide's and the alt haplotype's bins (meta), the alt counts; then the waves' bit-sliced
    // counters added up while still bit-sliced, one wave per (counter, role), the sum transposed and stored
    if (seen & tail) my_err |= DEVERR_WINDOW;  // a base at or beyond column W
#pragma unroll
    for (int l = 0; l < kLevels; ++l) cnt[2][l] &= ccols;  // columns that cannot start a codon were counted along
    __syncthreads();
    if (with_bounds && tid < 2) a.bounds_slots[2 * wg + tid] = draw_ctr[2 + tid];  // no atomics: 256 workgroups end together, and the fold comes anyway
    // no workgroup counts 65 536 reads, so a partial holds u16s, two to a u32: columns r and r + 16 of a 32-column word --
    // the way the transposed counters come out.  Rows: the NC counters, the alt nucleotide and codon counts, the three
    // ranks of the isolated substitutions; the fold works the guides' bins out from the guides themselves
    const int words16 = G * 16;
    uint32_t* const part = a.partials + (size_t)wg * (NC + 5) * words16;
    for (int i = tid; i < words16; i += T) {
        const int c0 = (i >> 4) * 32 + (i & 15), c1 = c0 + 16;
        const uint32_t a0 = altc[c0], a1 = altc[c1], s0 = subA[c0], s1 = subA[c1];
        part[(size_t)(NC + 0) * words16 + i] = (a0 & 0xffffu) | (a1 << 16);
        part[(size_t)(NC + 1) * words16 + i] = (a0 >> 16) | (a1 & 0xffff0000u);
        part[(size_t)(NC + 2) * words16 + i] = (s0 & 0xffffu) | (s1 << 16);
        part[(size_t)(NC + 3) * words16 + i] = (s0 >> 16) | (s1 & 0xffff0000u);
        part[(size_t)(NC + 4) * words16 + i] = (subB[c0] & 0xffffu) | (subB[c1] << 16);
    }
    constexpr int SL = kLevels + 4;  // levels of a sum over <= 16 waves
    uint32_t* const slab = lds;      // [roles of the round][NC][waves of a role][kLevels][64]: queues and guide are dead
    auto round = [&](const int r_lo, const int r_hi) {
        __syncthreads();  // the guide words / the previous round's sums have been read
        if (role >= r_lo && role < r_hi) {
#pragma unroll
            for (int k = 0; k < NC; ++k)
#pragma unroll
                for (int l = 0; l < kLevels; ++l) slab[((((role - r_lo) * NC + k) * waves_per_role + wave_in_role) * kLevels + l) * 64 + lane] = cnt[k][l];
        }
        __syncthreads();
        for (int job = wave; job < NC * (r_hi - r_lo); job += NWV) {  // wave-uniform
            const int k = job % NC, rj = r_lo + job / NC;
            const uint32_t* const base = slab + (size_t)((rj - r_lo) * NC + k) * waves_per_role * kLevels * 64;
            uint32_t sum[SL];
#pragma unroll
            for (int l = 0; l < SL; ++l) sum[l] = 0;
            for (int e = 0; e < waves_per_role; ++e) {
                const uint32_t* src = base + e * kLevels * 64 + lane;
                uint32_t carry = 0;
#pragma unroll
                for (int l = 0; l < SL; ++l) csa(sum[l], l < kLevels ? src[l * 64] : 0u, carry, sum[l], carry);
            }
            const int cgj = kLanesPerPass * rj + lane;  // this lane's word in role rj
            if (lane < kLanesPerPass && cgj < G) {
                static_assert(SL <= 16, "the transposition holds 16 levels");
                uint32_t t16[16];
#pragma unroll
                for (int l = 0; l < 16; ++l) t16[l] = l < SL ? sum[l] : 0u;
                transpose16x2(t16);
                uint4* dst = reinterpret_cast<uint4*>(part + (size_t)k * words16 + cgj * 16);  // t16[r] = column r | column r + 16 << 16
#pragma unroll
                for (int r = 0; r < 16; r += 4) dst[r >> 2] = make_uint4(t16[r], t16[r + 1], t16[r + 2], t16[r + 3]);
            }
        }
    };
    if (a.flush_wide) {
        round(0, n_roles);
    } else {
        for (int r = 0; r < n_roles; ++r) round(r, r + 1);
    }
    if (my_err) atomicOr(a.err, my_err);
#if MSQ_ROWS_EXP & 4
    for (int i = 0; i < 8; ++i)
        if (dbg[i]) atomicAdd(&rows_dbg[i], dbg[i]);
#endif
}

// ---- the fold of the aligned-set kernel's partials into the count matrices (MSAColumn's counts, src/MSA.cpp:68-87; the
// codon map of CodonsAt, :126-136).  Block = one 32-column word x 32 "subs": thread (column, sub) adds up the u16s of
// the workgroups sub, sub + 32, ... -- runs of workgroups of one haplotype group in registers, then per (group, column)
// in LDS -- and thread (column, sub = group) turns the group's sums into bins with the group's guide: matches to the
// guide's nucleotide / codon bin, gaps to '-', Ns to 'N' (bins 4 and 5, NucleotideConversion.h:69-88), alt counts to the
// alt's bins, an isolated substitution of rank r at column c to nucleotide (guide + 1 + r) & 3 and to the three codons
// that hold it.  blockIdx.y slices the workgroups (more blocks; the bins then leave as global atomics).
struct FoldRowsArgs {
    const uint32_t* partials;
    const uint32_t* guides;  // group h's guide at guides + h * guide_words * 8 (no groups: the set's guide)
    uint32_t guide_words;
    int n_wg, NC, G, slab_w0, Gtot, word0, W;
    int n_haps;
    int wg_first[MSQ_MAX_HAPS + 1];
    uint32_t *nuc, *codon, *dev_nuc, *dev_codon;
    const uint32_t* bounds_slots;  // msq_pileup_step: the workgroups' shares of the shard's bounds; block 0 finishes them
    int32_t* bounds_out;
};
constexpr int kFoldRowsSubs = 64, kFoldRowsTrip = 4;  // 16 lanes (a u32 = columns r and r + 16) x 64 subs; 4 workgroups per thread and trip

__global__ __launch_bounds__(1024) void fold_rows_kernel(FoldRowsArgs f)
{
    constexpr int kCodStride = 65, kNucStride = 7, kQ = 9;  // per (group, column): m, gap, cm, n, alt nuc, alt codon, r0, r1, r2
    __shared__ uint32_t cod_acc[32 * kCodStride];
    __shared__ uint32_t nuc_acc[32 * kNucStride];
    __shared__ uint32_t gacc[MSQ_MAX_HAPS * 34 * kQ];  // columns 32, 33: the next word's first two (substitution ranks only)
    __shared__ uint32_t gmeta[MSQ_MAX_HAPS * 32], gmeta2[MSQ_MAX_HAPS * 32];
    __shared__ int wgf[MSQ_MAX_HAPS + 1];
    __shared__ uint32_t bmax[2];
    const int T = 1024, tid = threadIdx.x, r = tid & 15, sub = tid >> 4, NC = f.NC, G = f.G;
    const int w = blockIdx.x;              // word of the slab
    const int cb = (f.slab_w0 + w) * 32;   // window column of the block's column 0
    const int ncol = min(32, f.W - cb);    // columns of this word inside the window
    const int per = (f.n_wg + gridDim.y - 1) / gridDim.y;
    const int wg_lo = blockIdx.y * per, wg_hi = min(f.n_wg, wg_lo + per);
    const size_t row_words = (size_t)G * 16, wg_words = (size_t)(NC + 5) * row_words;
    // ---- every global load of the kernel is issued here, in one go: what the deviation path left in its matrix
    // (msq_pileup_step), the guides' words, the first trip over the partials and the next word's first columns
    uint32_t dev_c0 = 0, dev_c1 = 0, dev_n = 0;
    if (f.dev_nuc) {
        if (tid < ncol * 64) dev_c0 = f.dev_codon[(size_t)cb * 64 + tid];
        if (tid + T < ncol * 64) dev_c1 = f.dev_codon[(size_t)cb * 64 + tid + T];
        if (tid < ncol * 6) dev_n = f.dev_nuc[(size_t)cb * 6 + tid];
    }
    const bool does_bounds = f.bounds_slots && blockIdx.x == 0 && blockIdx.y == 0;
    uint32_t bk_lo = 0, bk_hi = 0;  // keyed maxima: 0 is below every key
    if (does_bounds)
        for (int i = tid; i < f.n_wg; i += T) {
            bk_lo = max(bk_lo, f.bounds_slots[2 * i]);
            bk_hi = max(bk_hi, f.bounds_slots[2 * i + 1]);
        }
    uint32_t g0[8], g1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g0[k] = g1[k] = 0;
    if (tid < MSQ_MAX_HAPS * 32 && (tid >> 5) < f.n_haps) {
        const uint32_t* g = f.guides + (size_t)(tid >> 5) * f.guide_words * 8;
        const long long gw = (long long)f.word0 + f.slab_w0 + w;
        if (f.slab_w0 + w < f.Gtot && gw >= 0 && gw < (long long)f.guide_words)
#pragma unroll
            for (int k = 0; k < 8; ++k) g0[k] = g[(size_t)gw * 8 + k];
        if (f.slab_w0 + w + 1 < f.Gtot && gw + 1 >= 0 && gw + 1 < (long long)f.guide_words)
#pragma unroll
            for (int k = 0; k < 8; ++k) g1[k] = g[(size_t)(gw + 1) * 8 + k];
    }
    const bool margin = r < 2 && w + 1 < G;
    const uint32_t* const pcol = f.partials + (size_t)w * 16 + r;
    uint32_t v[kFoldRowsTrip][kQ], vm[kFoldRowsTrip][3];
    auto load_trip = [&](int wg0) {
#pragma unroll
        for (int u = 0; u < kFoldRowsTrip; ++u) {
            const int wg = min(wg0 + kFoldRowsSubs * u, wg_hi - 1);  // clamped: a slot past the end is loaded and ignored
            const uint32_t* p = pcol + (size_t)wg * wg_words;
            v[u][0] = p[0];
            v[u][1] = p[row_words];
            v[u][2] = p[2 * row_words];
            v[u][3] = NC > 3 ? p[3 * row_words] : 0u;
#pragma unroll
            for (int q = 0; q < 5; ++q) v[u][4 + q] = p[(size_t)(NC + q) * row_words];
#pragma unroll
            for (int q = 0; q < 3; ++q) vm[u][q] = margin ? p[(size_t)(NC + 2 + q) * row_words + 16] : 0u;  // word w + 1, columns 0 / 1
        }
    };
    const bool any = wg_lo + sub < wg_hi;
    if (any) load_trip(wg_lo + sub);
    // ---- LDS: cleared; the groups' first workgroups; the bins of group h's guide (and alt) at every column of this word
    // (the kernel's meta_of over the same guide words)
    for (int i = tid; i < 32 * kCodStride; i += T) cod_acc[i] = 0;
    for (int i = tid; i < 32 * kNucStride; i += T) nuc_acc[i] = 0;
    for (int i = tid; i < MSQ_MAX_HAPS * 34 * kQ; i += T) gacc[i] = 0;
    if (tid < 2) bmax[tid] = 0;
    if (tid <= MSQ_MAX_HAPS) {
        int x = f.wg_first[MSQ_MAX_HAPS];
#pragma unroll
        for (int k = MSQ_MAX_HAPS - 1; k >= 0; --k)
            if (tid == k) x = f.wg_first[k];
        wgf[tid] = x;
    }
    if (tid < MSQ_MAX_HAPS * 32) {
        const int cl = tid & 31;
        auto meta_of = [&](int o) -> uint32_t {  // o = 0: the guide's planes, 4: the alt's
            auto bit = [&](int plane, int cc) -> uint32_t { return cc < 32 ? (g0[o + plane] >> cc) & 1u : (g1[o + plane] >> (cc - 32)) & 1u; };
            const uint32_t v0 = bit(0, cl), v1 = bit(0, cl + 1), v2 = bit(0, cl + 2);
            const uint32_t k0 = bit(1, cl) | (bit(2, cl) << 1), k1 = bit(1, cl + 1) | (bit(2, cl + 1) << 1), k2 = bit(1, cl + 2) | (bit(2, cl + 2) << 1);
            return k0 | (v0 << 2) | (((k0 << 4) | (k1 << 2) | k2) << 8) | ((v0 & v1 & v2) << 14);
        };
        gmeta[tid] = meta_of(0);
        gmeta2[tid] = meta_of(4);
    }
    __syncthreads();
    if (does_bounds) {  // UpdateBoundaries over the shard (src/MSA.cpp:96-97,138-142): the maxima of the workgroups' keyed shares
        if (bk_lo) atomicMax(&bmax[0], bk_lo);
        if (bk_hi) atomicMax(&bmax[1], bk_hi);
    }
    // ---- thread (r, sub): the u16s of columns r and r + 16 of the workgroups sub, sub + 64, ...; runs of workgroups of
    // one group in registers, then per (group, column) in LDS
    auto hap_of = [&](int wg) -> int {
        int h = 0;
#pragma unroll
        for (int k = 1; k < MSQ_MAX_HAPS; ++k)
            if (k < f.n_haps && wg >= f.wg_first[k]) h = k;
        return h;
    };
    if (any) {
        uint32_t lo[kQ], hi[kQ], mg[3];
        int cur = -1;
        auto flush = [&]() {
            if (cur >= 0) {
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    if (lo[q]) atomicAdd(&gacc[(cur * 34 + r) * kQ + q], lo[q]);
                    if (hi[q]) atomicAdd(&gacc[(cur * 34 + r + 16) * kQ + q], hi[q]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (mg[q]) atomicAdd(&gacc[(cur * 34 + 32 + r) * kQ + 6 + q], mg[q]);  // r < 2 only
            }
#pragma unroll
            for (int q = 0; q < kQ; ++q) lo[q] = hi[q] = 0;
            mg[0] = mg[1] = mg[2] = 0;
        };
        flush();
        for (int wg0 = wg_lo + sub; wg0 < wg_hi; wg0 += kFoldRowsSubs * kFoldRowsTrip) {
            if (wg0 != wg_lo + sub) load_trip(wg0);
#pragma unroll
            for (int u = 0; u < kFoldRowsTrip; ++u) {
                if (wg0 + kFoldRowsSubs * u >= wg_hi) break;
                const int h = f.n_haps > 1 ? hap_of(wg0 + kFoldRowsSubs * u) : 0;
                if (h != cur) {
                    flush();
                    cur = h;
                }
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    lo[q] += v[u][q] & 0xffffu;
                    hi[q] += v[u][q] >> 16;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) mg[q] += vm[u][q] & 0xffffu;
            }
        }
        flush();
    }
    __syncthreads();
    // ---- thread (column, group): the group's sums into bins, with the group's guide
    {
        const int cl = tid & 31, h = tid >> 5;
        if (h < f.n_haps && cl < ncol && wgf[min(h, MSQ_MAX_HAPS - 1) + 1] > wgf[min(h, MSQ_MAX_HAPS - 1)]) {
            const int wc = cb + cl;
            const uint32_t* const ga = gacc + (h * 34 + cl) * kQ;
            const uint32_t meta = gmeta[h * 32 + cl], meta2 = gmeta2[h * 32 + cl];
            const uint32_t k0 = meta & 3u, cidx = (meta >> 8) & 63u;
            uint32_t* const na = nuc_acc + cl * kNucStride;
            uint32_t* const ca = cod_acc + cl * kCodStride;
            if (ga[0]) atomicAdd(&na[k0], ga[0]);  // a count > 0 implies a valid base in that plane set
            if (ga[1]) atomicAdd(&na[4], ga[1]);
            if (ga[2]) atomicAdd(&ca[cidx], ga[2]);
            if (ga[3]) atomicAdd(&na[5], ga[3]);
            if (ga[4]) atomicAdd(&na[meta2 & 3u], ga[4]);
            if (ga[5]) atomicAdd(&ca[(meta2 >> 8) & 63u], ga[5]);
#pragma unroll
            for (int k = 0; k < 3; ++k)  // MSAColumn::IncCounts: the base read at this column
                if (const uint32_t n = ga[6 + k]) atomicAdd(&na[(k0 + 1u + (uint32_t)k) & 3u], n);
            if (wc >= 1 && wc <= f.W - 3) {  // a column that can start a codon (MSARow::CodonAt's i > 0, CodingCodonAt's bound)
#pragma unroll
                for (int q = 0; q < 3; ++q) {  // the substitution sits in position q of the codon that starts here
                    const int shift = 4 - 2 * q;
                    const uint32_t kq = (cidx >> shift) & 3u;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (const uint32_t n = ga[q * kQ + 6 + k]) atomicAdd(&ca[(cidx & ~(3u << shift)) | (((kq + 1u + (uint32_t)k) & 3u) << shift)], n);
                }
            }
        }
    }
    __syncthreads();
    if (does_bounds && tid == 0) {
        const int32_t begin0 = (int32_t)(~bmax[0] ^ 0x80000000u), end0 = (int32_t)(bmax[1] ^ 0x80000000u);
        f.bounds_out[0] = -begin0;
        f.bounds_out[1] = end0 > 0 ? end0 : 0;
    }
    uint32_t* const nuc = f.nuc + (size_t)cb * 6;
    uint32_t* const codon = f.codon + (size_t)cb * 64;
    if (f.dev_nuc) {
        // msq_pileup_step: this block is the only writer of its columns (gridDim.y == 1, one launch): the matrices are
        // STORED -- the fold's sums plus what the deviation path added to its own matrix, which is handed back cleared
        uint32_t* const dev_nuc = f.dev_nuc + (size_t)cb * 6;
        uint32_t* const dev_codon = f.dev_codon + (size_t)cb * 64;
        if (tid < ncol * 64) {
            codon[tid] = cod_acc[(tid >> 6) * kCodStride + (tid & 63)] + dev_c0;
            if (dev_c0) dev_codon[tid] = 0;
        }
        if (tid + T < ncol * 64) {
            const int i = tid + T;
            codon[i] = cod_acc[(i >> 6) * kCodStride + (i & 63)] + dev_c1;
            if (dev_c1) dev_codon[i] = 0;
        }
        if (tid < ncol * 6) {
            nuc[tid] = nuc_acc[(tid / 6) * kNucStride + tid % 6] + dev_n;
            if (dev_n) dev_nuc[tid] = 0;
        }
        return;
    }
    for (int i = tid; i < 32 * 64; i += T) {  // only columns inside the window hold anything
        const uint32_t x = cod_acc[(i >> 6) * kCodStride + (i & 63)];
        if (x) atomicAdd(&codon[i], x);
    }
    for (int i = tid; i < 32 * 6; i += T) {
        const uint32_t x = nuc_acc[(i / 6) * kNucStride + i % 6];
        if (x) atomicAdd(&nuc[i], x);
    }
}

typedef void (*rows_kernel_fn)(RowsArgs);
rows_kernel_fn pick_rows_kernel(bool many_n) { return many_n ? pileup_rows_kernel<true> : pileup_rows_kernel<false>; }

}  // namespace

// tools: the kernel's event counts since the last call (zeros unless built with MSQ_ROWS_EXP & 4)
extern "C" int msq_debug_rows_counters(unsigned long long* out8)
{
#if MSQ_ROWS_EXP & 4
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MSQ_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(rows_dbg), sizeof(zero)));
    MSQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rows_dbg), zero, sizeof(zero)));
#else
    for (int i = 0; i < 8; ++i) out8[i] = 0;
#endif
    return MSQ_OK;
}

// The plan of launch j of n_launch over a set with haplotype groups: group h's share of the launch is the j-th of n_launch
// equal ranges of its places; wr waves per role and workgroup (fewer for small inputs), p batches per workgroup -- the
// smallest p for which the groups need no more workgroups than the chip has CUs.  false: p exceeds what a workgroup's
// counters hold (more launches are needed).
static bool plan_haps(const msq_packed_reads& R, const uint64_t* hap_off, uint64_t j, uint64_t n_launch, int n_cu, int wr_full, HapPlan* plan, int* wr_out)
{
    const int K = (int)R.hap_count;
    uint64_t batches[MSQ_MAX_HAPS], total = 0;
    plan->n = K;
    for (int h = 0; h < K; ++h) {
        const uint64_t n_h = R.hap_reads[h];
        const uint64_t lo = hap_off[h] + j * n_h / n_launch, hi = hap_off[h] + (j + 1) * n_h / n_launch;
        plan->lo[h] = (uint32_t)lo;
        plan->hi[h] = (uint32_t)hi;
        batches[h] = (hi - lo + kRows - 1) / kRows;
        total += batches[h];
    }
    for (int h = K; h < MSQ_MAX_HAPS; ++h) plan->lo[h] = plan->hi[h] = 0;
    int wr = wr_full;
    if ((uint64_t)n_cu * wr > total) wr = (int)std::max<uint64_t>(1, std::min<uint64_t>(wr, (total + n_cu - 1) / n_cu));  // small inputs: thinner workgroups
    uint64_t p = std::max<uint64_t>((uint64_t)wr, (total + n_cu - 1) / n_cu);
    for (;; ++p) {
        uint64_t wgs = 0;
        for (int h = 0; h < K; ++h) wgs += (batches[h] + p - 1) / p;
        if (wgs <= (uint64_t)n_cu) break;
    }
    *wr_out = wr;
    if (p > (uint64_t)wr * kMaxBatchesPerWave) return false;
    int at = 0;
    for (int h = 0; h < K; ++h) {
        plan->wg_first[h] = at;
        at += (int)((batches[h] + p - 1) / p);
    }
    for (int h = K; h <= MSQ_MAX_HAPS; ++h) plan->wg_first[h] = at;
    return at > 0;
}

// The pileup of an aligned set inside [win_begin0, win_begin0 + W).  *handled = false: not this kernel's case (no guide,
// a window off the set's word grid, exotic symbols, QV planes a threshold would read) -- the
// caller takes the read-space kernels, for which an aligned set is a plain-layout set like any other.
// fixed_partials / fixed_bytes: msq_pileup_noalloc's built-in buffer for the workgroup partials (nothing is allocated).
// d_step_bounds / d_dev: msq_pileup_step -- the shard's bounds out of the same launch, deviations into the context's
// matrix, the fold STORES the caller's matrices.
int msq_pileup_rows_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                           uint32_t* d_nuc, uint32_t* d_codon, uint32_t* fixed_partials, size_t fixed_bytes, int32_t* d_step_bounds,
                           uint32_t* d_dev, bool* handled)
{
    *handled = false;
    if (!(R.flags & MSQ_READS_ALIGNED) || (R.flags & MSQ_READS_HAS_EXOTIC) || !R.guide || R.n_groups == 0 || W <= 0) return MSQ_OK;
    const uint8_t* planes[4] = {R.qual, R.delqv, R.subqv, R.insqv};
    const int32_t thrs[4] = {thr.qual, thr.del, thr.sub, thr.ins};
    for (int k = 0; k < 4; ++k)
        if (thrs[k] >= 0 && planes[k]) return MSQ_OK;
    const int64_t off = (int64_t)win_begin0 - R.align_origin;
    if (off % 32 != 0) return MSQ_OK;
    const int Gtot = (W + 31) / 32;
    // windows of more than 126 words go slab by slab: a launch loads, of every read, only the words of its slab (and the
    // one word behind it that the last codons reach into) -- no read is streamed twice, no partition is needed
    const int n_slabs = (Gtot + 2 * kLanesPerPass - 1) / (2 * kLanesPerPass);
    const int Gslab = (Gtot + n_slabs - 1) / n_slabs;
    if (d_step_bounds && (n_slabs != 1 || !R.ref_end)) return MSQ_OK;  // the fused step is one launch
    const bool many_n = (R.flags & MSQ_READS_QV_BAKED) != 0;
    const int NC = many_n ? 4 : 3;

    // msq_pileup_noalloc: partials that would not fit the built-in buffer -> the read-space kernel (7 rows per workgroup
    // instead of 9) takes the call
    if (fixed_partials && (size_t)ctx->n_cu * ((NC + 5) * Gslab * 16 * 4 + 8) > fixed_bytes) return MSQ_OK;

    RowsArgs a;
    memset(&a, 0, sizeof(a));
    a.ref_start = R.ref_start;
    a.ref_end = R.ref_end;
    a.group_off = R.group_off;
    a.bop = reinterpret_cast<const uint4*>(R.bop);
    a.guide = R.guide;
    a.guide_words = R.guide_words;
    a.win_begin0 = win_begin0;
    a.W = W;
    a.origin = R.align_origin;
    a.word0 = (int32_t)(off / 32);
    a.Gtot = Gtot;
    a.nuc = d_dev ? d_dev : d_nuc;
    a.codon = d_dev ? d_dev + (size_t)W * 6 : d_codon;
    a.err = ctx->d_err;

    rows_kernel_fn kern = pick_rows_kernel(many_n);
    bool marked = false;
    // haplotype groups (msq_reads_align): the workgroups are dealt out among the groups in proportion to their reads
    const bool haps = R.hap_count >= 2 && R.hap_count <= MSQ_MAX_HAPS && R.hap_perm && R.hap_guides && R.n_reads < 0xffffffffull;
    a.perm = haps ? R.hap_perm : nullptr;
    a.hap_guides = haps ? R.hap_guides : nullptr;
    uint64_t hap_off[MSQ_MAX_HAPS + 1] = {0};
    if (haps) {
        for (uint32_t h = 0; h < R.hap_count; ++h) hap_off[h + 1] = hap_off[h] + R.hap_reads[h];
        if (hap_off[R.hap_count] != R.n_reads) return msq_fail(MSQ_ERR_ARG, "msq_pileup: the haplotype groups do not add up to the set's reads");
    }
    // one launch + its fold: n_wg workgroups of wr waves per role; the reads lo .. hi (no groups) or the plan's places
    auto launch = [&](int slab, int G, int n_roles, int wr, int n_wg, uint64_t lo, uint64_t hi, const HapPlan* plan, bool last) -> int {
        const int w = wr * n_roles;
        const size_t stream_lds = (size_t)w * kQueueCap * 20 + 80 + (size_t)(G + 2) * 32 + (size_t)G * 128 * 3;
        const size_t slab_lds = (size_t)NC * wr * kLevels * 256;  // one role's levels in the flush
        const bool flush_wide = slab_lds * n_roles <= 160 * 1024 - 512;
        const size_t lds = std::max(stream_lds, flush_wide ? slab_lds * n_roles : slab_lds);
        if (lds > 160 * 1024 - 512) return msq_fail(MSQ_ERR_ARG, "msq_pileup: no LDS for a %d-word slab", G);
        a.flush_wide = flush_wide ? 1u : 0u;
        const size_t part_bytes = (size_t)n_wg * (NC + 5) * G * 16 * 4;
        const size_t need = part_bytes + (d_step_bounds ? (size_t)n_wg * 8 : 0);  // + the workgroups' shares of the bounds
        void* sc = fixed_partials;
        if (fixed_partials) {
            if (need > fixed_bytes)
                return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: %zu bytes of workgroup partials exceed the built-in %zu", need, fixed_bytes);
        } else {
            int rc = msq_scratch(ctx, need, &sc);
            if (rc != MSQ_OK) return rc;
        }
        a.partials = static_cast<uint32_t*>(sc);
        a.bounds_slots = d_step_bounds ? reinterpret_cast<uint32_t*>(static_cast<char*>(sc) + part_bytes) : nullptr;
        a.read_lo = lo;
        a.read_hi = hi;
        a.n_batches = (hi - lo + kRows - 1) / kRows;
        FoldRowsArgs fr;
        memset(&fr, 0, sizeof(fr));
        fr.n_haps = 1;
        fr.wg_first[0] = 0;
        for (int h = 1; h <= MSQ_MAX_HAPS; ++h) fr.wg_first[h] = n_wg;
        fr.guides = R.guide;
        if (plan) {
            a.hp = *plan;
            fr.n_haps = plan->n;
            for (int h = 0; h <= MSQ_MAX_HAPS; ++h) fr.wg_first[h] = plan->wg_first[h];
            fr.guides = R.hap_guides;
        }
        if (!marked) {
            msq_timing_mark_begin(ctx);
            marked = true;
        }
        if (int rc = msq_ensure_dynamic_lds(ctx->device, reinterpret_cast<const void*>(kern), lds)) return rc;
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(w * 64), lds, ctx->stream, a);
        if (last) msq_timing_mark_end(ctx);
        MSQ_HIP(hipGetLastError());
        const int slices = d_dev ? 1 : std::max(1, std::min(n_wg / (kFoldRowsSubs * kFoldRowsTrip), (ctx->n_cu / 2 + G - 1) / G));
        fr.partials = a.partials;
        fr.guide_words = R.guide_words;
        fr.n_wg = n_wg;
        fr.NC = NC;
        fr.G = G;
        fr.slab_w0 = a.slab_w0;
        fr.Gtot = Gtot;
        fr.word0 = a.word0;
        fr.W = W;
        fr.nuc = d_nuc;
        fr.codon = d_codon;
        fr.dev_nuc = d_dev ? a.nuc : nullptr;
        fr.dev_codon = d_dev ? a.codon : nullptr;
        fr.bounds_slots = a.bounds_slots;
        fr.bounds_out = d_step_bounds;
        hipLaunchKernelGGL(fold_rows_kernel, dim3(G, slices), dim3(1024), 0, ctx->stream, fr);
        MSQ_HIP(hipGetLastError());
        (void)slab;
        return MSQ_OK;
    };
    for (int slab = 0; slab < n_slabs; ++slab) {
        const int G = std::min(Gslab, Gtot - slab * Gslab);
        const int n_roles = G <= kLanesPerPass ? 1 : 2;
        a.G = G;
        a.slab_w0 = slab * Gslab;
        a.n_roles = n_roles;
        const int wr_full = (many_n ? kRowsWavesN : kRowsWaves) / n_roles;  // waves per role
        // every role sees every batch: a workgroup counts (waves of a role) x 127 batches
        const uint64_t reads_per_launch = (uint64_t)ctx->n_cu * wr_full * kMaxBatchesPerWave * kRows;
        if (d_step_bounds && R.n_reads > reads_per_launch) return marked ? msq_fail(MSQ_ERR_ARG, "msq_pileup_step: internal") : MSQ_OK;
        if (haps) {
            // launches: every group's reads in `n_launch` equal ranges, one range per launch; a launch's workgroups in
            // proportion to the ranges: p batches per workgroup, the smallest p that needs no more workgroups than CUs
            uint64_t n_launch = (R.n_reads + reads_per_launch - 1) / reads_per_launch;
            HapPlan plan;
            int wr = wr_full;
            for (;; ++n_launch) {
                bool fits = true;
                for (uint64_t j = 0; j < n_launch && fits; ++j) fits = plan_haps(R, hap_off, j, n_launch, ctx->n_cu, wr_full, &plan, &wr);
                if (fits) break;
                if (n_launch > R.n_reads) return msq_fail(MSQ_ERR_ARG, "msq_pileup: no launch plan for the haplotype groups");
            }
            if (d_step_bounds && n_launch != 1) return marked ? msq_fail(MSQ_ERR_ARG, "msq_pileup_step: internal") : MSQ_OK;
            for (uint64_t j = 0; j < n_launch; ++j) {
                plan_haps(R, hap_off, j, n_launch, ctx->n_cu, wr_full, &plan, &wr);
                if (int rc = launch(slab, G, n_roles, wr, plan.wg_first[plan.n], 0, R.n_reads, &plan, slab + 1 == n_slabs && j + 1 == n_launch)) return rc;
            }
            continue;
        }
        for (uint64_t lo = 0; lo < R.n_reads; lo += reads_per_launch) {
            const uint64_t hi = std::min<uint64_t>(R.n_reads, lo + reads_per_launch);
            const uint64_t n_batches = (hi - lo + kRows - 1) / kRows;
            int wr = wr_full, n_wg = ctx->n_cu;
            if ((uint64_t)n_wg * wr > n_batches) {  // small inputs: fewer, thinner workgroups
                wr = (int)std::max<uint64_t>(1, std::min<uint64_t>(wr, (n_batches + ctx->n_cu - 1) / ctx->n_cu));
                n_wg = (int)((n_batches + wr - 1) / wr);
            }
            if (int rc = launch(slab, G, n_roles, wr, n_wg, lo, hi, nullptr, slab + 1 == n_slabs && hi == R.n_reads)) return rc;
        }
    }
    *handled = true;
    return MSQ_OK;
}
