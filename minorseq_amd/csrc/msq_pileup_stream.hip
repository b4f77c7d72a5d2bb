// msq_pileup_stream.hip -- the streaming bit-plane pileup kernel (fast path of msq_pileup).
//
// What it computes is exactly MSAByRow::AddRead + MSAByColumn + CodonsAt for every column
// (src/MSA.cpp:55-92,126-136,144-219); how it computes it is built for gfx950:
//
//  * Every WAVE is an autonomous worker that streams whole reads, eight at a time (a "wave batch"), through a
//    private slice of LDS; there is no workgroup barrier in the steady state.  The eight reads of a batch are
//    contiguous in the packed format, so the batch is cut into chunks of 64 groups regardless of read borders:
//    every lane of every chunk load carries data (16 B of bit planes + 32 B per QV plane, coalesced).  The
//    chunk loads go through NS statically indexed register slots and are issued unconditionally (clamped
//    addresses instead of predicates), so the compiler can count them: NS-1 chunks stay in flight behind
//    `s_waitcnt vmcnt(n)` with n > 0 across every phase.
//  * Phase 1 (read space, lane = one 32-base group): the input planes become four column-space planes
//    V (QV-passing ACGT), X ('-' or 'N'), B0, B1 (nucleotide code; for X: B0 = 1 means 'N'); insertions / pads
//    are squeezed out of the 32-bit lane words (software pext, rare), the lane's column offset is a DPP prefix
//    sum of popcount(column-consuming ops) plus its read's column base (scalar bookkeeping per read border),
//    and the words are OR-ed at that bit offset into the read's LDS row (rows keep the four planes of a
//    32-column word side by side: one 64-bit LDS atomic per plane pair).
//  * Phase 2 (column space, lane = 32 columns): rows are compared against a GUIDE sequence 32 columns per
//    instruction.  Cells that match the guide -- nearly all of them in amplicon data -- are counted in
//    bit-sliced vertical counters held in registers (Harley-Seal carry-save adders); a (row, word) pair that
//    holds any deviating base / codon is pushed on a small per-wave LDS queue and the queue is drained with
//    one lane per entry, so the rare atomics do not serialise the wave.  When the last 64-word pass is at
//    most half full, its lanes split the batch's rows in two halves.
//  * At the end each workgroup reduces its waves' counters through LDS and stores one partial; a small kernel
//    folds the partials into the count matrices under each workgroup's guide symbols.
//
// The guide is only a speed hint (per-column majority base of the workgroup's first reads): any guide gives
// bit-exact counts, a bad one only sends more cells down the atomic path.
#include "msq_internal.h"
#include "msq_stream_common.h"

#include <algorithm>
#include <type_traits>
#include <vector>

// MSQ_EXP: compile-time timing experiments only (make EXTRA=-DMSQ_EXP=n); any non-zero value gives WRONG counts.
// bit 0 skip the LDS ORs of phase 1, bit 1 skip phase 2 + drain, bit 2 skip the squeeze, bit 3 skip the drain,
// bit 4 skip zero_rows, bit 5 skip the counter transposition of the flush.  The shipped library is always built with 0.
#ifndef MSQ_NO_CARRIER
#define MSQ_NO_CARRIER 0  // experiment builds: any base a read carries can be the runner-up of its column
#endif
#ifndef MSQ_NO_IDX
#define MSQ_NO_IDX 0  // experiment builds: ignore the sets' column index
#endif
#ifndef MSQ_NO_ALT  // experiment builds: every deviation from the guide pays a global atomic, as before the alt haplotype
#define MSQ_NO_ALT 0
#endif
#ifndef MSQ_W12
#define MSQ_W12 0
#endif
#ifndef MSQ_EXP
#define MSQ_EXP 0
#endif

namespace {

// Waves per workgroup (one workgroup per CU): 12 = 3 per SIMD where the kernel fits 168 VGPRs, else 8 = 2 per SIMD.
constexpr int max_waves_of(int np, int passes) { return ((passes == 1 && np <= 2) || (MSQ_W12 && np == 0)) ? 12 : 8; }
// Register slots of the chunk pipeline (NS - 1 chunk loads in flight behind the chunk being decoded).
constexpr int slots_of(int np, int /*passes*/) { return np == 0 ? MSQ_NS0 : np == 1 ? 3 : 2; }

// One column slab of a MERGED launch (pileup_kernel<.., MERGED = true>): a launch over several slabs at once, each with
// a share of the workgroups, so that a wide window pays the per-launch costs (dispatch, the guide, the flush: ~30 us)
// once and not once per slab.  The slab's read set (its own with a partition, else the same for all), its geometry,
// its partials and its workgroups [wg_first, wg_first + wg_count).
struct SlabDesc {
    const int32_t* ref_start;
    const uint64_t* group_off;
    const uint4* bop;
    const uint32_t* gcol;
    uint32_t* partials;  // [wg_count][NC + 4][G*32]
    uint64_t read_lo, read_hi, n_batches;  // this launch's reads of the slab's set
    int32_t slab0, G;
    uint32_t wg_first, wg_count;
};

struct StreamArgs {
    const int32_t* ref_start;
    const uint64_t* group_off;
    const uint4* bop;
    const uint4* qv[4];  // the QV planes that can mask a base (two uint4 per group), compacted
    int32_t thr[4];      // their thresholds
    uint64_t read_lo, read_hi;  // this launch covers reads [read_lo, read_hi)
    uint64_t n_batches;         // ceil((read_hi - read_lo) / 8)
    int32_t win_begin0, W;
    const int32_t* d_bounds;  // NULL, or {-begin0, end0} in device memory (msq_window_bounds_device): the window is
                              // read from there and W above is the capacity of the count matrices
    int32_t slab0;       // first window column of this launch's column slab (a multiple of 32)
    int32_t G;           // 32-column words of the slab (<= 128)
    int32_t RD;          // words per LDS row: a margin word, the slab's G words, two margin words, a zero pad word
    uint32_t normalized; // MSQ_READS_NORMALIZED: consuming bases lead every group, no compaction needed
    uint32_t alt_on;     // the launch has LDS for the alt-haplotype counters (one u32 per slab column)
    uint32_t carrier;    // alt haplotype by the carrier rule (learn_alt_words: ~4 us per launch, for launches that are long enough)
    uint32_t flush_wide; // the LDS holds the counter levels of all passes at once (flush in one round)
    uint32_t* nuc;
    uint32_t* codon;
    uint32_t* partials;  // [workgroups][NC + 4][G*32]
    uint32_t* err;
    // the phasing gather (pileup_kernel<.., PHASE = true>): window positions whose codon starts in this slab
    const int32_t* ph_pos;  // [ph_n] CodonAt arguments (window columns)
    const int32_t* ph_idx;  // [ph_n] their index k in the caller's position list
    int32_t ph_n, ph_total; // positions of this launch, of the caller's list (the key row length)
    uint16_t* ph_keys;      // [n_reads][ph_total]
    // msq_pileup_step: the shard's bounds come out of this launch too (NULL otherwise)
    const uint32_t* gcol;   // [n_groups] (kernels with IDX): first reference column of the group | (read index & 7) << 29
    const int32_t* ref_end;
    uint32_t* bounds_acc;   // {max of ~begin, max of end, workgroups done}, order-encoded; left at zero
    int32_t* bounds_out;    // {-begin0, end0}
    // MERGED kernels: the slabs of the launch; the per-slab fields above are taken from here
    uint32_t n_merged;
    SlabDesc slabs[kMaxMerged];
};

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
    uint32_t holes = keep ? ~keep & (0xffffffffu >> __clz(keep)) : 0u;
    while (holes) {
        const int j = 31 - __clz(holes);  // highest hole first: lower hole positions stay valid
        const uint32_t low = __builtin_amdgcn_ubfe(0xffffffffu, 0u, (uint32_t)j);  // bits below j
        p0 = (p0 & low) | ((p0 >> 1) & ~low);
        p1 = (p1 & low) | ((p1 >> 1) & ~low);
        p2 = (p2 & low) | ((p2 >> 1) & ~low);
        p3 = (p3 & low) | ((p3 >> 1) & ~low);
        holes &= low;
    }
}

__device__ __forceinline__ void lds_or64(uint64_t* p, uint64_t v)
{
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// bit-sliced compare: a > b, per column
template <int L>
__device__ __forceinline__ uint32_t sliced_gt(const uint32_t (&a)[L], const uint32_t (&b)[L])
{
    uint32_t gt = 0, eq = ~0u;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        gt |= eq & a[l] & ~b[l];
        eq &= ~(a[l] ^ b[l]);
    }
    return gt;
}

// Rank the four bases of every column of a word by their bit-sliced counts: best = the most common base (the smaller
// code among equals), second = the most common of the OTHER bases that `ok` admits (0 where it admits none).
__device__ __forceinline__ void rank_bases(const uint32_t (&cnt)[4][8], const uint32_t (&ok)[4], uint32_t& any, uint32_t (&best)[4],
                                           uint32_t (&second)[4], uint32_t& has_alt)
{
    any = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int l = 0; l < 8; ++l) any |= cnt[b][l];
    // beats(i, j): base i ranks before base j -- the larger count, the smaller index among equals
    uint32_t beats[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const uint32_t later_wins = sliced_gt(cnt[j], cnt[i]);
            beats[j][i] = later_wins;
            beats[i][j] = ~later_wins;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t m = any;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j != i) m &= beats[i][j];
        best[i] = m;
    }
    has_alt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t m = any & ~best[i] & ok[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j != i) m &= best[j] | ~ok[j] | beats[i][j];  // ahead of every other admitted base
        second[i] = m;
        has_alt |= m;
    }
}

// The workgroup's guide, learnt from every wave's first batch (64 .. 96 reads): per column the most common base
// -> gl[G + 3][4] = (valid, code bit 0, code bit 1, 0) -- and the ALT haplotype gl2[G + 3][4] = (valid, bit 0, bit 1,
// has-a-runner-up): the runner-up's base where there is one, the guide's base elsewhere.  Cells that match the guide
// are counted in registers; cells that deviate from it but match the alt haplotype are counted in LDS (the drain);
// only what matches neither costs a global atomic.  Which runner-up counts:
//   cnt_tmp == nullptr: any base a read carries (decided here).
//   cnt_tmp != nullptr: decided later by learn_alt_words, which needs the counts: they are parked in cnt_tmp
//   ([32][G + 1] dwords, conflict-free) and gl2 is left alone.
// Runs once per workgroup; kept out of line so its registers do not add to the streaming loop's budget.  Rows are
// [8][RD][4 planes V X B0 B1], WV dwords per wave.
__device__ __noinline__ void learn_guide_words(const uint32_t* lds, uint32_t* gl, uint32_t* gl2, uint32_t* cnt_tmp, int NWV, int WV, int RD,
                                               int G, int tid, int T)
{
    const int ROWB = RD * 4;
    for (int cg = tid; cg <= G; cg += T) {  // word G = the first word behind the slab: codons reach into it
        uint32_t cnt[4][8];  // per base; 8 levels: up to 255 rows (12 waves x 8 = 96)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) cnt[b][l] = 0;
        for (int wv = 0; wv < NWV; ++wv) {
            uint32_t x[4][kRows];
#pragma unroll
            for (int i = 0; i < kRows; ++i) {
                const uint4 w = *reinterpret_cast<const uint4*>(lds + wv * WV + i * ROWB + (cg + 1) * 4);
                x[0][i] = w.x & ~w.z & ~w.w;  // only valid bases vote (the X cells reuse B0 for 'N')
                x[1][i] = w.x & w.z & ~w.w;
                x[2][i] = w.x & ~w.z & w.w;
                x[3][i] = w.x & w.z & w.w;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) counter_add8(cnt[b], x[b]);
        }
        uint32_t any, best[4], second[4], has_alt, seen[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            seen[b] = 0;
#pragma unroll
            for (int l = 0; l < 8; ++l) seen[b] |= cnt[b][l];
        }
        rank_bases(cnt, seen, any, best, second, has_alt);
        const uint32_t g0 = best[1] | best[3], g1 = best[2] | best[3];
        *reinterpret_cast<uint4*>(gl + cg * 4) = make_uint4(any, g0, g1, 0u);
        if (cnt_tmp) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int l = 0; l < 8; ++l) cnt_tmp[(b * 8 + l) * (G + 1) + cg] = cnt[b][l];
        } else {
            const uint32_t a0 = second[1] | second[3], a1 = second[2] | second[3];
            *reinterpret_cast<uint4*>(gl2 + cg * 4) = make_uint4(any, (has_alt & a0) | (~has_alt & g0), (has_alt & a1) | (~has_alt & g1), has_alt);
        }
    }
}

// The alt haplotype with the CARRIER rule (after the first batch's phase 2 has counted, per row, the bases that miss
// the guide -- rowdev[NWV * 8]): a runner-up base counts when at least two reads carry it, or when one of its reads
// is a carrier -- a read that misses the guide in kCarrierMin columns or more, i.e. a read of another haplotype, not
// a read with a sequencing error.  One read's error (3 % of the columns at 0.05 % errors among 64 reads) then no
// longer stands as the runner-up of its column, where it made every read of a real minor variant next to it miss
// the alt CODON and pay global atomics; and a minor variant seen in a single read still teaches all its columns.
constexpr uint32_t kCarrierMin = 6;
constexpr int kCarrierMinBatches = 16;
__device__ __noinline__ void learn_alt_words(const uint32_t* lds, const uint32_t* gl, uint32_t* gl2, uint32_t* cnt_tmp, const uint32_t* rowdev,
                                             int NWV, int WV, int RD, int G, int tid, int T)
{
    const int ROWB = RD * 4, lane = tid & 63, rows = NWV * kRows;  // rows <= 96
    const uint64_t car0 = __ballot(lane < rows && rowdev[min(lane, rows - 1)] >= kCarrierMin);
    const uint64_t car1 = __ballot(lane + 64 < rows && rowdev[min(lane + 64, rows - 1)] >= kCarrierMin);
    for (int cg = tid; cg <= G; cg += T) {
        uint32_t cnt[4][8];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                cnt[b][l] = cnt_tmp[(b * 8 + l) * (G + 1) + cg];
                cnt_tmp[(b * 8 + l) * (G + 1) + cg] = 0;  // the area becomes the alt counters: all zero
            }
        uint32_t ok[4];  // at least two reads
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            ok[b] = 0;
#pragma unroll
            for (int l = 1; l < 8; ++l) ok[b] |= cnt[b][l];
        }
        for (int half = 0; half < 2; ++half) {  // ... or a carrier's base
            uint64_t m = half ? car1 : car0;
            while (m) {
                const int r = 64 * half + (__ffsll((long long)m) - 1);
                m &= m - 1;
                const uint4 w = *reinterpret_cast<const uint4*>(lds + (r >> 3) * WV + (r & 7) * ROWB + (cg + 1) * 4);
                ok[0] |= w.x & ~w.z & ~w.w;
                ok[1] |= w.x & w.z & ~w.w;
                ok[2] |= w.x & ~w.z & w.w;
                ok[3] |= w.x & w.z & w.w;
            }
        }
        uint32_t any, best[4], second[4], has_alt;
        rank_bases(cnt, ok, any, best, second, has_alt);
        const uint32_t g0 = best[1] | best[3], g1 = best[2] | best[3];
        const uint32_t a0 = second[1] | second[3], a1 = second[2] | second[3];
        *reinterpret_cast<uint4*>(gl2 + cg * 4) = make_uint4(any, (has_alt & a0) | (~has_alt & g0), (has_alt & a1) | (~has_alt & g1), has_alt);
    }
}

template <int NP>
struct Slot {  // one chunk (64 groups) in flight
    uint4 w;
    uint32_t m;  // IDX kernels: the group's column index word
    uint4 qa[NP ? NP : 1], qb[NP ? NP : 1];
};

__device__ __forceinline__ uint4 buffer_load16(__amdgpu_buffer_rsrc_t r, uint32_t voffset, uint32_t soffset)
{
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voffset, (int)soffset, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// constant address space + uniform index = s_load (the arrays are read-only for the whole launch)
typedef const uint64_t __attribute__((address_space(4))) * const_u64_ptr;
typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;

// Which batches a wave takes.  The waves of a workgroup do not run at the same speed (of the two waves of a SIMD the
// older one wins the issue arbitration: with a fixed share per wave the older ones sat idle for the last ~13 % of
// the kernel), so they DRAW their batches from one LDS counter: the t-th draw of the workgroup is batch
// t * (workgroups of the launch) + (this workgroup).  A draw is an LDS atomic whose result is looked at one batch later; the ids of the
// wave's last few batches sit in one VGPR (lane s & 3 = the wave's s-th batch, ~0 = none left), where both the
// producer (up to two batches ahead) and the consumer find them.  A wave stops drawing at `cap` batches, what its
// bit-sliced counters hold.
struct BatchDraw {
    uint32_t* ctr;
    uint32_t stride, first, n_batches, cap;
    uint32_t ring;     // VGPR
    uint32_t pending;  // VGPR: lane 0 = the draw in flight
    uint32_t n_res, n_issued;

    __device__ __forceinline__ void draw(int lane)
    {
        pending = 0xffffffffu;
        if (n_issued < cap) {  // wave-uniform
            if (lane == 0) pending = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        n_issued = uni(n_issued + 1u);
    }
    __device__ __forceinline__ void resolve(int lane)
    {
        const uint32_t t = uni(pending);
        uint32_t bid = t == 0xffffffffu ? 0xffffffffu : t * stride + first;
        if (bid >= n_batches) bid = 0xffffffffu;
        ring = (uint32_t)lane == (n_res & 3u) ? uni(bid) : ring;
        n_res = uni(n_res + 1u);
    }
    __device__ __forceinline__ uint32_t id(uint32_t s) const { return uni((uint32_t)__builtin_amdgcn_readlane((int)ring, (int)(s & 3u))); }
    __device__ __forceinline__ void init(uint32_t* counter, uint32_t nb, uint32_t cap_, int lane, uint32_t wg, uint32_t nwg)
    {
        ctr = counter;
        stride = nwg;
        first = wg;
        n_batches = nb;
        cap = cap_;
        ring = 0xffffffffu;
        n_res = n_issued = 0;
        for (int k = 0; k < 3; ++k) {
            draw(lane);
            resolve(lane);
        }
        draw(lane);
    }
};

// Producer side: walks this wave's batches and yields the group range of the next chunk to load.  Every field is
// wave-uniform (SGPRs); the next batch's range is fetched one batch ahead.
struct ChunkProducer {
    const_u64_ptr go;
    uint64_t read_lo, read_hi;
    uint32_t n_batches, P;  // P: sequence number (BatchDraw) of the batch being issued
    uint64_t B0, nB0;
    uint32_t ng, nng, c, live, nlive;
    uint32_t nslots;  // chunk slots the current batch takes: its chunks rounded up to a multiple of the slot count
    uint64_t bop_base;
    uint64_t cur_base;  // the bit planes' buffer resource over the current batch: base address and bytes
    uint32_t cur_bytes;
    uint64_t idx_base, cur_ibase;  // the same for the group column index (4 bytes per group), when the set has one

    // buffer resource over the current batch's groups of one plane (2^shift bytes per group)
    template <typename T>
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const T* plane, int shift) const
    {
        const char* base = reinterpret_cast<const char*>(plane) + (B0 << shift);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)((live ? ng : 0u) << shift), 0x00020000);
    }
    __device__ __forceinline__ void batch_changed(uint32_t ns)
    {
        const uint32_t nch = max(1u, (ng + 63u) >> 6);
        nslots = uni((nch + ns - 1u) / ns * ns);
        cur_base = uni64(bop_base + (B0 << 4));
        cur_ibase = uni64(idx_base + (B0 << 2));
        cur_bytes = uni(live ? ng << 4 : 0u);  // zero records for a wave that has run out of batches: every load reads zeros
    }
    __device__ __forceinline__ void fetch(uint32_t bb, uint64_t& o0, uint32_t& n, uint32_t& alive)
    {
        o0 = 0;
        n = 0;
        alive = bb < n_batches ? 1u : 0u;
        if (alive) {
            const uint64_t r0 = read_lo + (uint64_t)bb * kRows;
            const uint64_t r1 = min(r0 + (uint64_t)kRows, read_hi);
            o0 = uni64(go[r0]);
            n = uni((uint32_t)(uni64(go[r1]) - o0));
        }
    }
    __device__ __forceinline__ void init(const BatchDraw& bd, const uint64_t* group_off, const void* bop, const void* gcol, uint64_t lo,
                                         uint64_t hi, uint32_t nb, uint32_t ns)
    {
        go = (const_u64_ptr)(uintptr_t)group_off;
        bop_base = (uint64_t)(uintptr_t)bop;
        idx_base = (uint64_t)(uintptr_t)gcol;
        read_lo = lo; read_hi = hi; n_batches = nb;
        P = 0;
        c = 0;
        fetch(bd.id(0), B0, ng, live);
        fetch(bd.id(1), nB0, nng, nlive);
        batch_changed(ns);
    }
    // after a chunk slot of a live batch has been issued; a batch takes a multiple of `ns` slots (the tail ones
    // carry no chunk), so that slot k of the register pipeline always holds chunks k, k + ns, ... of a batch
    __device__ __forceinline__ void advance(BatchDraw& bd, uint32_t ns, int lane)
    {
        c = uni(c + 1);
        if (c >= nslots) {
            P = uni(P + 1u);
            B0 = nB0;
            ng = nng;
            live = nlive;
            c = 0;
            bd.resolve(lane);  // the draw of a batch ago: sequence number P + 2
            fetch(bd.id(P + 1u), nB0, nng, nlive);
            bd.draw(lane);
            batch_changed(ns);
        }
    }
};

// Consumer side: the same batch sequence, with what the decode needs per read.  The per-read tables of a batch
// live in two VGPRs (lane i = read i of the batch): bqv = group offset, relative to the batch's first group,
// where read i begins (lanes 1..7; every other lane holds ~0), rqv = its start column.  The decode walks the
// read borders with a scalar cursor (bi, bnext) and v_readlane.  The next batch's raw entries come through two
// plain vector loads issued when the current batch's chunks are done.
template <bool IDX>
struct BatchConsumer {
    const uint32_t* go32;  // group_off viewed as dwords (a batch spans < 2^32 groups: low dwords suffice)
    const int32_t* rs;
    uint64_t read_lo, read_hi;
    uint32_t n_batches, Cn, b;  // Cn: sequence number (BatchDraw) of the batch being decoded, b: its batch index
    int32_t win;
    uint32_t live;
    uint32_t bqv;        // VGPR
    int32_t rqv;         // VGPR
    uint32_t bi, bnext;  // next border: read index, group offset (~0 when none is left)
    uint32_t rel8, nchunks, c, rowoff_cur, C;
    int32_t cb_cur;      // column base of the read being decoded: column = cb_cur + C + lane's exclusive prefix
    uint32_t n_go;       // VGPR, prefetched: lane i = low dword of group_off[first read + min(i, 8)]
    int32_t n_rs;        // VGPR, prefetched: lane i = ref_start[first read + min(i, 7)]

    __device__ __forceinline__ void fetch(uint32_t bb, int lane)  // unconditional loads (clamped indices)
    {
        const uint64_t r0 = read_lo + (uint64_t)min(bb, n_batches - 1u) * kRows;
        n_go = go32[2 * min(r0 + (uint64_t)min(lane, kRows), read_hi)];
        if (!IDX) n_rs = rs[min(r0 + (uint64_t)min(lane, kRows - 1), read_hi - 1)];
    }
    __device__ __forceinline__ void begin(const BatchDraw& bd, int lane)  // make the prefetched batch current
    {
        b = bd.id(Cn);
        live = b < n_batches ? 1u : 0u;
        if (live) {
            const uint32_t rel = n_go - (uint32_t)__builtin_amdgcn_readlane((int)n_go, 0);
            rel8 = (uint32_t)__builtin_amdgcn_readlane((int)rel, kRows);
            if (!IDX) {
            bqv = (lane >= 1 && lane < kRows) ? rel : 0xffffffffu;
            rqv = n_rs - win;  // MSA.cpp:148: pos = ReferenceStart - beginPos
            cb_cur = __builtin_amdgcn_readlane(rqv, 0);
            bi = 1;
            bnext = (uint32_t)__builtin_amdgcn_readlane((int)bqv, 1);
            }
            nchunks = uni(max(1u, (rel8 + 63u) >> 6));
            c = 0;
            rowoff_cur = 0;
            C = 0;
        }
    }
    __device__ __forceinline__ void prefetch_next(const BatchDraw& bd, int lane) { fetch(bd.id(Cn + 1u), lane); }
    __device__ __forceinline__ void init(const BatchDraw& bd, const uint64_t* group_off, const int32_t* ref_start, uint64_t lo, uint64_t hi,
                                         uint32_t nb, int32_t win_begin0, int lane)
    {
        go32 = reinterpret_cast<const uint32_t*>(group_off);
        rs = ref_start;
        read_lo = lo; read_hi = hi; n_batches = nb;
        win = win_begin0;
        Cn = 0;
        fetch(bd.id(0), lane);
        begin(bd, lane);
    }
    __device__ __forceinline__ void next(const BatchDraw& bd, int lane)
    {
        Cn = uni(Cn + 1u);
        begin(bd, lane);
    }
};

// PHASE = false: the pileup.  PHASE = true: phase 1 only, then the per-read gather of AminoAcidCaller::PhaseVariants
// (MSARow::CodonAt at the requested positions, src/AminoAcidCaller.cpp:149-160) instead of the counting.
// MANY_N: Ns are common (a QV mask, applied here or baked into the planes at pack time): they get a counter of
// their own instead of going down the deviation path.
// IDX: the set carries the group column index (msq_reads_index), see phase 1.
#ifdef MSQ_WGT  // timing experiment: when each workgroup of the last launch started [wg], when the last of its waves
// ended [1024 + wg] and when the last of its waves left the batch loop [2048 + wg] (100 MHz); read back with
// msq_debug_wgt (tools/wgt.py)
static __device__ unsigned long long g_wgt[6 * 1024];  // + wave 0: [3072 + wg] set up, [4096 + wg] first batch in the rows, [5120 + wg] guide learnt
extern "C" int msq_debug_wgt(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), sizeof(g_wgt)); }
#endif
// MERGED: the launch covers several column slabs (StreamArgs::slabs), this workgroup works on one of them.
template <int NP, int PASSES, bool PHASE, bool MANY_N = (NP != 0), bool IDX = false, bool MERGED = false>
__global__ __launch_bounds__(max_waves_of(NP, PASSES) * 64) void pileup_kernel(StreamArgs a)
{
    constexpr int NS = slots_of(NP, PASSES);
    constexpr bool COUNT_N = MANY_N;
    constexpr int NC = COUNT_N ? 4 : 3;
    extern __shared__ __align__(16) uint32_t lds[];
#ifdef MSQ_TRACE  // timing experiment: phase timestamps (100 MHz) of a few waves, printed at the end
    uint64_t tr_[8];
    int tr_n = 0;
    uint32_t n_glob_ = 0, n_alt_ = 0;  // this lane's global / alt-counter atomics of the deviation path
    uint64_t ta_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = 0;  // per-batch segments, summed: chunks, phase 2, drain, zero rows, vmcnt(0), next
#define MSQ_TR() (tr_[tr_n++] = wall_clock64())
#define MSQ_TA0() (tl_ = wall_clock64())
#define MSQ_TA(k) do { const uint64_t t_ = wall_clock64(); ta_[k] += t_ - tl_; tl_ = t_; } while (0)
#else
#define MSQ_TR() ((void)0)
#define MSQ_TA0() ((void)0)
#define MSQ_TA(k) ((void)0)
#endif
    MSQ_TR();
#ifdef MSQ_WGT
    if (!PHASE && threadIdx.x == 0 && blockIdx.x < 1024) {
        g_wgt[blockIdx.x] = wall_clock64();
        g_wgt[1024 + blockIdx.x] = 0;
        g_wgt[2048 + blockIdx.x] = 0;
    }
#endif
    uint32_t wg = blockIdx.x, nwg = gridDim.x;  // this workgroup among those that share its reads
    if constexpr (MERGED) {
        // the table sits in the kernel arguments (this kernel's only parameter, at offset 0 of the segment): read it
        // through the constant address space, i.e. with scalar loads at a scalar index
        typedef const SlabDesc __attribute__((address_space(4))) * const_desc_ptr;
        typedef const char __attribute__((address_space(4))) * const_char_ptr;
        const const_desc_ptr tab = (const_desc_ptr)((const_char_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(StreamArgs, slabs));
        uint32_t sl = 0;
        while (sl + 1 < a.n_merged && wg >= tab[sl + 1].wg_first) ++sl;
        const const_desc_ptr d = tab + sl;
        a.ref_start = d->ref_start;
        a.group_off = d->group_off;
        a.bop = d->bop;
        a.gcol = d->gcol;
        a.partials = d->partials;
        a.read_lo = d->read_lo;
        a.read_hi = d->read_hi;
        a.n_batches = d->n_batches;
        a.slab0 = d->slab0;
        a.G = d->G;
        a.RD = d->G + 4;
        wg -= d->wg_first;
        nwg = d->wg_count;
    }
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, NWV = T >> 6;
    const int wave = (int)uni((uint32_t)tid >> 6);   // tell the compiler it is wave-uniform: all stream control is scalar
    const int RD = a.RD, G = a.G;
    uint32_t my_err = 0, exotic = 0;
    int W = a.W;
    int32_t win_begin0 = a.win_begin0;
    if (a.d_bounds) {  // the window as the bounds kernel (and the all-reduce over the shards) left it in HBM
        win_begin0 = -a.d_bounds[0];
        const int w = a.d_bounds[1] - win_begin0;
        // not the window the caller sized its matrices and plan for: another origin, or wider than the matrices
        if (win_begin0 != a.win_begin0 || w < 0 || w > W) my_err |= DEVERR_WINDOW;
        W = w < 0 ? 0 : w < W ? w : W;
    }
    const int slab0 = a.slab0, wbase = slab0 >> 5;            // window word of the slab's word 0
    const uint32_t slab_span = 32u * (uint32_t)(G + 1) + 31u;
    const int ROWB = RD * 4;                                  // dwords per row: [RD words][V X B0 B1]
    const int QW = ((kRows * G * 2 + 15) / 16) * 4;           // dwords of the deviation queue (u16 entries)
    const int WV = kRows * ROWB + QW;                         // dwords per wave
    uint32_t* const wrow = lds + wave * WV;                   // this wave's rows
    uint16_t* const queue = reinterpret_cast<uint16_t*>(wrow + kRows * ROWB);
    uint32_t* const draw_ctr = lds + NWV * WV;                // the workgroup's batch counter (BatchDraw)
    uint32_t* const gl = draw_ctr + 4;                        // the workgroup's guide words [G + 3][4]
    uint32_t* const gl2 = gl + 4 * (G + 3);                   // the alt haplotype [G + 3][4] (learn_guide_words)
    uint32_t* const altc = gl2 + 4 * (G + 3);                 // [(G + 1) * 32] per column: alt nucleotide count | alt codon count << 16
    uint32_t* const rowdev = altc + (G + 1) * 32;             // [96] per row of the first batches: bases that miss the guide (learn_alt_words)
    const bool alt_on = !PHASE && a.alt_on != 0;
    const bool carrier = alt_on && a.carrier != 0;

    // ---- phase-2 identity: in pass p this lane owns word cgp[p].  The last pass, when at most 32 words wide,
    // splits the batch's rows between the two halves of the wave (rows rbase .. rbase+3).
    const bool split = (G - 64 * (PASSES - 1)) <= 32;  // wave-uniform
    int cgp[PASSES];
    bool act[PASSES];
    uint32_t ccols[PASSES];
    uint32_t cnt[PASSES][NC][kLevels];
    const int rbase = split ? (lane >> 5) * 4 : 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        cgp[p] = (p == PASSES - 1 && split) ? 64 * p + (lane & 31) : 64 * p + lane;
        act[p] = cgp[p] < G;
        ccols[p] = act[p] ? codon_cols_of(wbase + cgp[p], W) : 0u;
    }

    auto zero_rows = [&]() {  // MSARow(size): all ' '
        for (int i = lane * 4; i < kRows * ROWB; i += 256) *reinterpret_cast<uint4*>(&wrow[i]) = make_uint4(0, 0, 0, 0);
    };
    zero_rows();
    if (tid == 0) *draw_ctr = 0;
    // (msq_pileup_step) this workgroup's share of the shard's reads for the window bounds: the loads go out first,
    // they are looked at once the first batch's chunk loads are under way
    int32_t b_lo = INT32_MAX, b_hi = INT32_MIN;
    const bool with_bounds = !PHASE && a.bounds_acc != nullptr;
    if (with_bounds) {
        if (tid < 2) draw_ctr[1 + tid] = 0;
#pragma unroll 4
        for (uint64_t i = a.read_lo + (uint64_t)blockIdx.x * T + tid; i < a.read_hi; i += (uint64_t)gridDim.x * T) {
            b_lo = min(b_lo, a.ref_start[i]);
            b_hi = max(b_hi, a.ref_end[i]);
        }
    }
    if (PHASE) {  // the guide area holds the position list instead: (window column, key index) pairs
        for (int i = tid; i < a.ph_n; i += T) {
            gl[2 * i] = (uint32_t)a.ph_pos[i];
            gl[2 * i + 1] = (uint32_t)a.ph_idx[i];
        }
    } else {
        for (int i = tid; i < 8 * (G + 3) + (alt_on ? (G + 1) * 32 + 96 : 0); i += T) gl[i] = 0;  // guide, alt haplotype, alt counters, rowdev
    }
    __syncthreads();
    BatchDraw bd;
    bd.init(draw_ctr, (uint32_t)a.n_batches, PHASE ? 0xffffffffu : (uint32_t)kMaxBatchesPerWave, lane, wg, nwg);

    // ---- producer: one unconditional chunk load per call (clamped addresses, no predicate around the loads)
    ChunkProducer pr;
    pr.init(bd, a.group_off, a.bop, a.gcol, a.read_lo, a.read_hi, (uint32_t)a.n_batches, NS);
    // The chunk loads are buffer loads: the resource covers exactly the producer's current batch (base = its first
    // group, num_records = its bytes; zero records for a wave that has run out of batches), the lane's byte offset is
    // a constant VGPR and the chunk's offset rides in the scalar operand -- no address arithmetic per load, and a
    // slot past the batch's last group (a padding slot, the tail of the last chunk) reads zeros by the range check.
    const uint32_t lane16 = (uint32_t)lane * 16u, lane32 = (uint32_t)lane * 32u;
    auto issue = [&](Slot<NP>& s) {
        const uint32_t soff = uni(pr.c * 1024u);
        s.w = buffer_load16(__builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)pr.cur_base, 0, (int)pr.cur_bytes, 0x00020000), lane16, soff);
        if (IDX)
            s.m = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)pr.cur_ibase, 0, (int)(pr.cur_bytes >> 2), 0x00020000), (int)lane * 4,
                (int)(soff >> 2), 0);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            s.qa[k] = buffer_load16(pr.rsrc(a.qv[k], 5), lane32, 2u * soff);
            s.qb[k] = buffer_load16(pr.rsrc(a.qv[k], 5), lane32 + 16u, 2u * soff);
        }
        if (pr.live) pr.advance(bd, NS, lane);
    };

    // ---- phase 1 for one chunk: read space -> column-space planes in LDS (MSAByRow::AddRead, MSA.cpp:144-186)
    BatchConsumer<IDX> cs;
    auto decode = [&](const Slot<NP>& s) {
        const uint32_t g0 = 64u * cs.c;
        const uint32_t nvalid = cs.rel8 > g0 ? min(cs.rel8 - g0, 64u) : 0u;
        const uint4 w = s.w;
        uint32_t pass = ~0u;  // ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87
#pragma unroll
        for (int k = 0; k < NP; ++k) pass &= qv_pass_mask(s.qa[k], s.qb[k], a.thr[k]);
        // ops 0, 1 consume a column; a lane past the batch's last group loaded zeros (the buffer's range check): with an
        // index its word says so (valid words are never 0), without one the lane number does
        const uint32_t cons = (IDX ? s.m != 0u : (uint32_t)lane < nvalid) ? ~w.w : 0u;
        const uint32_t is_m = cons & ~w.z;          // op 0: '=' / 'X'
        const uint32_t op1 = cons & w.z;            // op 1: '-', literal N, exotic
        exotic |= op1 & w.y;                        // exotic codes are the simple kernel's job (checked at the end)
        const uint32_t masked = is_m & ~pass;       // row char = 'N'          (MSA.cpp:165-166)
        uint32_t pv = is_m & pass;                  // row char = nucleotide   (MSA.cpp:163-164)
        uint32_t px = op1 | masked;                 // row char = '-' (MSA.cpp:168-171) or 'N'
        uint32_t b0 = (w.x & (pv | op1)) | masked;  // code bit 0; for X: 1 = 'N' (literal N is op 1 code 1)
        uint32_t b1 = w.y & pv;
        const int n = __popc(cons);
        if (!(MSQ_EXP & 4) && !a.normalized) squeeze4(cons, pv, px, b0, b1);  // wave-uniform
        uint32_t rowoffv;
        int c0;
        if constexpr (IDX) {
            // the set carries a column index (msq_reads_index): a group knows its first column and its read's row in the
            // batch -- no prefix sum over the wave, no walk over the read borders
            rowoffv = __umul24(((s.m >> 29) - ((uint32_t)a.read_lo & 7u)) & 7u, (uint32_t)ROWB);
            c0 = (int)(s.m & 0x1fffffffu) - 1 - win_begin0;
            cs.c = uni(cs.c + 1u);
        } else {
        const int incl = wave_inclusive_scan(n);
        const int excl = incl - n;
        // read borders inside this chunk: lanes from the border on belong to the next read / row
        rowoffv = cs.rowoff_cur;
        int32_t cbv = cs.cb_cur;
        while (cs.bnext < g0 + 64u) {  // wave-uniform
            const uint32_t sl = cs.bnext - g0;
            const uint32_t before = cs.C + (uint32_t)__builtin_amdgcn_readlane(excl, (int)sl);
            cs.rowoff_cur = uni(cs.rowoff_cur + (uint32_t)ROWB);
            cs.cb_cur = (int32_t)uni((uint32_t)__builtin_amdgcn_readlane(cs.rqv, (int)cs.bi) - before);
            const bool inr = (uint32_t)lane >= sl;
            rowoffv = inr ? cs.rowoff_cur : rowoffv;
            cbv = inr ? cs.cb_cur : cbv;
            cs.bi = uni(cs.bi + 1u);
            cs.bnext = uni((uint32_t)__builtin_amdgcn_readlane((int)cs.bqv, (int)cs.bi));  // lane 8 holds ~0: no border left
        }
        c0 = cbv + (int32_t)cs.C + excl;  // column of this lane's first base
        cs.C = uni(cs.C + (uint32_t)__builtin_amdgcn_readlane(incl, 63));
        cs.c = uni(cs.c + 1u);
        }
        if (n) {
            if (c0 < 0 || c0 + n > W) {
                my_err |= DEVERR_WINDOW;  // assert(pos >= 0) / write past the row, MSA.cpp:149
            } else if ((uint32_t)(c0 - slab0 + 31) < slab_span) {  // touches the slab or its margin words (-32 < cs0 < 32 (G + 1))
                // the lane's words at bit offset sh of slab word cs0 >> 5 (row word + 1; -1 = the leading margin):
                // low part, and the spill into the next word
                const int cs0 = c0 - slab0;
                // byte address of row word (cs0 >> 5) + 1: ((cs0 >> 5) + 1) * 16 = ((cs0 >> 1) & ~15) + 16
                uint64_t* p = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(wrow) + (rowoffv << 2) + (((cs0 >> 1) & ~15) + 16));
                // a plane rotated left by the bit offset holds its low part above the offset and its spill below:
                // one funnel shift and two masks per plane instead of three shifts
                const uint32_t keep_lo = 0xffffffffu << (cs0 & 31);
                auto rot = [&](uint32_t x) { return __builtin_amdgcn_alignbit(x, x, (uint32_t)(-cs0)); };  // the shift takes -cs0 mod 32
                const uint32_t rv = rot(pv), rx = rot(px), r0 = rot(b0), r1 = rot(b1);
                const uint64_t lo_vx = (uint64_t)(rv & keep_lo) | ((uint64_t)(rx & keep_lo) << 32);
                const uint64_t lo_b = (uint64_t)(r0 & keep_lo) | ((uint64_t)(r1 & keep_lo) << 32);
                const uint64_t hi_vx = (uint64_t)(rv & ~keep_lo) | ((uint64_t)(rx & ~keep_lo) << 32);
                const uint64_t hi_b = (uint64_t)(r0 & ~keep_lo) | ((uint64_t)(r1 & ~keep_lo) << 32);
                if (MSQ_EXP & 1) {  // timing experiment: keep the operands alive, leave out the LDS atomics
                    if (((lo_vx ^ lo_b ^ hi_vx ^ hi_b) + (uint64_t)(uintptr_t)p) == 0x123456789abcdef1ull) my_err |= DEVERR_RANGE;
                } else {
                    lds_or64(p, lo_vx);
                    lds_or64(p + 1, lo_b);
                    lds_or64(p + 2, hi_vx);
                    lds_or64(p + 3, hi_b);
                }
            }
        }
    };

    // ---- phase 2: column space (MSAByColumn ctor MSA.cpp:55-92 + CodonsAt MSA.cpp:126-136,197-219)
    uint32_t qn = 0;  // entries on the deviation queue (wave-uniform)
    // four consecutive rows of one word: match / gap / codon / N masks into levels 0-1 of the counters
    // (returns the weight-4 carries); (row, word) pairs with deviations go on the queue
    auto rows4 = [&](const int p, const int wofs, const uint32_t* rp, const int rowb, const uint32_t tag, uint32_t (&carry)[NC]) {
        uint32_t even[NC], twos[NC];  // streaming Harley-Seal: rows fold in pairs as they come
        // the guide words of this lane's columns (valid, code bit 0, code bit 1) and of the next word: re-read from
        // LDS per call, cheaper than twelve more live registers
        const uint4 ga = *reinterpret_cast<const uint4*>(gl + wofs), gb = *reinterpret_cast<const uint4*>(gl + wofs + 4);
        // with registers to spare (no QV planes in the load slots) all four rows are fetched up front: the LDS
        // latency is paid once per call, not once per row (the queue write between rows pins the later loads)
        constexpr bool FETCH_AHEAD = NP == 0 && (MSQ_FA || PASSES == 1);
        uint4 lo4[FETCH_AHEAD ? 4 : 1], hi4[FETCH_AHEAD ? 4 : 1];
        if (FETCH_AHEAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo4[i] = *reinterpret_cast<const uint4*>(rp + i * rowb);
                hi4[i] = *reinterpret_cast<const uint4*>(rp + i * rowb + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 lo = FETCH_AHEAD ? lo4[i] : *reinterpret_cast<const uint4*>(rp + i * rowb);
            const uint4 hi = FETCH_AHEAD ? hi4[i] : *reinterpret_cast<const uint4*>(rp + i * rowb + 4);
            // valid base equal to the guide base
            const uint32_t m_lo = lo.x & ga.x & ~((lo.z ^ ga.y) | (lo.w ^ ga.z));
            const uint32_t m_hi = hi.x & gb.x & ~((hi.z ^ gb.y) | (hi.w ^ gb.z));
            // codon at column c = columns c, c+1, c+2
            const uint32_t cm = m_lo & funnel(m_lo, m_hi, 1) & funnel(m_lo, m_hi, 2) & ccols[p];
            const uint32_t cv = lo.x & funnel(lo.x, hi.x, 1) & funnel(lo.x, hi.x, 2) & ccols[p];
            const uint32_t n_lo = lo.y & lo.z;
            uint32_t x[NC];
            x[0] = m_lo;
            x[1] = lo.y & ~lo.z;
            x[2] = cm;
            if (COUNT_N) x[NC - 1] = n_lo;
            const uint32_t dev = (lo.x & ~m_lo) | (cv & ~cm) | (COUNT_N ? 0u : n_lo);
            const bool has = dev != 0;
            const uint64_t bal = __ballot(has);
            if (bal) {  // wave-uniform, rare
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, qn));
                if (has) queue[pos] = (uint16_t)(tag + ((uint32_t)i << 8));
                qn = uni(qn + (uint32_t)__popcll(bal));
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (i == 0 || i == 2) {
                    even[k] = x[k];
                } else if (i == 1) {
                    csa(cnt[p][k][0], even[k], x[k], cnt[p][k][0], twos[k]);
                } else {
                    uint32_t t2;
                    csa(cnt[p][k][0], even[k], x[k], cnt[p][k][0], t2);
                    csa(cnt[p][k][1], twos[k], t2, cnt[p][k][1], carry[k]);
                }
            }
        }
    };
    // carries into level FROM of every counter of pass p; the upper levels only when some carry gets there
    auto ripple_all = [&](const int p, auto from, uint32_t (&carry)[NC]) {
        constexpr int FROM = decltype(from)::value;
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            carry[k] = ripple<FROM, kLowLevels>(cnt[p][k], carry[k]);
            any |= carry[k];
        }
        if (__ballot(any != 0)) {
#pragma unroll
            for (int k = 0; k < NC; ++k) (void)ripple<kLowLevels, kLevels>(cnt[p][k], carry[k]);
        }
    };
    // every lane runs phase 2 (no predicate, so qn and the queue bookkeeping stay scalar): a lane without a word
    // in a pass reads the all-zero pad word of its rows (row word G + 3; guide word G + 1) and adds nothing
    auto phase2 = [&]() {
        // the row stride as the compiler must re-read it per batch: otherwise it hoists a dozen row addresses out
        // of the batch loop and, short of registers, spills them
        int rowb = ROWB;
        asm volatile("" : "+s"(rowb));
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int wofs = (act[p] ? cgp[p] : G + 1) * 4;         // guide word
            const int rofs = (act[p] ? cgp[p] + 1 : G + 3) * 4;     // row word
            if (p == PASSES - 1 && split) {
                uint32_t c4[NC];
                rows4(p, wofs, wrow + rbase * rowb + rofs, rowb, (uint32_t)cgp[p] | ((uint32_t)rbase << 8), c4);
                ripple_all(p, std::integral_constant<int, 2>{}, c4);
            } else {
                uint32_t c4a[NC], c4b[NC], c8[NC];
                rows4(p, wofs, wrow + rofs, rowb, (uint32_t)cgp[p], c4a);
                rows4(p, wofs, wrow + 4 * rowb + rofs, rowb, (uint32_t)cgp[p] | (4u << 8), c4b);
#pragma unroll
                for (int k = 0; k < NC; ++k) csa(cnt[p][k][2], c4a[k], c4b[k], cnt[p][k][2], c8[k]);
                ripple_all(p, std::integral_constant<int, 3>{}, c8);
            }
        }
    };
    // the queued (row, word) pairs, one lane each: deviations from the guide cost one global atomic each
    // (MSAColumn::IncCounts MSA.inl:66; ++codons[codon])
    auto drain = [&]() {
        // qn was advanced under phase 2's lane predicates: lane 0 owns a word in every pass and holds the total
        const uint32_t nq = uni(qn);
        for (uint32_t e = lane; e < nq; e += 64) {
            const uint32_t ent = queue[e];
            const int cg = (int)(ent & 0xffu), row = (int)(ent >> 8);
            const uint32_t* rp = wrow + row * ROWB + (cg + 1) * 4;
            const uint4 lo = *reinterpret_cast<const uint4*>(rp), hi = *reinterpret_cast<const uint4*>(rp + 4);
            const uint4 gl_lo = *reinterpret_cast<const uint4*>(gl + cg * 4), gl_hi = *reinterpret_cast<const uint4*>(gl + cg * 4 + 4);
            const uint32_t cc = codon_cols_of(wbase + cg, W);
            const uint32_t m_lo = lo.x & gl_lo.x & ~((lo.z ^ gl_lo.y) | (lo.w ^ gl_lo.z));
            const uint32_t m_hi = hi.x & gl_hi.x & ~((hi.z ^ gl_hi.y) | (hi.w ^ gl_hi.z));
            const uint32_t cm = m_lo & funnel(m_lo, m_hi, 1) & funnel(m_lo, m_hi, 2) & cc;
            const uint32_t cv = lo.x & funnel(lo.x, hi.x, 1) & funnel(lo.x, hi.x, 2) & cc;
            const uint32_t nd = lo.x & ~m_lo, cd = cv & ~cm, nn = COUNT_N ? 0u : (lo.y & lo.z);
            const int cbase = (wbase + cg) * 32;
            // deviations that match the workgroup's alt haplotype -- a minor variant carried by a share of the reads --
            // are counted in LDS (returnless ds_add, ~1/100 of a global atomic's cost; the sum goes out with the
            // workgroup's partial); only what matches neither the guide nor the alt haplotype pays a global atomic
            uint32_t nd_alt = 0, cd_alt = 0;
            if (alt_on) {  // wave-uniform
                const uint4 al = *reinterpret_cast<const uint4*>(gl2 + cg * 4), ah = *reinterpret_cast<const uint4*>(gl2 + cg * 4 + 4);
                const uint32_t am_lo = lo.x & al.x & ~((lo.z ^ al.y) | (lo.w ^ al.z));
                const uint32_t am_hi = hi.x & ah.x & ~((hi.z ^ ah.y) | (hi.w ^ ah.z));
                nd_alt = nd & am_lo;
                cd_alt = cd & am_lo & funnel(am_lo, am_hi, 1) & funnel(am_lo, am_hi, 2);
                uint32_t bits = nd_alt | cd_alt;
                while (bits) {
                    const int j = __ffs(bits) - 1;
                    bits &= bits - 1;
                    __hip_atomic_fetch_add(&altc[cg * 32 + j], ((nd_alt >> j) & 1u) | (((cd_alt >> j) & 1u) << 16), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            // two tight loops instead of one that carries all three cases: a deviating base (or, without an N
            // counter, an N) is one atomic on its nucleotide bin, a deviating codon one on its codon bin
            uint32_t bits = (nd & ~nd_alt) | nn;  // disjoint: nd lies in V, nn in X
#ifdef MSQ_TRACE
            n_glob_ += __popc(bits) + __popc(cd & ~cd_alt);
            n_alt_ += __popc(nd_alt | cd_alt);
#endif
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t code = ((nn >> j) & 1u) ? 5u : (((lo.z >> j) & 1u) | (((lo.w >> j) & 1u) << 1));
                atomicAdd(&a.nuc[(uint32_t)(cbase + j) * 6u + code], 1u);
            }
            bits = cd & ~cd_alt;
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t t0 = funnel(lo.z, hi.z, j) & 7u, t1 = funnel(lo.w, hi.w, j) & 7u;
                const uint32_t idx = ((t1 & 1u) << 5) | ((t0 & 1u) << 4) | ((t1 & 2u) << 2) | ((t0 & 2u) << 1) |
                                     ((t1 & 4u) >> 1) | ((t0 & 4u) >> 2);
                atomicAdd(&a.codon[(uint32_t)(cbase + j) * 64u + idx], 1u);
            }
        }
        qn = 0;
    };

    // ---- phasing: MSARow::CodonAt (src/MSA.cpp:188-195) of the batch's eight rows at every listed position, as keys
    // s0 | s1 << 3 | s2 << 6 (0 ' ', 1..4 ACGT, 5 '-', 6 'N', 7 = index CodonAt leaves out); see msq_phase_codons
    auto gather = [&]() {
        const uint64_t r0 = a.read_lo + (uint64_t)cs.b * kRows;
        for (int it = lane; it < kRows * a.ph_n; it += 64) {
            const int row = it / a.ph_n, k = it - row * a.ph_n;
            const uint64_t r = r0 + (uint64_t)row;
            if (r >= a.read_hi) continue;
            const int p = (int)gl[2 * k];
            uint32_t key = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = p + j;
                uint32_t sym = 7u;
                if (c > 0 && c < W) {
                    const int cs0 = c - slab0, bit = cs0 & 31;
                    const uint4 w = *reinterpret_cast<const uint4*>(wrow + row * ROWB + ((cs0 >> 5) + 1) * 4);
                    const uint32_t v = (w.x >> bit) & 1u, x = (w.y >> bit) & 1u, b0 = (w.z >> bit) & 1u, b1 = (w.w >> bit) & 1u;
                    sym = v ? 1u + (b0 | (b1 << 1)) : x ? 5u + b0 : 0u;
                }
                key |= sym << (3 * j);
            }
            a.ph_keys[r * (uint64_t)a.ph_total + gl[2 * k + 1]] = (uint16_t)key;
        }
    };

    // ---- the workgroup's guide = per-column majority base over every wave's first batch
    auto learn_guide = [&]() {
        __syncthreads();  // every wave's first batch is in its rows
        learn_guide_words(lds, gl, gl2, carrier ? altc : nullptr, NWV, WV, RD, G, tid, T);
        __syncthreads();
    };
    // ---- the alt haplotype (two more workgroup barriers, met by every wave, before the counters come to life).  First
    // how far each read of this wave's first batch is from the guide: lane = word, as in phase 2, bases that miss the
    // guide's, summed per row with an LDS add where there are any
    auto learn_alt = [&](const bool have_batch) {
        if (have_batch) {
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                if (!act[p]) continue;
                const bool sp = p == PASSES - 1 && split;
                const uint4 g4 = *reinterpret_cast<const uint4*>(gl + cgp[p] * 4);
                for (int i = 0; i < (sp ? 4 : kRows); ++i) {
                    const int row = (sp ? rbase : 0) + i;
                    const uint4 lo = *reinterpret_cast<const uint4*>(wrow + row * ROWB + (cgp[p] + 1) * 4);
                    const uint32_t nd = lo.x & ~(g4.x & ~((lo.z ^ g4.y) | (lo.w ^ g4.z)));
                    if (nd) atomicAdd(&rowdev[wave * kRows + row], (uint32_t)__popc(nd));
                }
            }
        }
        __syncthreads();  // every wave's rowdev counts are in
        learn_alt_words(lds, gl, gl2, altc, rowdev, NWV, WV, RD, G, tid, T);
        __syncthreads();  // the alt words are there before any drain looks at them
    };

    // ---- the stream: slot k holds chunk k, k + NS, ...; a slot is refilled right after its chunk is decoded
    Slot<NP> sl[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) issue(sl[q]);
    cs.init(bd, a.group_off, a.ref_start, a.read_lo, a.read_hi, (uint32_t)a.n_batches, win_begin0, lane);
    // the batch's chunks, NS slots per trip in a fixed order: every wait for a slot sees the same NS - 1 younger
    // loads behind it
    auto chunks = [&]() {
        for (uint32_t k = 0; k < cs.nchunks; k += NS) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
#ifdef MSQ_TRACE
                MSQ_TA0();
                __builtin_amdgcn_s_waitcnt(0x0F70 | (NS - 1));  // vmcnt(NS - 1): the slot's load has arrived
                MSQ_TA(6);
#endif
                if (cs.c < cs.nchunks) decode(sl[q]);
                MSQ_TA(7);
                issue(sl[q]);
                MSQ_TA(8);
            }
        }
        cs.prefetch_next(bd, lane);
        wave_sync();
    };
    // The first batch goes into the rows before the guide exists; the guide is learnt from it (the only call and
    // the only workgroup barriers before the flush: a wave without reads meets them too), and only then do the
    // counters come to life, so nothing but the load slots is live across the call.
    bool have = cs.live;
    if (with_bounds) {  // wave maxima (order-preserving unsigned codes) meet in LDS; the barriers of learn_guide publish them
        uint32_t k_lo = ~((uint32_t)b_lo ^ 0x80000000u), k_hi = (uint32_t)b_hi ^ 0x80000000u;
        for (int d = 32; d >= 1; d >>= 1) {
            k_lo = max(k_lo, (uint32_t)__shfl_xor((int)k_lo, d));
            k_hi = max(k_hi, (uint32_t)__shfl_xor((int)k_hi, d));
        }
        if (lane == 0) {
            atomicMax(&draw_ctr[1], k_lo);
            atomicMax(&draw_ctr[2], k_hi);
        }
    }
    MSQ_TR();
#ifdef MSQ_WGT
    if (tid == 0 && blockIdx.x < 1024) g_wgt[3072 + blockIdx.x] = wall_clock64();
#endif
    if (have) chunks();
    MSQ_TR();
#ifdef MSQ_WGT
    if (tid == 0 && blockIdx.x < 1024) g_wgt[4096 + blockIdx.x] = wall_clock64();
#endif
    if (!PHASE) {
        learn_guide();
        if (carrier) learn_alt(have);
        if (with_bounds && tid == 0) {  // one atomic pair per workgroup; the last workgroup in publishes and clears
            atomicMax(&a.bounds_acc[0], draw_ctr[1]);
            atomicMax(&a.bounds_acc[1], draw_ctr[2]);
            __threadfence();
            if (atomicAdd(&a.bounds_acc[2], 1u) == gridDim.x - 1) {
                __threadfence();
                const uint32_t k_lo = atomicExch(&a.bounds_acc[0], 0u), k_hi = atomicExch(&a.bounds_acc[1], 0u);
                atomicExch(&a.bounds_acc[2], 0u);
                const int32_t begin0 = (int32_t)(~k_lo ^ 0x80000000u), end0 = (int32_t)(k_hi ^ 0x80000000u);
                a.bounds_out[0] = -begin0;
                a.bounds_out[1] = end0 > 0 ? end0 : 0;
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int k = 0; k < NC; ++k)
#pragma unroll
                for (int l = 0; l < kLevels; ++l) cnt[p][k][l] = 0;
    }
    MSQ_TR();
#ifdef MSQ_WGT
    if (tid == 0 && blockIdx.x < 1024) g_wgt[5120 + blockIdx.x] = wall_clock64();
#endif
    while (have) {  // a wave batch is complete in the rows
        MSQ_TA0();
        if (PHASE) {
            gather();
        } else {
            if (!(MSQ_EXP & 2)) phase2();
            wave_sync();
            MSQ_TA(1);
            if (!(MSQ_EXP & (2 | 8))) drain();
            else qn = 0;
            MSQ_TA(2);
        }
        wave_sync();
        if (!(MSQ_EXP & 16)) zero_rows();
        MSQ_TA(3);
        // gfx9 counts loads and returnless atomics / stores on one vmcnt; with both kinds pending the compiler must
        // assume out-of-order completion and drains to vmcnt(0) at every later wait.  Retire them here, once per
        // batch (behind the row clearing, which does not depend on them), so the chunk loads of the next batch
        // are waited for by count.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        wave_sync();
        MSQ_TA(4);
        cs.next(bd, lane);
        have = cs.live;
        MSQ_TA(5);
        if (have) chunks();
    }

    if (PHASE) {
        if (exotic) my_err |= DEVERR_EXOTIC;
        if (my_err) atomicOr(a.err, my_err);
        return;
    }
    // ---- flush: per column the guide's nucleotide / codon bins (meta); then the waves' bit-sliced counters are added
    // up while still bit-sliced (two bitop3 per level and counter, against some sixty instructions per counter to
    // transpose it), one wave per counter, and only the sum is transposed and stored as this workgroup's partial
    MSQ_TR();
#ifdef MSQ_WGT
    if (lane == 0 && blockIdx.x < 1024) atomicMax(&g_wgt[2048 + blockIdx.x], (unsigned long long)wall_clock64());
#endif
    __syncthreads();
    MSQ_TR();
    const int cols = G * 32;
    uint32_t* const part = a.partials + (size_t)wg * (NC + 4) * cols;
    // per column what the counters stand for: the guide's (plane set 0) and the alt haplotype's (plane set 1)
    // nucleotide and codon bins; then the alt counts the drain collected in LDS
    auto meta_of = [&](const uint32_t* g, int c) -> uint32_t {
        auto bit = [&](int plane, int cc) -> uint32_t { return (g[(cc >> 5) * 4 + plane] >> (cc & 31)) & 1u; };
        const uint32_t v0 = bit(0, c), v1 = bit(0, c + 1), v2 = bit(0, c + 2);
        const uint32_t k0 = bit(1, c) | (bit(2, c) << 1), k1 = bit(1, c + 1) | (bit(2, c + 1) << 1), k2 = bit(1, c + 2) | (bit(2, c + 2) << 1);
        return k0 | (v0 << 2) | (((k0 << 4) | (k1 << 2) | k2) << 8) | ((v0 & v1 & v2) << 14);
    };
    for (int c = tid; c < cols; c += T) {
        part[(size_t)NC * cols + c] = meta_of(gl, c);
        const uint32_t ac = alt_on ? altc[c] : 0u;
        part[(size_t)(NC + 1) * cols + c] = ac & 0xffffu;
        part[(size_t)(NC + 2) * cols + c] = ac >> 16;
        part[(size_t)(NC + 3) * cols + c] = meta_of(gl2, c);
    }
    // The kernels that carry QV planes through the loop sit at the register limit: they keep the plain flush (every
    // wave transposes its own counters and adds them into LDS accumulators); a run normally bakes the thresholds
    // at pack time and never gets there.
    constexpr bool SLICED_FLUSH = NP == 0;
    if constexpr (SLICED_FLUSH) {
        constexpr int SL = kLevels + 5;  // levels of a sum over <= 16 waves x 2 row halves
        uint32_t* const slab = lds;      // [passes of the round][NC][NWV][kLevels][64] (the launcher sizes the LDS; rows and guide are dead)
        // One round: the waves park the levels of passes [p_lo, p_hi) in LDS, then every (counter, pass) pair is one job
        // -- add the waves' entries up, transpose, store -- and the jobs are dealt out over the waves.  With room for
        // all passes at once (a.flush_wide) there is a single round and PASSES * NC jobs run side by side.
        auto round = [&](const int p_lo, const int p_hi) {
            __syncthreads();  // the guide words / the previous round's sums have been read
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
                if (p >= p_lo && p < p_hi) {
#pragma unroll
                    for (int k = 0; k < NC; ++k)
#pragma unroll
                        for (int l = 0; l < kLevels; ++l)
                            slab[((((p - p_lo) * NC + k) * NWV + wave) * kLevels + l) * 64 + lane] = cnt[p][k][l];
                }
            __syncthreads();
            for (int job = wave; job < NC * (p_hi - p_lo) && !(MSQ_EXP & 32); job += NWV) {  // wave-uniform
                const int k = job % NC, pj = p_lo + job / NC;
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    if (p != pj) continue;
                    // in a split pass lanes L and L + 32 counted different rows of the same word: two entries per wave
                    const bool sp = p == PASSES - 1 && split;
                    const bool mine = act[p] && (!sp || lane < 32);
                    const uint32_t* const base = slab + (size_t)((p - p_lo) * NC + k) * NWV * kLevels * 64;
                    uint32_t sum[SL];
#pragma unroll
                    for (int l = 0; l < SL; ++l) sum[l] = 0;
                    for (int e = 0; e < (sp ? 2 * NWV : NWV); ++e) {
                        const uint32_t* src = base + (sp ? e >> 1 : e) * kLevels * 64 + ((lane + (sp ? 32 * (e & 1) : 0)) & 63);
                        uint32_t carry = 0;
#pragma unroll
                        for (int l = 0; l < SL; ++l) csa(sum[l], l < kLevels ? src[l * 64] : 0u, carry, sum[l], carry);
                    }
                    if (mine) {
                        // bit-sliced sum -> 32 column counts: the levels are the rows of two 16 x 16 bit matrices side
                        // by side (columns 0..15 | 16..31), transposed together in four swap stages
                        static_assert(SL <= 16, "the transposition holds 16 levels");
                        uint32_t t16[16];
#pragma unroll
                        for (int l = 0; l < 16; ++l) t16[l] = l < SL ? sum[l] : 0u;
                        transpose16x2(t16);
                        uint4* dst = reinterpret_cast<uint4*>(part + (size_t)k * cols + cgp[p] * 32);
#pragma unroll
                        for (int r = 0; r < 16; r += 4) {
                            dst[r >> 2] = make_uint4(t16[r] & 0xffffu, t16[r + 1] & 0xffffu, t16[r + 2] & 0xffffu, t16[r + 3] & 0xffffu);
                            dst[4 + (r >> 2)] = make_uint4(t16[r] >> 16, t16[r + 1] >> 16, t16[r + 2] >> 16, t16[r + 3] >> 16);
                        }
                    }
                }
            }
        };
        if (a.flush_wide) {
            round(0, PASSES);
        } else {
            for (int p = 0; p < PASSES; ++p) round(p, p + 1);
        }
    } else {
        uint32_t* const acc = lds;  // [cols], over the rows (no wave needs them any more)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            for (int i = tid; i < cols; i += T) acc[i] = 0;
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                if (act[p] && !(MSQ_EXP & 32)) {
                    for (int j = 0; j < 32; ++j) {
                        uint32_t v = 0;
#pragma unroll
                        for (int l = 0; l < kLevels; ++l) v |= ((cnt[p][k][l] >> j) & 1u) << l;
                        if (v) atomicAdd(&acc[cgp[p] * 32 + j], v);
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < cols; i += T) part[(size_t)k * cols + i] = acc[i];
            __syncthreads();
        }
    }
    if (exotic) my_err |= DEVERR_EXOTIC;
    if (my_err) atomicOr(a.err, my_err);
#ifdef MSQ_WGT
    if (lane == 0 && blockIdx.x < 1024) atomicMax(&g_wgt[1024 + blockIdx.x], (unsigned long long)wall_clock64());
#endif
#ifdef MSQ_TRACE
    MSQ_TR();
    for (int d = 32; d >= 1; d >>= 1) {
        n_glob_ += __shfl_xor(n_glob_, d);
        n_alt_ += __shfl_xor(n_alt_, d);
    }
    // every workgroup (its wave 0): when it ended relative to its start, what its drains cost, how many deviations
    // went to global atomics and how many to the alt counters -- the kernel ends with its slowest workgroup
    if (lane == 0 && wave == 0)
        printf("wgt %3d total %6llu loop %6llu drain %6llu vmcnt0 %5llu glob %u alt %u\n", (int)blockIdx.x, (unsigned long long)(tr_[6] - tr_[0]),
               (unsigned long long)(tr_[4] - tr_[3]), (unsigned long long)ta_[2], (unsigned long long)ta_[4], n_glob_, n_alt_);
    if (lane == 0 && (wave == 0 || wave == NWV - 1) && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == gridDim.x - 1))
        printf("trace wg %3d wave %d: start %llu | init %4llu first-batch %4llu guide %4llu loop %5llu wait %4llu flush %4llu (x10 ns)\n", (int)blockIdx.x, wave,
               (unsigned long long)tr_[0], (unsigned long long)(tr_[1] - tr_[0]), (unsigned long long)(tr_[2] - tr_[1]),
               (unsigned long long)(tr_[3] - tr_[2]), (unsigned long long)(tr_[4] - tr_[3]), (unsigned long long)(tr_[5] - tr_[4]),
               (unsigned long long)(tr_[6] - tr_[5]));
    if (lane == 0 && (wave == 0 || wave == NWV - 1) && (blockIdx.x == 0 || blockIdx.x == 101 || blockIdx.x == gridDim.x - 1))
        printf("      wg %3d wave %d: per-batch sums: load-wait %6llu decode %6llu issue %6llu | phase2 %6llu drain %6llu zero %5llu vmcnt0 %5llu next %5llu (x10 ns)\n", (int)blockIdx.x, wave,
               (unsigned long long)ta_[6], (unsigned long long)ta_[7], (unsigned long long)ta_[8], (unsigned long long)ta_[1], (unsigned long long)ta_[2], (unsigned long long)ta_[3],
               (unsigned long long)ta_[4], (unsigned long long)ta_[5]);
#endif
}

typedef void (*pileup_kernel_fn)(StreamArgs);

pileup_kernel_fn pick_kernel(int np, int passes, bool many_n, bool idx)
{
    if (np == 0 && idx) {  // the kernels a run takes (thresholds baked or unset) come with and without the column index
        if (many_n) return passes == 1 ? pileup_kernel<0, 1, false, true, true> : pileup_kernel<0, 2, false, true, true>;
        return passes == 1 ? pileup_kernel<0, 1, false, false, true> : pileup_kernel<0, 2, false, false, true>;
    }
    if (np == 0 && many_n) return passes == 1 ? pileup_kernel<0, 1, false, true> : pileup_kernel<0, 2, false, true>;
    if (passes == 1) {
        return np == 0   ? pileup_kernel<0, 1, false>
               : np == 1 ? pileup_kernel<1, 1, false>
               : np == 2 ? pileup_kernel<2, 1, false>
                         : pileup_kernel<4, 1, false>;
    }
    return np == 0 ? pileup_kernel<0, 2, false> : np == 1 ? pileup_kernel<1, 2, false> : np == 2 ? pileup_kernel<2, 2, false> : pileup_kernel<4, 2, false>;
}
// the launches over several column slabs at once: the kernels a run takes (no QV planes in the loop)
pileup_kernel_fn pick_merged_kernel(int passes, bool many_n, bool idx)
{
    if (idx) {
        if (many_n) return passes == 1 ? pileup_kernel<0, 1, false, true, true, true> : pileup_kernel<0, 2, false, true, true, true>;
        return passes == 1 ? pileup_kernel<0, 1, false, false, true, true> : pileup_kernel<0, 2, false, false, true, true>;
    }
    if (many_n) return passes == 1 ? pileup_kernel<0, 1, false, true, false, true> : pileup_kernel<0, 2, false, true, false, true>;
    return passes == 1 ? pileup_kernel<0, 1, false, false, false, true> : pileup_kernel<0, 2, false, false, false, true>;
}
pileup_kernel_fn pick_phase_kernel(int np, bool idx)
{
    if (np == 0 && idx) return pileup_kernel<0, 1, true, false, true>;
    return np == 0 ? pileup_kernel<0, 1, true> : np == 1 ? pileup_kernel<1, 1, true> : np == 2 ? pileup_kernel<2, 1, true> : pileup_kernel<4, 1, true>;
}

}  // namespace

// One streaming pass over the reads per column slab: the pileup (phase == nullptr) or the phasing gather.
struct PhaseRequest {
    int32_t n_pos;
    const int32_t* pos;  // host
    uint16_t* d_keys;
};

// The column slabs of a W-column window: windows wider than 128 words (4096 columns) are cut into equal slabs of
// <= 128 words.  Slab s covers window columns [s * Gslab * 32, min(W, (s + 1) * Gslab * 32)).
// max_words: the widest slab wanted (1..128; a partition may ask for narrower slabs than the kernel's limit).
void msq_slab_layout(int32_t W, int max_words, int* n_slabs, int* Gslab)
{
    const int Gtot = (W + 31) / 32;
    const int cap = max_words < 1 ? 1 : max_words > 128 ? 128 : max_words;
    *n_slabs = Gtot ? (Gtot + cap - 1) / cap : 0;
    *Gslab = *n_slabs ? (Gtot + *n_slabs - 1) / *n_slabs : 0;
}

// msq_pileup_step: bounds, cleared matrices and the pileup from one kernel + its fold.  Only when the whole call is
// ONE launch (one column slab, every read in it); otherwise the launch reports "not handled" and the caller takes
// the separate calls.
struct FusedStep {
    int32_t* d_bounds;  // out: {-begin0, end0} of the reads
    uint32_t* d_dev;    // the context's deviation matrix [W * 70], all zero between calls
};

// fixed_partials: NULL, or a caller-provided buffer of fixed_bytes for the workgroup partials (msq_pileup_noalloc:
// nothing is allocated then, and a launch that needs more is refused).
// The pileup of a window of several column slabs as ONE launch + one fold per slab: every slab gets a share of the
// workgroups in proportion to its groups.  sets[s] = the read set of slab s (a partition's own set per slab, or the
// same set for every slab; NULL or empty = nothing to do for that slab).  *handled = false: not applicable (QV planes
// in the loop, slabs that need different kernels, more reads than one launch counts, inputs too small to share the
// chip) -- the caller takes one launch per slab.
static int g_merged_launches = 0;  // launches taken by stream_launch_merged in this process (tests ask: did the merged path run?)
extern "C" int msq_debug_merged_launches(void) { return g_merged_launches; }

static int stream_launch_merged(msq_ctx* ctx, const msq_packed_reads* const* sets, int n_sets, const msq_qv_thresholds& thr,
                                int32_t win_begin0, int32_t W, int slab_words, const int32_t* d_bounds, uint32_t* d_nuc, uint32_t* d_codon,
                                uint32_t* fixed_partials, size_t fixed_bytes, bool* handled)
{
    *handled = false;
    if (ctx->no_merge) return MSQ_OK;
    const int Gtot = (W + 31) / 32;
    int n_slabs = 0, Gslab = 0;
    msq_slab_layout(W, slab_words, &n_slabs, &Gslab);
    if (n_slabs < 2 || n_slabs != n_sets || n_slabs > kMaxMerged) return MSQ_OK;
    StreamArgs a;
    memset(&a, 0, sizeof(a));
    uint32_t flags = 0;
    bool have_flags = false, idx = !MSQ_NO_IDX && win_begin0 >= 0 && (int64_t)win_begin0 + W < 0x1fffff00;
    int passes = 0, n_live = 0;
    uint64_t total_groups = 0;
    const int32_t thrs[4] = {thr.qual, thr.del, thr.sub, thr.ins};
    for (int sl = 0; sl < n_slabs; ++sl) {
        const msq_packed_reads* R = sets[sl];
        if (!R || R->n_reads == 0 || R->n_groups == 0) continue;
        if (R->flags & MSQ_READS_HAS_EXOTIC) return MSQ_OK;
        const uint8_t* planes[4] = {R->qual, R->delqv, R->subqv, R->insqv};
        for (int k = 0; k < 4; ++k)
            if (thrs[k] >= 0 && planes[k]) return MSQ_OK;  // a QV plane in the loop: those kernels are a fallback, one launch per slab
        if (have_flags && R->flags != flags) return MSQ_OK;
        flags = R->flags;
        have_flags = true;
        idx = idx && R->gcol != nullptr;
        const int G = std::min(Gslab, Gtot - sl * Gslab);
        const int ps = G <= 64 ? 1 : 2;
        if (passes && ps != passes) return MSQ_OK;
        passes = ps;
        total_groups += R->n_groups;
        ++n_live;
    }
    if (n_live < 2) return MSQ_OK;
    const bool many_n = (flags & MSQ_READS_QV_BAKED) != 0;
    const int NC = many_n ? 4 : 3;
    // the LDS of the widest slab decides the waves per workgroup, as in a launch of its own
    const int G = Gslab, RD = G + 4;
    const size_t wave_lds = (size_t)kRows * RD * 16 + (size_t)((kRows * G * 2 + 15) / 16) * 16;
    const size_t guide_lds = 16 + (size_t)(G + 3) * 32;
    const int max_waves = max_waves_of(0, passes);
    const size_t alt_lds = ((size_t)(G + 1) * 32 + 96) * 4;
    const bool alt_on = !MSQ_NO_ALT && std::min<size_t>(max_waves, (160 * 1024 - guide_lds - alt_lds - 512) / wave_lds) ==
                                           std::min<size_t>(max_waves, (160 * 1024 - guide_lds - 512) / wave_lds);
    const size_t shared_lds = guide_lds + (alt_on ? alt_lds : 0);
    const int w = (int)std::min<size_t>(max_waves, (160 * 1024 - shared_lds - 512) / wave_lds);
    if (w < 1) return MSQ_OK;
    const size_t slab_lds = (size_t)NC * w * kLevels * 256;
    const bool flush_wide = slab_lds * passes <= 160 * 1024 - 512;
    const size_t lds = std::max((size_t)w * wave_lds + shared_lds, flush_wide ? slab_lds * passes : slab_lds);

    // workgroups per slab in proportion to its groups, at least one, n_cu in all
    const int n_wg = ctx->n_cu;
    if (n_wg < n_live) return MSQ_OK;
    int share[kMaxMerged], given = 0, biggest = -1;
    for (int sl = 0; sl < n_slabs; ++sl) {
        const msq_packed_reads* R = sets[sl];
        share[sl] = 0;
        if (!R || R->n_reads == 0 || R->n_groups == 0) continue;
        share[sl] = std::max(1, (int)((double)n_wg * (double)R->n_groups / (double)total_groups));
        given += share[sl];
        if (biggest < 0 || R->n_groups > sets[biggest]->n_groups) biggest = sl;
    }
    share[biggest] += n_wg - given;
    if (share[biggest] < 1) return MSQ_OK;
    const uint64_t min_batches = ctx->carrier_min_batches >= 0 ? (uint64_t)ctx->carrier_min_batches : (uint64_t)kCarrierMinBatches;
    // rounds: a wave's counters hold kMaxBatchesPerWave batches, so a slab with more reads than its workgroups can count
    // takes the reads in `rounds` equal ranges, one merged launch (+ folds) per range
    uint64_t rounds = 1;
    size_t need = 0;
    for (int sl = 0; sl < n_slabs; ++sl) {
        if (!share[sl]) continue;
        const uint64_t cap = (uint64_t)share[sl] * w * kMaxBatchesPerWave * kRows;
        rounds = std::max<uint64_t>(rounds, (sets[sl]->n_reads + cap - 1) / cap);
        need += (size_t)share[sl] * (NC + 4) * std::min(Gslab, Gtot - sl * Gslab) * 32 * 4;
    }
    bool carrier = !MSQ_NO_CARRIER;
    for (int sl = 0; sl < n_slabs; ++sl) {
        if (!share[sl]) continue;
        const uint64_t n_batches = (sets[sl]->n_reads + kRows - 1) / kRows / rounds;  // of the shortest round
        if (n_batches < (uint64_t)share[sl] * w) return MSQ_OK;  // too few batches to fill the slab's waves: one launch per slab
        carrier = carrier && n_batches >= min_batches * w * share[sl];
    }
    void* scratch_base = fixed_partials;
    if (fixed_partials) {
        if (need > fixed_bytes)
            return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: %zu bytes of workgroup partials exceed the built-in %zu", need, fixed_bytes);
    } else {
        int rc = msq_scratch(ctx, need, &scratch_base);
        if (rc != MSQ_OK) return rc;
    }
    a.win_begin0 = win_begin0;
    a.W = W;
    a.d_bounds = d_bounds;
    a.normalized = (flags & MSQ_READS_NORMALIZED) ? 1u : 0u;
    a.alt_on = alt_on ? 1u : 0u;
    a.carrier = carrier ? 1u : 0u;
    a.flush_wide = flush_wide ? 1u : 0u;
    a.nuc = d_nuc;
    a.codon = d_codon;
    a.err = ctx->d_err;
    for (int k = 0; k < 4; ++k) a.thr[k] = -1;
    pileup_kernel_fn kern = pick_merged_kernel(passes, many_n, idx);
    if (int rc = msq_ensure_dynamic_lds(ctx->device, reinterpret_cast<const void*>(kern), lds)) return rc;
    msq_timing_mark_begin(ctx);
    for (uint64_t round = 0; round < rounds; ++round) {
        uint32_t wg_first = 0;
        char* part = static_cast<char*>(scratch_base);  // the folds of a round have read the partials before the next round writes them (stream order)
        a.n_merged = 0;
        for (int sl = 0; sl < n_slabs; ++sl) {
            if (!share[sl]) continue;
            const msq_packed_reads* R = sets[sl];
            const uint64_t all_batches = (R->n_reads + kRows - 1) / kRows;
            SlabDesc& d = a.slabs[a.n_merged++];
            d.ref_start = R->ref_start;
            d.group_off = R->group_off;
            d.bop = reinterpret_cast<const uint4*>(R->bop);
            d.gcol = idx ? R->gcol : nullptr;
            d.partials = reinterpret_cast<uint32_t*>(part);
            d.read_lo = all_batches * round / rounds * kRows;  // whole batches: a round starts at a multiple of eight reads
            d.read_hi = std::min<uint64_t>(R->n_reads, all_batches * (round + 1) / rounds * kRows);
            d.n_batches = (d.read_hi - d.read_lo + kRows - 1) / kRows;
            d.slab0 = sl * Gslab * 32;
            d.G = std::min(Gslab, Gtot - sl * Gslab);
            d.wg_first = wg_first;
            d.wg_count = (uint32_t)share[sl];
            wg_first += d.wg_count;
            part += (size_t)share[sl] * (NC + 4) * d.G * 32 * 4;
        }
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(w * 64), lds, ctx->stream, a);
        if (round + 1 == rounds) msq_timing_mark_end(ctx);
        MSQ_HIP(hipGetLastError());
        FoldSlabs ft;
        memset(&ft, 0, sizeof(ft));
        int col_blocks = 0;
        for (uint32_t k = 0; k < a.n_merged; ++k) {
            const SlabDesc& d = a.slabs[k];
            ft.partials[k] = d.partials;
            ft.n_wg[k] = (int)d.wg_count;
            ft.cols[k] = d.G * 32;
            ft.col0[k] = d.slab0;
            col_blocks = std::max(col_blocks, (d.G * 32 + kFoldCols - 1) / kFoldCols);
        }
        hipLaunchKernelGGL(fold_slabs_kernel, dim3(col_blocks, 1, a.n_merged), dim3(kFoldCols * kFoldSubs), 0, ctx->stream, ft, NC, W, d_nuc, d_codon);
        MSQ_HIP(hipGetLastError());
    }
    ++g_merged_launches;
    *handled = true;
    return MSQ_OK;
}

static int stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                         const int32_t* d_bounds, uint32_t* d_nuc, uint32_t* d_codon, const PhaseRequest* phase, bool* handled,
                         int only_slab = -1, int slab_words = 128, uint32_t* fixed_partials = nullptr, size_t fixed_bytes = 0,
                         const FusedStep* fused = nullptr)
{
    *handled = false;
    if (R.flags & MSQ_READS_HAS_EXOTIC) return MSQ_OK;
    if (R.n_groups == 0) {
        *handled = only_slab >= 0;  // a slab's read set may well be empty
        return MSQ_OK;
    }
    // the reads are streamed once per slab (a launch places only the bases that fall into its slab and margins)
    const int Gtot = (W + 31) / 32;
    int n_slabs = 0, Gslab = 0;
    msq_slab_layout(W, slab_words, &n_slabs, &Gslab);

    // the QV planes that can mask a base (threshold set and plane present; ArrayRead.inl:67-87), compacted
    StreamArgs a;
    const uint8_t* planes[4] = {R.qual, R.delqv, R.subqv, R.insqv};
    const int32_t thrs[4] = {thr.qual, thr.del, thr.sub, thr.ins};
    int np = 0;
    for (int k = 0; k < 4; ++k)
        if (thrs[k] >= 0 && planes[k]) {
            a.qv[np] = reinterpret_cast<const uint4*>(planes[k]);
            a.thr[np] = thrs[k];
            ++np;
        }
    const int np_kernel = np == 3 ? 4 : np;
    for (int k = np; k < 4; ++k) {  // a kernel with more plane slots than planes re-applies the last one (idempotent)
        a.qv[k] = np ? a.qv[np - 1] : nullptr;
        a.thr[k] = np ? a.thr[np - 1] : -1;
    }
    const bool many_n = np_kernel != 0 || (R.flags & MSQ_READS_QV_BAKED) != 0;  // a QV mask makes Ns common
    const int NC = many_n ? 4 : 3;

    a.ref_start = R.ref_start;
    a.group_off = R.group_off;
    a.bop = reinterpret_cast<const uint4*>(R.bop);
    a.win_begin0 = win_begin0;
    a.W = W;
    a.d_bounds = d_bounds;
    a.normalized = (R.flags & MSQ_READS_NORMALIZED) ? 1u : 0u;
    a.nuc = d_nuc;
    a.codon = d_codon;
    a.err = ctx->d_err;

    constexpr int kMaxPhasePositions = 1024;  // per launch: 8 KB of LDS for the (column, index) pairs
    a.ph_pos = a.ph_idx = nullptr;
    a.ph_n = a.ph_total = 0;
    a.ph_keys = nullptr;
    a.gcol = R.gcol;
    // the QV-plane kernels (a fallback) keep the prefix sum; so do windows the index's 29 column bits cannot address
    const bool idx = R.gcol != nullptr && np_kernel == 0 && !MSQ_NO_IDX && win_begin0 >= 0 && (int64_t)win_begin0 + W < 0x1fffff00;
    a.ref_end = nullptr;
    a.bounds_acc = nullptr;
    a.bounds_out = nullptr;
    if (fused && (n_slabs != 1 || !R.ref_end)) return MSQ_OK;
    if (!phase && !fused && only_slab < 0 && n_slabs > 1 && n_slabs <= kMaxMerged) {  // every slab streams the same set: one launch for all
        const msq_packed_reads* same[kMaxMerged];
        for (int sl = 0; sl < n_slabs; ++sl) same[sl] = &R;
        int rc = stream_launch_merged(ctx, same, n_slabs, thr, win_begin0, W, slab_words, d_bounds, d_nuc, d_codon, fixed_partials, fixed_bytes, handled);
        if (rc != MSQ_OK || *handled) return rc;
    }
    bool marked = false;  // the timing slot opens with the first launch
    if (only_slab >= n_slabs) return msq_fail(MSQ_ERR_ARG, "msq_pileup_slab: slab %d of %d", only_slab, n_slabs);
    for (int slab = only_slab >= 0 ? only_slab : 0; slab < (only_slab >= 0 ? only_slab + 1 : n_slabs); ++slab) {
        const int G = std::min(Gslab, Gtot - slab * Gslab);
        const int passes = phase ? 1 : G <= 64 ? 1 : 2;
        const int RD = G + 4;
        a.slab0 = slab * Gslab * 32;
        a.G = G;
        a.RD = RD;
        // phasing: the positions whose codon starts in this slab (those left of the window go with the first slab,
        // those right of it with the last: all their symbols are "left out" anyway), at most kMaxPhasePositions per launch
        std::vector<int32_t> ph_pos, ph_idx;
        if (phase)
            for (int32_t k = 0; k < phase->n_pos; ++k) {
                const int64_t p = phase->pos[k];
                const int sl = (int)std::min<int64_t>(n_slabs - 1, std::max<int64_t>(0, p / ((int64_t)Gslab * 32)));
                if (sl == slab) {
                    ph_pos.push_back((int32_t)p);
                    ph_idx.push_back(k);
                }
            }
        if (phase && ph_pos.empty()) continue;
      for (size_t ph0 = 0; ph0 < (phase ? ph_pos.size() : 1); ph0 += kMaxPhasePositions) {
        const int ph_n = phase ? (int)std::min<size_t>(kMaxPhasePositions, ph_pos.size() - ph0) : 0;
        const size_t wave_lds = (size_t)kRows * RD * 16 + (size_t)((kRows * G * 2 + 15) / 16) * 16;  // rows + deviation queue
        // batch counter; then guide + alt haplotype words, or the phasing position pairs; then (pileup, when it costs
        // no wave) one alt counter per slab column
        const size_t guide_lds = 16 + std::max((size_t)(G + 3) * 32, (size_t)ph_n * 8);
        const int max_waves_probe = max_waves_of(np_kernel, passes);
        const size_t alt_lds = ((size_t)(G + 1) * 32 + 96) * 4;  // alt counters (first: the parked base counts), rowdev
        const bool alt_on = !MSQ_NO_ALT && !phase && std::min<size_t>(max_waves_probe, (160 * 1024 - guide_lds - alt_lds - 512) / wave_lds) ==
                                          std::min<size_t>(max_waves_probe, (160 * 1024 - guide_lds - 512) / wave_lds);
        const size_t shared_lds = guide_lds + (alt_on ? alt_lds : 0);
        a.alt_on = alt_on ? 1u : 0u;
        // waves per workgroup: as many as fit one CU (LDS, 2-3 waves per SIMD), one workgroup per CU
        const int max_waves = max_waves_of(np_kernel, passes);
        const int nwv = (int)std::min<size_t>(max_waves, (160 * 1024 - shared_lds - 512) / wave_lds);
        if (nwv < 1) return msq_fail(MSQ_ERR_ARG, "msq_pileup: no LDS for a %d-word slab", G);
        // the pileup's counters hold 1023 reads per wave and launch; the gather has no such limit
        const uint64_t reads_per_launch = phase ? R.n_reads : (uint64_t)ctx->n_cu * nwv * kMaxBatchesPerWave * kRows;
        pileup_kernel_fn kern = phase ? pick_phase_kernel(np_kernel, idx) : pick_kernel(np_kernel, passes, many_n, idx);
        if (fused && R.n_reads > reads_per_launch) return MSQ_OK;  // more than one launch: not for the fused step
        if (phase) {
            void* s = nullptr;
            int rc = msq_scratch(ctx, (size_t)ph_n * 8, &s);
            if (rc != MSQ_OK) return rc;
            MSQ_HIP(hipMemcpyAsync(s, ph_pos.data() + ph0, (size_t)ph_n * 4, hipMemcpyHostToDevice, ctx->stream));
            MSQ_HIP(hipMemcpyAsync((char*)s + (size_t)ph_n * 4, ph_idx.data() + ph0, (size_t)ph_n * 4, hipMemcpyHostToDevice, ctx->stream));
            a.ph_pos = (const int32_t*)s;
            a.ph_idx = (const int32_t*)s + ph_n;
            a.ph_n = ph_n;
            a.ph_total = phase->n_pos;
            a.ph_keys = phase->d_keys;
        }
        const bool last_slab = slab == (only_slab >= 0 ? only_slab : n_slabs - 1);
        for (uint64_t lo = 0; lo < R.n_reads; lo += reads_per_launch) {
            const uint64_t hi = std::min<uint64_t>(R.n_reads, lo + reads_per_launch);
            const uint64_t n_batches = (hi - lo + kRows - 1) / kRows;
            int w = nwv, n_wg = ctx->n_cu;
            if ((uint64_t)n_wg * w > n_batches) {  // small inputs: fewer, thinner workgroups
                w = (int)std::max<uint64_t>(1, std::min<uint64_t>(w, (n_batches + ctx->n_cu - 1) / ctx->n_cu));
                n_wg = (int)((n_batches + w - 1) / w);
            }
            // the flush of the kernels without QV planes adds the waves' counters up through LDS: [NC][w][kLevels][64] words
            const size_t slab_lds = (phase || np_kernel) ? (size_t)0 : (size_t)NC * w * kLevels * 256;  // one pass's levels
            const bool flush_wide = slab_lds * passes <= 160 * 1024 - 512;
            a.flush_wide = flush_wide ? 1u : 0u;
            const size_t lds = std::max((size_t)w * wave_lds + shared_lds, flush_wide ? slab_lds * passes : slab_lds);
            a.partials = nullptr;
            if (!phase) {
                // scratch: partials [n_wg][NC + 4][G*32] (counts per counter, the workgroup's guide bins per column, the alt
                // haplotype's nucleotide and codon counts and its bins)
                const size_t need = (size_t)ctx->n_cu * (NC + 4) * Gslab * 32 * 4;
                void* s = fixed_partials;
                if (fixed_partials) {
                    if (need > fixed_bytes)
                        return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: %zu bytes of workgroup partials exceed the built-in %zu", need, fixed_bytes);
                } else {
                    int rc = msq_scratch(ctx, need, &s);
                    if (rc != MSQ_OK) return rc;
                }
                a.partials = (uint32_t*)s;
            }
            a.read_lo = lo;
            a.read_hi = hi;
            a.n_batches = n_batches;
            // the carrier rule costs two workgroup barriers and a second pass over the counts (~4 us per launch): for
            // launches of at least kCarrierMinBatches batches per wave
            // (MSQ_CARRIER_MIN_BATCHES in the environment when the context is made, or msq_debug_tune, overrides the
            // threshold: the tests run small sets through both rules)
            const uint64_t min_batches = ctx->carrier_min_batches >= 0 ? (uint64_t)ctx->carrier_min_batches : (uint64_t)kCarrierMinBatches;
            a.carrier = (!MSQ_NO_CARRIER && n_batches >= min_batches * w * n_wg) ? 1u : 0u;
            if (fused) {  // deviations into the context's matrix; the fold stores the sum and clears that matrix again
                a.nuc = fused->d_dev;
                a.codon = fused->d_dev + (size_t)W * 6;
                a.ref_end = R.ref_end;
                a.bounds_acc = ctx->d_small + 16;
                a.bounds_out = fused->d_bounds;
            }
            if (!phase && !marked) {
                msq_timing_mark_begin(ctx);
                marked = true;
            }
            if (int rc = msq_ensure_dynamic_lds(ctx->device, reinterpret_cast<const void*>(kern), lds)) return rc;
            hipLaunchKernelGGL(kern, dim3(n_wg), dim3(w * 64), lds, ctx->stream, a);
            if (!phase && last_slab && hi == R.n_reads) msq_timing_mark_end(ctx);
            MSQ_HIP(hipGetLastError());
            if (phase) continue;
            const int col_blocks = (G * 32 + kFoldCols - 1) / kFoldCols;
            const int slices = fused ? 1 : std::max(1, std::min(n_wg / (kFoldSubs * kFoldTrip), (ctx->n_cu / 2 + col_blocks - 1) / col_blocks));
            hipLaunchKernelGGL(fold_partials_kernel, dim3(col_blocks, slices), dim3(kFoldCols * kFoldSubs), 0, ctx->stream, a.partials, n_wg,
                               NC, G * 32, a.slab0, W, d_nuc, d_codon, fused ? a.nuc : (uint32_t*)nullptr, fused ? a.codon : (uint32_t*)nullptr);
            MSQ_HIP(hipGetLastError());
        }
        if (phase) MSQ_HIP(hipStreamSynchronize(ctx->stream));  // the host position vectors and the scratch are reused
      }
    }
    *handled = true;
    return MSQ_OK;
}

int msq_pileup_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                             int32_t W, const int32_t* d_bounds, uint32_t* d_nuc, uint32_t* d_codon, bool* handled)
{
    return stream_launch(ctx, R, thr, win_begin0, W, d_bounds, d_nuc, d_codon, nullptr, handled);
}

// one column slab only (msq_pileup_slab): the caller's read set need hold no bases outside that slab and its margins
int msq_pileup_stream_slab_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                                  int32_t W, int slab, int slab_words, uint32_t* d_nuc, uint32_t* d_codon, bool* handled)
{
    return stream_launch(ctx, R, thr, win_begin0, W, nullptr, d_nuc, d_codon, nullptr, handled, slab, slab_words);
}

// every slab of a partition in one launch (msq_pileup_partition); *handled = false: one msq_pileup_stream_slab_launch per slab
int msq_pileup_stream_slabs_launch(msq_ctx* ctx, const msq_packed_reads* const* sets, int n_sets, const msq_qv_thresholds& thr,
                                   int32_t win_begin0, int32_t W, int slab_words, uint32_t* d_nuc, uint32_t* d_codon, bool* handled)
{
    return stream_launch_merged(ctx, sets, n_sets, thr, win_begin0, W, slab_words, nullptr, d_nuc, d_codon, nullptr, 0, handled);
}

int msq_pileup_rows_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                           uint32_t* d_nuc, uint32_t* d_codon, uint32_t* fixed_partials, size_t fixed_bytes, int32_t* d_step_bounds,
                           uint32_t* d_dev, bool* handled);

// msq_pileup_step's streaming half; *handled = false: take msq_pileup_begin + msq_pileup instead
int msq_pileup_stream_step_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0, int32_t W,
                                  int32_t* d_bounds, uint32_t* d_counts, bool* handled)
{
    *handled = false;
    if (R.flags & MSQ_READS_HAS_EXOTIC) return MSQ_OK;
    const size_t words = (size_t)W * 70;
    if (ctx->dev_words < words) {  // the deviation matrix: grown on demand, zero from here on (the fold hands it back cleared)
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_dev) MSQ_HIP(hipFree(ctx->d_dev));
        ctx->d_dev = nullptr;
        ctx->dev_words = 0;
        hipError_t e = hipMalloc(&ctx->d_dev, words * sizeof(uint32_t));
        if (e != hipSuccess) return msq_fail(MSQ_ERR_NOMEM, "msq_pileup_step: hipMalloc(%zu): %s", words * sizeof(uint32_t), hipGetErrorString(e));
        ctx->dev_words = words;
        ctx->dev_clean = false;
    }
    if (!ctx->dev_clean) MSQ_HIP(hipMemsetAsync(ctx->d_dev, 0, ctx->dev_words * sizeof(uint32_t), ctx->stream));
    ctx->dev_clean = false;  // until the fold that clears it is in the stream
    FusedStep f{d_bounds, ctx->d_dev};
    // a column-aligned set has a kernel of its own (msq_pileup_rows.hip); everything else, and an aligned set it does
    // not take, streams through the read-space kernel
    int rc = msq_pileup_rows_launch(ctx, R, thr, win_begin0, W, d_counts, d_counts + (size_t)W * 6, nullptr, 0, d_bounds, ctx->d_dev, handled);
    if (rc == MSQ_OK && !*handled)
        rc = stream_launch(ctx, R, thr, win_begin0, W, nullptr, d_counts, d_counts + (size_t)W * 6, nullptr, handled, -1, 128, nullptr, 0, &f);
    if (rc != MSQ_OK) return rc;
    ctx->dev_clean = true;  // untouched (not handled), or cleared by the fold
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------ no-allocation door
// msq_pileup_noalloc: msq_pileup's streaming path with the workgroup partials -- the one buffer the pileup needs
// beyond its inputs and outputs -- in a file-scope device array instead of the context's grown-on-demand scratch.
// One copy of the array exists per GPU; its address on a GPU is asked for once, with that GPU current
// (msq_noalloc_init, from msq_ctx_create), and kept in the context.  Only that pointer ever reaches a kernel or a
// runtime call: `scratch` itself is a host-side handle here, not a device address.
constexpr int kNoallocMaxColumns = 4096;  // one column slab
constexpr int kNoallocMaxCU = 256;
constexpr size_t kNoallocWords = (size_t)kNoallocMaxCU * 8 * kNoallocMaxColumns;  // [n_cu][NC + 4 <= 8][columns] u32 = 32 MiB
static __device__ __align__(256) uint32_t scratch[kNoallocWords];

int msq_noalloc_init(msq_ctx* ctx)  // the context's device is current
{
    void* p = nullptr;
    MSQ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    ctx->d_noalloc = static_cast<uint32_t*>(p);
    ctx->noalloc_bytes = sizeof(uint32_t) * kNoallocWords;
    return MSQ_OK;
}

extern "C" int msq_pileup_noalloc(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                                  int32_t W, uint32_t* d_nuc, uint32_t* d_codon)
{
    if (!ctx || !d_reads || !thr) return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: NULL argument");
    if (W < 0) return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: negative window width %d", W);
    if (W > kNoallocMaxColumns)
        return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: window of %d columns exceeds the maximum of %d (use msq_pileup)", W, kNoallocMaxColumns);
    if (d_reads->n_reads == 0 || W == 0) return MSQ_OK;
    int rc = msq_validate_reads(d_reads, thr, "msq_pileup_noalloc");
    if (rc != MSQ_OK) return rc;
    if (!d_nuc || !d_codon) return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: NULL count matrix");
    if (d_reads->flags & MSQ_READS_HAS_EXOTIC)
        return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: reads with exotic symbols need msq_pileup (flags=%u)", d_reads->flags);
    if (!ctx->d_noalloc) return msq_fail(MSQ_ERR_HIP, "msq_pileup_noalloc: the built-in scratch was not resolved at context creation");
    MSQ_HIP(hipSetDevice(ctx->device));
    bool handled = false;
    rc = msq_pileup_rows_launch(ctx, *d_reads, *thr, win_begin0, W, d_nuc, d_codon, ctx->d_noalloc, ctx->noalloc_bytes, nullptr, nullptr, &handled);
    if (rc == MSQ_OK && !handled)
        rc = stream_launch(ctx, *d_reads, *thr, win_begin0, W, nullptr, d_nuc, d_codon, nullptr, &handled, -1, 128, ctx->d_noalloc,
                           ctx->noalloc_bytes);
    if (rc != MSQ_OK) return rc;
    if (!handled) return msq_fail(MSQ_ERR_ARG, "msq_pileup_noalloc: streaming kernel not applicable (flags=%u)", d_reads->flags);
    return MSQ_OK;
}

int msq_phase_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                            int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys, bool* handled)
{
    const PhaseRequest req{n_pos, pos, d_keys};
    return stream_launch(ctx, R, thr, win_begin0, W, nullptr, nullptr, nullptr, &req, handled);
}
