// msq_stream_common.h -- device helpers shared by the pileup kernels (msq_pileup_stream.hip: the read-space streaming
// kernel; msq_pileup_rows.hip: the kernel over column-aligned sets): bit-sliced counters, the wave-level primitives, and
// the fold of the workgroup partials into the count matrices.  Not part of the ABI.
#pragma once

#include "msq_internal.h"

namespace {

#ifndef MSQ_LEVELS
#define MSQ_LEVELS 10
#endif
#ifndef MSQ_FA
#define MSQ_FA 1
#endif
#ifndef MSQ_NS0
#define MSQ_NS0 4
#endif
constexpr int kLevels = MSQ_LEVELS;       // bit-sliced counter depth: 1023 reads per wave and launch (register arrays are sized for it)
constexpr int kLowLevels = 6;     // levels rippled on every add; the upper ones only when a carry reaches them
constexpr int kRows = 8;          // reads per wave batch
constexpr int kMaxBatchesPerWave = ((1 << kLevels) - 1) / kRows;
constexpr int kMaxMerged = 8;     // column slabs one merged launch (and one fold launch) can take


__device__ __forceinline__ uint32_t funnel(uint32_t lo, uint32_t hi, int k)  // bits k.. of the 64-bit hi:lo
{
    return __builtin_amdgcn_alignbit(hi, lo, k);
}

// carry-save adder: (a + b + c) -> sum (weight 1), carry (weight 2); two v_bitop3 on gfx950
__device__ __forceinline__ void csa(uint32_t a, uint32_t b, uint32_t c, uint32_t& sum, uint32_t& carry)
{
    const uint32_t s = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // a ^ b ^ c
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8);             // majority(a, b, c)
    sum = s;
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

// Half-adder chain over levels [FROM, TO): returns the carry out of level TO - 1.
template <int FROM, int TO>
__device__ __forceinline__ uint32_t ripple(uint32_t (&cnt)[kLevels], uint32_t carry)
{
#pragma unroll
    for (int l = FROM; l < TO; ++l) {
        const uint32_t c = cnt[l] & carry;
        cnt[l] ^= carry;
        carry = c;
    }
    return carry;
}

// a[l] bit c -> a[c & 15] bit l (+ 16 for c >= 16): the 16 x 16 bit matrices in the low and the high halves of the 16
// words are transposed, both at once (block swaps of 8, 4, 2, 1)
__device__ __forceinline__ void transpose16x2(uint32_t (&a)[16])
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int j = 8 >> s;
        const uint32_t m = s == 0 ? 0x00FF00FFu : s == 1 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (!(k & j)) {
                const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
                a[k] ^= t << j;
                a[k + j] ^= t;
            }
    }
}


// Order this wave's LDS traffic between phases (lanes exchange data through LDS; no instructions on gfx9).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

// Columns of word cg that can start a codon: 1 <= c <= W-3 (MSARow::CodonAt's i > 0, CodingCodonAt's bound;
// MSA.cpp:193,207).
__device__ __forceinline__ uint32_t codon_cols_of(int cg, int W)  // cg: word of the window
{
    const int c0 = cg * 32;
    const int lo = max(1 - c0, 0), hi = min(W - 3 - c0, 31);
    return lo <= hi ? (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo) : 0u;
}


// Fold the workgroup partials into the count matrices: matches go to each workgroup's guide nucleotide / codon
// bin, gaps to '-', Ns to 'N' (bins 4 and 5, NucleotideConversion.h:69-88).  Block = 32 columns x 32 workgroup
// lanes: a thread has the loads of eight workgroups' partials in flight at once (the kernel is latency-bound: one
// trip for 256 workgroups), the 32 lanes of a column meet in LDS, and what is not zero there leaves as one global
// atomic per bin.  blockIdx.y slices the workgroups further (more blocks, but as many more global atomics).
constexpr int kFoldCols = 32, kFoldSubs = 32, kFoldTrip = 8;
__device__ __forceinline__ void fold_partials(const uint32_t* __restrict__ partials, int n_wg, int NC, int cols, int col0, int W,
                                              uint32_t* __restrict__ nuc, uint32_t* __restrict__ codon, uint32_t* __restrict__ dev_nuc,
                                              uint32_t* __restrict__ dev_codon)
{
    constexpr int kCodStride = 65, kNucStride = 7;  // odd strides: the lanes of a wave are columns
    __shared__ uint32_t cod_acc[kFoldCols * kCodStride];
    __shared__ uint32_t nuc_acc[kFoldCols * kNucStride];
    const int T = kFoldCols * kFoldSubs, tid = threadIdx.x;
    for (int i = tid; i < kFoldCols * kCodStride; i += T) cod_acc[i] = 0;
    for (int i = tid; i < kFoldCols * kNucStride; i += T) nuc_acc[i] = 0;
    const int cl = tid & (kFoldCols - 1), sub = tid / kFoldCols;
    const int c = blockIdx.x * kFoldCols + cl;  // column of the slab; col0 + c = column of the window
    const int per = (n_wg + gridDim.y - 1) / gridDim.y;
    const int wg_lo = blockIdx.y * per, wg_hi = min(n_wg, wg_lo + per);
    const bool valid = c < cols && col0 + c < W;
    __syncthreads();
    if (valid) {
        uint32_t base[4] = {0, 0, 0, 0}, gap = 0, nn = 0;
        // codon counts: runs of workgroups that agree on the bin are added up before they cost an atomic (set 0 = the
        // guide's codon, set 1 = the alt haplotype's)
        uint32_t cod_sum[2] = {0, 0};
        int cod_idx[2] = {-1, -1};
        auto add_codon = [&](int set, uint32_t meta, uint32_t n) {
            if (!n) return;
            const int idx = (int)((meta >> 8) & 63u);
            if (idx != cod_idx[set]) {
                if (cod_sum[set]) atomicAdd(&cod_acc[cl * kCodStride + cod_idx[set]], cod_sum[set]);
                cod_idx[set] = idx;
                cod_sum[set] = 0;
            }
            cod_sum[set] += n;
        };
        for (int wg0 = wg_lo + sub; wg0 < wg_hi; wg0 += kFoldSubs * kFoldTrip) {
            uint32_t meta[kFoldTrip], m[kFoldTrip], g[kFoldTrip], cm[kFoldTrip], n4[kFoldTrip], am[kFoldTrip], acm[kFoldTrip], meta2[kFoldTrip];
#pragma unroll
            for (int u = 0; u < kFoldTrip; ++u) {
                const int wg = min(wg0 + kFoldSubs * u, wg_hi - 1);  // clamped: a slot past the end is loaded and ignored
                const uint32_t* p = partials + (size_t)wg * (NC + 4) * cols;
                meta[u] = p[(size_t)NC * cols + c];
                m[u] = p[c];
                g[u] = p[(size_t)cols + c];
                cm[u] = p[(size_t)2 * cols + c];
                n4[u] = NC > 3 ? p[(size_t)3 * cols + c] : 0u;
                am[u] = p[(size_t)(NC + 1) * cols + c];
                acm[u] = p[(size_t)(NC + 2) * cols + c];
                meta2[u] = p[(size_t)(NC + 3) * cols + c];
            }
#pragma unroll
            for (int u = 0; u < kFoldTrip; ++u) {
                if (wg0 + kFoldSubs * u >= wg_hi) break;
#pragma unroll
                for (int k = 0; k < 4; ++k)  // a count > 0 implies a valid base in that plane set
                    base[k] += ((meta[u] & 3u) == (uint32_t)k ? m[u] : 0u) + ((meta2[u] & 3u) == (uint32_t)k ? am[u] : 0u);
                gap += g[u];
                nn += n4[u];
                add_codon(0, meta[u], cm[u]);
                add_codon(1, meta2[u], acm[u]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base[k]) atomicAdd(&nuc_acc[cl * kNucStride + k], base[k]);
        if (gap) atomicAdd(&nuc_acc[cl * kNucStride + 4], gap);
        if (nn) atomicAdd(&nuc_acc[cl * kNucStride + 5], nn);
#pragma unroll
        for (int set = 0; set < 2; ++set)
            if (cod_sum[set]) atomicAdd(&cod_acc[cl * kCodStride + cod_idx[set]], cod_sum[set]);
    }
    __syncthreads();
    const int cb = col0 + blockIdx.x * kFoldCols;  // window column of the block's column 0
    nuc += (size_t)cb * 6;
    codon += (size_t)cb * 64;
    if (dev_nuc) {
        // msq_pileup_step: this block is the only writer of its columns (gridDim.y == 1, one launch): the matrices are
        // STORED -- the fold's sums plus what the deviation path added to its own matrix, which is handed back cleared
        dev_nuc += (size_t)cb * 6;
        dev_codon += (size_t)cb * 64;
        const int ncol = min(kFoldCols, min(cols - blockIdx.x * kFoldCols, W - cb));
        for (int i = tid; i < ncol * 64; i += T) {
            codon[i] = cod_acc[(i >> 6) * kCodStride + (i & 63)] + dev_codon[i];
            dev_codon[i] = 0;
        }
        for (int i = tid; i < ncol * 6; i += T) {
            nuc[i] = nuc_acc[(i / 6) * kNucStride + i % 6] + dev_nuc[i];
            dev_nuc[i] = 0;
        }
        return;
    }
    for (int i = tid; i < kFoldCols * 64; i += T) {  // only columns that were valid hold anything
        const uint32_t v = cod_acc[(i >> 6) * kCodStride + (i & 63)];
        if (v) atomicAdd(&codon[i], v);
    }
    for (int i = tid; i < kFoldCols * 6; i += T) {
        const uint32_t v = nuc_acc[(i / 6) * kNucStride + i % 6];
        if (v) atomicAdd(&nuc[i], v);
    }
}

__attribute__((unused)) __global__ __launch_bounds__(kFoldCols* kFoldSubs) void fold_partials_kernel(const uint32_t* __restrict__ partials, int n_wg, int NC, int cols,
                                                                             int col0, int W, uint32_t* __restrict__ nuc,
                                                                             uint32_t* __restrict__ codon, uint32_t* __restrict__ dev_nuc,
                                                                             uint32_t* __restrict__ dev_codon)
{
    fold_partials(partials, n_wg, NC, cols, col0, W, nuc, codon, dev_nuc, dev_codon);
}

// the folds of a merged launch as one launch: blockIdx.z = the slab
struct FoldSlabs {
    const uint32_t* partials[kMaxMerged];
    int n_wg[kMaxMerged], cols[kMaxMerged], col0[kMaxMerged];
};
__attribute__((unused)) __global__ __launch_bounds__(kFoldCols* kFoldSubs) void fold_slabs_kernel(FoldSlabs t, int NC, int W, uint32_t* __restrict__ nuc,
                                                                          uint32_t* __restrict__ codon)
{
    const int sl = blockIdx.z;
    if ((int)blockIdx.x * kFoldCols >= t.cols[sl]) return;  // the grid is as wide as the widest slab
    fold_partials(t.partials[sl], t.n_wg[sl], NC, t.cols[sl], t.col0[sl], W, nuc, codon, nullptr, nullptr);
}


}  // namespace
