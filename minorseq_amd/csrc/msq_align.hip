// msq_align.hip -- msq_reads_align: a packed read set re-laid in COLUMN space (MSQ_READS_ALIGNED) + its guide.
//
// MSAByRow::AddRead (src/MSA.cpp:144-186) places base after base with `row.Bases[pos++]`: where a base lands depends on
// every insertion before it.  The other group layouts keep 32 consecutive bases of a read per group, so each pass
// over the set has to find the columns again and shift every group's planes into place.  Here that placement is done
// once: group k of a read holds the reference columns [origin + 32 (a + k), + 32), a = (ReferenceStart - origin) >> 5,
// the base of column c in slot c & 31, padding (op 3) where the read has no base, inserted bases left out.  The
// encoding of a slot is unchanged (2 bits of nucleotide, 2 bits of op), so the result is still a valid plain-layout
// set for every kernel; the pileup has a kernel of its own for it (msq_pileup_rows.hip) that builds no rows at all.
//
// Three kernels, all at set-up time: count the column-consuming bases per read (-> groups per read, prefix sum),
// build every read's row in LDS by OR-ing its groups' planes at their bit offsets and write it out, and learn the
// set's guide (per column the majority base and a runner-up) from a sample of the aligned reads.
#include "msq_internal.h"
#include "msq_stream_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

int msq_packed_dev_index(msq_ctx* ctx, msq_packed_dev* d);

namespace {

// Remove the bits at the zero positions of `keep` below keep's top set bit (the plain layout: insertions and pads sit
// between the column-consuming bases), for three planes at once.
__device__ __forceinline__ void squeeze3(uint32_t keep, uint32_t& p0, uint32_t& p1, uint32_t& p2)
{
    uint32_t holes = keep ? ~keep & (0xffffffffu >> __clz(keep)) : 0u;
    while (holes) {
        const int j = 31 - __clz(holes);
        const uint32_t low = j ? (0xffffffffu >> (32 - j)) : 0u;
        p0 = (p0 & low) | ((p0 >> 1) & ~low);
        p1 = (p1 & low) | ((p1 >> 1) & ~low);
        p2 = (p2 & low) | ((p2 >> 1) & ~low);
        holes &= low;
    }
}

// ng_new[r] = 32-column words the row of read r spans on the grid of `origin` (0 for a read without a column-consuming
// base); acc[0] = max over reads of (first word + words) = words the set spans from the origin, acc[1] = max words of
// one read, acc[2] |= 1 when a read starts left of the origin.
__global__ __launch_bounds__(256) void align_count_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                          const uint4* __restrict__ bop, uint64_t n_reads, int32_t origin,
                                                          uint64_t* __restrict__ ng_new, uint32_t* __restrict__ acc)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // the maxima of the wave's reads, and one set of atomics per wave at the end: an atomic per read on the same three
    // words (10^6 reads: 2 * 10^6 atomicMax on one address, which the L2 takes one by one) was 22 of this kernel's 23 ms
    uint32_t span_max = 0, words_max = 0, left = 0;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t g0 = group_off[r], g1 = group_off[r + 1];
        long long n = 0;
        for (uint64_t g = g0 + lane; g < g1; g += 64) n += __popc(~bop[g].w);
        for (int d = 32; d >= 1; d >>= 1) n += __shfl_xor(n, d);
        if (lane == 0) {
            const long long rel = (long long)ref_start[r] - origin;
            uint64_t words = 0;
            if (rel < 0) left = 1u;
            if (n > 0 && rel >= 0) {
                const long long a = rel >> 5;
                words = (uint64_t)(((rel + n + 31) >> 5) - a);
                span_max = max(span_max, (uint32_t)min((long long)0xffffffffll, a + (long long)words));
                words_max = max(words_max, (uint32_t)min((uint64_t)0xffffffffu, words));
            }
            ng_new[r] = words;
        }
    }
    if (lane == 0) {
        if (left) atomicOr(&acc[2], 1u);
        if (span_max) atomicMax(&acc[0], span_max);
        if (words_max) atomicMax(&acc[1], words_max);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ng_new[n_reads] = 0;  // the scan's last input
}

// One wave per read: its row as four OR-planes per word in LDS -- C (a base lands here), O0 (op bit 0), N0, N1 -- built
// from the source groups exactly as AddRead walks them, then written out in the packed encoding (padding = op 3).
__global__ __launch_bounds__(256) void align_rows_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                         const uint4* __restrict__ bop, uint64_t n_reads, int32_t origin, uint32_t normalized,
                                                         const uint64_t* __restrict__ new_off, uint4* __restrict__ out, int row_words)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6, waves_per_block = blockDim.x >> 6;
    uint32_t* const row = lds + (size_t)wave_in_block * row_words * 4;
    const uint64_t wave = (uint64_t)blockIdx.x * waves_per_block + wave_in_block, n_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t o0 = new_off[r], o1 = new_off[r + 1];
        const int nw = (int)(o1 - o0);
        if (nw == 0) continue;  // wave-uniform
        for (int i = lane; i < (nw + 1) * 4; i += 64) row[i] = 0;
        wave_sync();
        const uint64_t g0 = group_off[r], g1 = group_off[r + 1];
        int col = (int)(((long long)ref_start[r] - origin) & 31);  // bit offset of the read's first base in its first word
        for (uint64_t gb = g0; gb < g1; gb += 64) {
            const uint64_t g = gb + lane;
            uint4 w = make_uint4(0u, 0u, ~0u, ~0u);
            if (g < g1) w = bop[g];
            const uint32_t cons = ~w.w;
            const int n = __popc(cons);
            int incl = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(incl, d);
                if (lane >= d) incl += t;
            }
            const int c0 = col + incl - n;
            if (n) {
                uint32_t p0 = w.z & cons, p1 = w.x & cons, p2 = w.y & cons;
                if (!normalized) squeeze3(cons, p0, p1, p2);
                const uint32_t c = n == 32 ? 0xffffffffu : ((1u << n) - 1u);
                const int wd = c0 >> 5, sh = c0 & 31;
                uint32_t* p = row + wd * 4;
                atomicOr(p + 0, c << sh);
                atomicOr(p + 1, p0 << sh);
                atomicOr(p + 2, p1 << sh);
                atomicOr(p + 3, p2 << sh);
                if (sh && sh + n > 32) {
                    atomicOr(p + 4, c >> (32 - sh));
                    atomicOr(p + 5, p0 >> (32 - sh));
                    atomicOr(p + 6, p1 >> (32 - sh));
                    atomicOr(p + 7, p2 >> (32 - sh));
                }
            }
            col += __shfl(incl, 63);
        }
        wave_sync();
        for (int k = lane; k < nw; k += 64) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + k * 4);  // C, O0, N0, N1
            out[o0 + k] = make_uint4(q.z, q.w, q.y | ~q.x, ~q.x);
        }
        wave_sync();
    }
}

// The guide of an aligned set: block = one 32-column word of the grid.  Every sampled read that covers the word adds
// its match bases ('=' / 'X' that are ACGT: op 0) to per-column counts in LDS; per column the most common base (the
// smaller code among equals) and, where at least `alt_min` reads agree on another base, the most common other one.
// With a read list (`list` != NULL: the reads of one haplotype group, `n_reads` of them) the sample is taken from the list.
__global__ __launch_bounds__(256) void align_guide_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ off,
                                                          const uint4* __restrict__ bop, uint64_t n_reads, int32_t origin, uint32_t sample,
                                                          uint32_t alt_min, uint32_t* __restrict__ guide, const uint32_t* __restrict__ list)
{
    __shared__ uint32_t cnt[32][4];
    const int w = blockIdx.x, tid = threadIdx.x;
    if (tid < 128) cnt[tid >> 2][tid & 3] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < sample; i += blockDim.x) {
        const uint64_t at = (uint64_t)i * n_reads / sample;
        const uint64_t r = list ? (uint64_t)list[at] : at;
        const uint64_t o0 = off[r], o1 = off[r + 1];
        const long long k = (long long)w - (((long long)ref_start[r] - origin) >> 5);
        if (k < 0 || (uint64_t)k >= o1 - o0) continue;
        const uint4 q = bop[o0 + (uint64_t)k];
        uint32_t m = ~(q.z | q.w);  // op 0
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1;
            atomicAdd(&cnt[j][((q.x >> j) & 1u) | (((q.y >> j) & 1u) << 1)], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int j = tid & 31;
        const uint32_t c[4] = {cnt[j][0], cnt[j][1], cnt[j][2], cnt[j][3]};
        int best = 0;
        for (int b = 1; b < 4; ++b)
            if (c[b] > c[best]) best = b;
        const bool any = c[best] > 0;
        int second = -1;
        for (int b = 0; b < 4; ++b)
            if (b != best && c[b] >= alt_min && (second < 0 || c[b] > c[second])) second = b;
        const bool has_alt = any && second >= 0;
        const int alt = has_alt ? second : best;
        const uint32_t lo = 0xffffffffu;
        const uint32_t v_any = (uint32_t)(__ballot(any) & lo), g0 = (uint32_t)(__ballot(any && (best & 1)) & lo),
                       g1 = (uint32_t)(__ballot(any && (best & 2)) & lo), a0 = (uint32_t)(__ballot(any && (alt & 1)) & lo),
                       a1 = (uint32_t)(__ballot(any && (alt & 2)) & lo), ha = (uint32_t)(__ballot(has_alt) & lo);
        if (tid == 0) {
            uint32_t* g = guide + (size_t)w * 8;
            g[0] = v_any; g[1] = g0; g[2] = g1; g[3] = 0;
            g[4] = v_any; g[5] = a0; g[6] = a1; g[7] = ha;
        }
    }
}

// ---- haplotype groups.  A sample of several clones (a mixture of strains, minor populations with many variant codons)
// strays from ONE guide in every read of a clone at every variant column; with a guide per clone every read matches its
// own but for its errors.  At set-up: the columns where the sample supports a second base (the guide's has_alt mask) are
// the variant columns; the sampled reads' bases there are clustered on the host (leader clustering, a few per cent of
// tolerance for errors) into <= MSQ_MAX_HAPS haplotypes; every read of the set is assigned to the nearest one in one
// streaming pass; the reads are listed group by group; every group learns its own guide + alt from its own reads.
// Any grouping gives the same counts -- it only decides which cells take the cheap path.

// codes[i][j] = base code (0 .. 3) of sampled read i at variant column j, 4 where it has no '=' / 'X' ACGT base
__global__ __launch_bounds__(256) void align_sample_codes_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ off,
                                                                 const uint4* __restrict__ bop, uint64_t n_reads, int32_t origin, uint32_t sample,
                                                                 const uint32_t* __restrict__ var_cols, uint32_t nvar, uint8_t* __restrict__ codes)
{
    const uint32_t i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nvar) return;
    const uint64_t r = (uint64_t)i * n_reads / sample;
    const uint64_t o0 = off[r], o1 = off[r + 1];
    const uint32_t c = var_cols[j];
    const long long k = (long long)(c >> 5) - (((long long)ref_start[r] - origin) >> 5);
    uint8_t code = 4;
    if (k >= 0 && (uint64_t)k < o1 - o0) {
        const uint4 q = bop[o0 + (uint64_t)k];
        const uint32_t b = c & 31u;
        if (((~(q.z | q.w)) >> b) & 1u) code = (uint8_t)(((q.x >> b) & 1u) | (((q.y >> b) & 1u) << 1));
    }
    codes[(size_t)i * nvar + j] = code;
}

// One wave per read: its mismatches against every haplotype at the variant columns; the nearest haplotype (the lower
// index among equals) is its group.  idx[r] = r (the sort's values); counts[h] += reads of group h.
__global__ __launch_bounds__(256) void align_assign_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ off,
                                                           const uint4* __restrict__ bop, uint64_t n_reads, int32_t origin,
                                                           const uint32_t* __restrict__ planes, const uint32_t* __restrict__ varmask, uint32_t words,
                                                           int n_haps, uint8_t* __restrict__ hap_id, uint32_t* __restrict__ idx,
                                                           uint32_t* __restrict__ counts)
{
    __shared__ uint32_t hist[MSQ_MAX_HAPS];
    if (threadIdx.x < MSQ_MAX_HAPS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t o0 = off[r], o1 = off[r + 1];
        const long long a = ((long long)ref_start[r] - origin) >> 5;
        uint32_t mism[MSQ_MAX_HAPS];
#pragma unroll
        for (int h = 0; h < MSQ_MAX_HAPS; ++h) mism[h] = 0;
        for (uint64_t k = lane; k < o1 - o0; k += 64) {
            const uint64_t w = (uint64_t)a + k;
            if (w >= words) continue;
            const uint32_t vm = varmask[w];
            if (!vm) continue;
            const uint4 q = bop[o0 + k];
            const uint32_t v = ~(q.z | q.w) & vm;
#pragma unroll
            for (int h = 0; h < MSQ_MAX_HAPS; ++h)
                if (h < n_haps) mism[h] += __popc(v & ((q.x ^ planes[((size_t)h * words + w) * 2]) | (q.y ^ planes[((size_t)h * words + w) * 2 + 1])));
        }
        uint32_t best = 0, best_d = 0xffffffffu;
#pragma unroll
        for (int h = 0; h < MSQ_MAX_HAPS; ++h) {
            if (h >= n_haps) break;
            uint32_t d = mism[h];
            for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s);
            if (d < best_d) {
                best_d = d;
                best = (uint32_t)h;
            }
        }
        if (lane == 0) {
            hap_id[r] = (uint8_t)best;
            idx[r] = (uint32_t)r;
            atomicAdd(&hist[best], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < MSQ_MAX_HAPS && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]);
}

// Leader clustering of the sampled reads' codes at the variant columns.  haps[h][j] = the group's base at variant column
// j (0 .. 3), 255 where its members do not say; groups ordered by size, the largest first.  Returns the number of groups
// (< 2: the sample is one population).
int cluster_sample(const std::vector<uint8_t>& codes, uint32_t sample, uint32_t nvar, std::vector<std::vector<uint8_t>>& haps)
{
    haps.clear();
    if (!nvar || sample < 8) return 1;
    const uint32_t tol = std::max<uint32_t>(2, nvar / 40), min_members = std::max<uint32_t>(4, sample / 256);
    const uint32_t min_cover = nvar - nvar / 10;  // reads that cover the variant columns (ragged reads do not vote)
    struct Leader {
        uint32_t rep;
        std::vector<uint32_t> members;
    };
    std::vector<Leader> leaders;
    for (uint32_t i = 0; i < sample; ++i) {
        const uint8_t* row = &codes[(size_t)i * nvar];
        uint32_t cover = 0;
        for (uint32_t j = 0; j < nvar; ++j) cover += row[j] < 4;
        if (cover < min_cover) continue;
        bool joined = false;
        for (Leader& l : leaders) {
            const uint8_t* rep = &codes[(size_t)l.rep * nvar];
            uint32_t d = 0;
            for (uint32_t j = 0; j < nvar && d <= tol; ++j) d += (row[j] < 4 && rep[j] < 4 && row[j] != rep[j]);
            if (d <= tol) {
                l.members.push_back(i);
                joined = true;
                break;
            }
        }
        if (!joined && leaders.size() < 64) leaders.push_back(Leader{i, {i}});
    }
    std::vector<const Leader*> kept;
    for (const Leader& l : leaders)
        if (l.members.size() >= min_members) kept.push_back(&l);
    std::stable_sort(kept.begin(), kept.end(), [](const Leader* x, const Leader* y) { return x->members.size() > y->members.size(); });
    if (kept.size() > MSQ_MAX_HAPS) kept.resize(MSQ_MAX_HAPS);
    if (kept.size() < 2) return 1;
    for (const Leader* l : kept) {
        std::vector<uint8_t> h(nvar, 255);
        for (uint32_t j = 0; j < nvar; ++j) {
            uint32_t c[4] = {0, 0, 0, 0};
            for (uint32_t m : l->members) {
                const uint8_t v = codes[(size_t)m * nvar + j];
                if (v < 4) ++c[v];
            }
            int best = 0;
            for (int b = 1; b < 4; ++b)
                if (c[b] > c[best]) best = b;
            if (c[best]) h[j] = (uint8_t)best;
        }
        haps.push_back(std::move(h));
    }
    return (int)haps.size();
}

void free_dev(msq_packed_dev* d)
{
    if (!d) return;
    for (int j = 0; j < 12; ++j)
        if (d->ptrs[j]) (void)hipFree(d->ptrs[j]);
    delete d;
}

// The haplotype groups of an aligned set whose guide has been learnt (see above).  Leaves the view without groups when the
// sample is one population.
int align_haplotypes(msq_ctx* ctx, msq_packed_dev* d, int32_t origin, uint32_t sample)
{
    msq_packed_reads& v = d->view;
    const uint64_t n = v.n_reads;
    const uint32_t words = v.guide_words;
    std::vector<uint32_t> guide((size_t)words * 8);
    MSQ_HIP(hipMemcpyAsync(guide.data(), v.guide, guide.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> var_cols;
    for (uint32_t w = 0; w < words; ++w)
        for (uint32_t m = guide[(size_t)w * 8 + 7]; m; m &= m - 1) var_cols.push_back(w * 32 + (uint32_t)__builtin_ctz(m));
    constexpr size_t kMaxVar = 4096;
    if (var_cols.size() > kMaxVar) {  // an even selection: the clustering needs enough columns, not all
        std::vector<uint32_t> pick;
        for (size_t i = 0; i < kMaxVar; ++i) pick.push_back(var_cols[i * var_cols.size() / kMaxVar]);
        var_cols.swap(pick);
    }
    const uint32_t nvar = (uint32_t)var_cols.size();
    if (nvar < 3 || sample < 8) return MSQ_OK;
    struct Tmp {
        std::vector<void*> p;
        ~Tmp()
        {
            for (void* q : p)
                if (q) (void)hipFree(q);
        }
        hipError_t alloc(void** out, size_t bytes)
        {
            const hipError_t e = hipMalloc(out, bytes ? bytes : 16);
            if (e == hipSuccess) p.push_back(*out);
            return e;
        }
    } tmp;
    uint32_t* d_var = nullptr;
    uint8_t* d_codes = nullptr;
    MSQ_HIP(tmp.alloc((void**)&d_var, (size_t)nvar * 4));
    MSQ_HIP(tmp.alloc((void**)&d_codes, (size_t)sample * nvar));
    MSQ_HIP(hipMemcpyAsync(d_var, var_cols.data(), (size_t)nvar * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(align_sample_codes_kernel, dim3((nvar + 255) / 256, sample), dim3(256), 0, ctx->stream, v.ref_start, v.group_off,
                       reinterpret_cast<const uint4*>(v.bop), n, origin, sample, d_var, nvar, d_codes);
    MSQ_HIP(hipGetLastError());
    std::vector<uint8_t> codes((size_t)sample * nvar);
    MSQ_HIP(hipMemcpyAsync(codes.data(), d_codes, codes.size(), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<std::vector<uint8_t>> haps;
    const int K = cluster_sample(codes, sample, nvar, haps);
    if (K < 2) return MSQ_OK;
    // the haplotypes as planes for the assignment: the guide's base everywhere, the group's own at the variant columns
    std::vector<uint32_t> planes((size_t)K * words * 2), varmask(words, 0);
    for (int h = 0; h < K; ++h)
        for (uint32_t w = 0; w < words; ++w) {
            planes[((size_t)h * words + w) * 2] = guide[(size_t)w * 8 + 1];
            planes[((size_t)h * words + w) * 2 + 1] = guide[(size_t)w * 8 + 2];
        }
    for (uint32_t j = 0; j < nvar; ++j) {
        const uint32_t w = var_cols[j] >> 5, bit = 1u << (var_cols[j] & 31u);
        varmask[w] |= bit;
        for (int h = 0; h < K; ++h) {
            const uint8_t c = haps[h][j];
            if (c > 3) continue;
            uint32_t* p = &planes[((size_t)h * words + w) * 2];
            p[0] = (p[0] & ~bit) | ((c & 1) ? bit : 0u);
            p[1] = (p[1] & ~bit) | ((c & 2) ? bit : 0u);
        }
    }
    uint32_t *d_planes = nullptr, *d_mask = nullptr, *d_idx = nullptr, *d_counts = nullptr;
    uint8_t *d_hap = nullptr, *d_hap_sorted = nullptr;
    MSQ_HIP(tmp.alloc((void**)&d_planes, planes.size() * 4));
    MSQ_HIP(tmp.alloc((void**)&d_mask, (size_t)words * 4));
    MSQ_HIP(tmp.alloc((void**)&d_idx, n * 4));
    MSQ_HIP(tmp.alloc((void**)&d_counts, MSQ_MAX_HAPS * 4));
    MSQ_HIP(tmp.alloc((void**)&d_hap, n));
    MSQ_HIP(tmp.alloc((void**)&d_hap_sorted, n));
    MSQ_HIP(hipMemcpyAsync(d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    MSQ_HIP(hipMemcpyAsync(d_mask, varmask.data(), (size_t)words * 4, hipMemcpyHostToDevice, ctx->stream));
    MSQ_HIP(hipMemsetAsync(d_counts, 0, MSQ_MAX_HAPS * 4, ctx->stream));
    const uint64_t blocks = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->n_cu * 16);
    hipLaunchKernelGGL(align_assign_kernel, dim3((unsigned)std::max<uint64_t>(1, blocks)), dim3(256), 0, ctx->stream, v.ref_start, v.group_off,
                       reinterpret_cast<const uint4*>(v.bop), n, origin, d_planes, d_mask, words, K, d_hap, d_idx, d_counts);
    MSQ_HIP(hipGetLastError());
    // the reads listed group by group (a stable sort by group: ascending inside a group)
    MSQ_HIP(hipMalloc(&d->ptrs[11], n * 4));
    size_t sort_bytes = 0;
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint8_t*)d_hap, d_hap_sorted, (const uint32_t*)d_idx, (uint32_t*)d->ptrs[11], (int)n, 0, 3,
                                               ctx->stream));
    void* d_sort = nullptr;
    MSQ_HIP(tmp.alloc(&d_sort, sort_bytes));
    MSQ_HIP(hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, (const uint8_t*)d_hap, d_hap_sorted, (const uint32_t*)d_idx, (uint32_t*)d->ptrs[11], (int)n, 0, 3,
                                               ctx->stream));
    uint32_t counts[MSQ_MAX_HAPS];
    MSQ_HIP(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    // every group's own guide + alt, from its own reads
    MSQ_HIP(hipMalloc(&d->ptrs[10], (size_t)K * words * 32));
    MSQ_HIP(hipMemsetAsync(d->ptrs[10], 0, (size_t)K * words * 32, ctx->stream));
    uint64_t lo = 0;
    for (int h = 0; h < K; ++h) {
        if (counts[h]) {
            const uint32_t smp = (uint32_t)std::min<uint64_t>(counts[h], 2048);
            hipLaunchKernelGGL(align_guide_kernel, dim3(words), dim3(256), 0, ctx->stream, v.ref_start, v.group_off, reinterpret_cast<const uint4*>(v.bop),
                               (uint64_t)counts[h], origin, smp, std::max<uint32_t>(3, smp / 256), static_cast<uint32_t*>(d->ptrs[10]) + (size_t)h * words * 8,
                               static_cast<const uint32_t*>(d->ptrs[11]) + lo);
            MSQ_HIP(hipGetLastError());
        }
        lo += counts[h];
    }
    if (lo != n) return msq_fail(MSQ_ERR_HIP, "msq_reads_align: the haplotype groups hold %llu of %llu reads", (unsigned long long)lo, (unsigned long long)n);
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    v.hap_count = (uint32_t)K;
    v.hap_guides = static_cast<const uint32_t*>(d->ptrs[10]);
    v.hap_perm = static_cast<const uint32_t*>(d->ptrs[11]);
    for (int h = 0; h < MSQ_MAX_HAPS; ++h) v.hap_reads[h] = h < K ? counts[h] : 0u;
    return MSQ_OK;
}

}  // namespace

extern "C" int msq_reads_align(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t origin,
                               msq_packed_dev** out)
{
    if (!ctx || !d_reads || !thr || !out) return msq_fail(MSQ_ERR_ARG, "msq_reads_align: NULL argument");
    *out = nullptr;
    if (d_reads->n_reads && (!d_reads->ref_start || !d_reads->ref_end || !d_reads->group_off || (d_reads->n_groups && !d_reads->bop)))
        return msq_fail(MSQ_ERR_ARG, "msq_reads_align: NULL read arrays");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_reads_align")) {
        if (d_reads->n_reads) return rc;
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    // thresholds first (failing match bases become the literal 'N' AddRead writes): a set without QV planes
    const uint8_t* planes[4] = {d_reads->qual, d_reads->delqv, d_reads->subqv, d_reads->insqv};
    const int32_t thrs[4] = {thr->qual, thr->del, thr->sub, thr->ins};
    bool masks = false;
    for (int k = 0; k < 4; ++k) masks = masks || (thrs[k] >= 0 && planes[k]);
    msq_packed_dev* baked = nullptr;
    const msq_packed_reads* src = d_reads;
    if (masks && d_reads->n_reads) {
        if (int rc = msq_reads_bake_qv(ctx, d_reads, thr, &baked)) return rc;
        src = msq_packed_dev_view(baked);
    }
    struct Cleanup {
        msq_ctx* ctx;
        msq_packed_dev* baked;
        void* tmp[3] = {nullptr, nullptr, nullptr};
        ~Cleanup()
        {
            (void)hipStreamSynchronize(ctx->stream);
            for (void* p : tmp)
                if (p) (void)hipFree(p);
            if (baked) msq_packed_dev_free(ctx, baked);
        }
    } keep{ctx, baked};
    const uint64_t n = src->n_reads;
    msq_packed_dev* d = new (std::nothrow) msq_packed_dev;
    if (!d) return msq_fail(MSQ_ERR_NOMEM, "msq_reads_align: out of memory");
    auto fail = [&](int code, const char* what, hipError_t e) {
        free_dev(d);
        return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : code, "msq_reads_align: %s: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    // ref_start / ref_end as they are; new group offsets
    if ((e = hipMalloc(&d->ptrs[0], n ? n * 4 : 16)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
    if ((e = hipMalloc(&d->ptrs[1], n ? n * 4 : 16)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
    if ((e = hipMalloc(&d->ptrs[2], (n + 1) * 8)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
    uint64_t total = 0;
    uint32_t acc_h[4] = {0, 0, 0, 0};
    if (n) {
        if ((e = hipMemcpyAsync(d->ptrs[0], src->ref_start, n * 4, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "copy", e);
        if ((e = hipMemcpyAsync(d->ptrs[1], src->ref_end, n * 4, hipMemcpyDeviceToDevice, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "copy", e);
        uint64_t* d_cnt = nullptr;
        uint32_t* d_acc = nullptr;
        if ((e = hipMalloc(&keep.tmp[0], (n + 1) * 8 + 64)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
        d_cnt = static_cast<uint64_t*>(keep.tmp[0]);
        d_acc = reinterpret_cast<uint32_t*>(d_cnt + n + 1);
        if ((e = hipMemsetAsync(d_acc, 0, 16, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "memset", e);
        const uint64_t blocks = std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->n_cu * 16);
        hipLaunchKernelGGL(align_count_kernel, dim3((unsigned)std::max<uint64_t>(1, blocks)), dim3(256), 0, ctx->stream, src->ref_start, src->group_off,
                           reinterpret_cast<const uint4*>(src->bop), n, origin, d_cnt, d_acc);
        if ((e = hipGetLastError()) != hipSuccess) return fail(MSQ_ERR_HIP, "count kernel", e);
        size_t tmp_bytes = 0;
        if ((e = hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const uint64_t*)d_cnt, (uint64_t*)d->ptrs[2], (int)(n + 1), ctx->stream)) != hipSuccess)
            return fail(MSQ_ERR_HIP, "scan", e);
        if ((e = hipMalloc(&keep.tmp[1], tmp_bytes ? tmp_bytes : 16)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
        if ((e = hipcub::DeviceScan::ExclusiveSum(keep.tmp[1], tmp_bytes, (const uint64_t*)d_cnt, (uint64_t*)d->ptrs[2], (int)(n + 1), ctx->stream)) != hipSuccess)
            return fail(MSQ_ERR_HIP, "scan", e);
        if ((e = hipMemcpyAsync(&total, (const uint64_t*)d->ptrs[2] + n, 8, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "copy", e);
        if ((e = hipMemcpyAsync(acc_h, d_acc, 16, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "copy", e);
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "sync", e);
        if (acc_h[2]) {
            free_dev(d);
            return msq_fail(MSQ_ERR_ARG, "msq_reads_align: a read starts left of the origin %d", origin);
        }
    } else {
        if ((e = hipMemsetAsync(d->ptrs[2], 0, 8, ctx->stream)) != hipSuccess) return fail(MSQ_ERR_HIP, "memset", e);
    }
    const uint32_t span_words = acc_h[0], max_words = acc_h[1];
    if ((size_t)(max_words + 1) * 16 > 150 * 1024) {
        free_dev(d);
        return msq_fail(MSQ_ERR_ARG, "msq_reads_align: a read spans %u columns, more than one LDS row holds", max_words * 32);
    }
    if ((e = hipMalloc(&d->ptrs[3], total ? total * 16 : 16)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
    if (total) {
        const int row_words = (int)max_words + 1;
        int waves = (int)std::min<size_t>(4, (150 * 1024) / ((size_t)row_words * 16));
        if (waves < 1) waves = 1;
        const size_t lds = (size_t)waves * row_words * 16;
        if (lds > 48 * 1024)
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(align_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
                return fail(MSQ_ERR_HIP, "LDS size", e);
        const uint64_t blocks = std::min<uint64_t>((n + waves - 1) / waves, (uint64_t)ctx->n_cu * 8);
        hipLaunchKernelGGL(align_rows_kernel, dim3((unsigned)std::max<uint64_t>(1, blocks)), dim3(64 * waves), lds, ctx->stream, src->ref_start,
                           src->group_off, reinterpret_cast<const uint4*>(src->bop), n, origin, (src->flags & MSQ_READS_NORMALIZED) ? 1u : 0u,
                           (const uint64_t*)d->ptrs[2], static_cast<uint4*>(d->ptrs[3]), row_words);
        if ((e = hipGetLastError()) != hipSuccess) return fail(MSQ_ERR_HIP, "row kernel", e);
    }
    msq_packed_reads& v = d->view;
    v.n_reads = n;
    v.n_groups = total;
    v.ref_start = (const int32_t*)d->ptrs[0];
    v.ref_end = (const int32_t*)d->ptrs[1];
    v.group_off = (const uint64_t*)d->ptrs[2];
    v.bop = (const uint32_t*)d->ptrs[3];
    v.qual = v.delqv = v.subqv = v.insqv = nullptr;
    // the slots keep their codes: exotic symbols stay exotic, baked thresholds stay baked; the layout is the aligned
    // one (no longer normalized, no marked padding); inserted bases are gone and with them MSQ_READS_INS_EXOTIC
    v.flags = (src->flags & (MSQ_READS_HAS_EXOTIC | MSQ_READS_QV_BAKED)) | MSQ_READS_ALIGNED;
    v.align_origin = origin;
    v.gcol = nullptr;
    v.guide = nullptr;
    v.guide_words = 0;
    if (span_words) {
        if ((e = hipMalloc(&d->ptrs[9], (size_t)span_words * 32)) != hipSuccess) return fail(MSQ_ERR_HIP, "hipMalloc", e);
        const uint32_t sample = (uint32_t)std::min<uint64_t>(n, 2048);
        const uint32_t alt_min = std::max<uint32_t>(3, sample / 256);
        hipLaunchKernelGGL(align_guide_kernel, dim3(span_words), dim3(256), 0, ctx->stream, v.ref_start, v.group_off, reinterpret_cast<const uint4*>(v.bop), n,
                           origin, sample, alt_min, static_cast<uint32_t*>(d->ptrs[9]), (const uint32_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return fail(MSQ_ERR_HIP, "guide kernel", e);
        v.guide = static_cast<const uint32_t*>(d->ptrs[9]);
        v.guide_words = span_words;
        if (n < 0xffffffffull && !ctx->no_haps) {
            const int rc = align_haplotypes(ctx, d, origin, sample);
            if (rc != MSQ_OK) {
                (void)hipStreamSynchronize(ctx->stream);
                free_dev(d);
                return rc;
            }
        }
    }
    if (int rc = msq_packed_dev_index(ctx, d)) {  // the generic kernels (phasing lookup, the read-space pileup) take it as any set
        (void)hipStreamSynchronize(ctx->stream);
        free_dev(d);
        return rc;
    }
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    *out = d;
    return MSQ_OK;
}
