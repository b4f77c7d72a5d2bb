// msq_cells.hip -- per-(position, codon) cell evaluation: CountNumberOfTests + the cell loop of
// AminoAcidCaller::CallVariants (src/AminoAcidCaller.cpp:86-103, 397-488) and Fisher::fisher_exact_tiss
// (src/Fisher.cpp:12-143) on a host-built log-factorial table.
//
// One workgroup per codon position, one lane per codon (64 lanes == 64 codons): coverage, the number of
// distinct codons and the majority codon are wave reductions; the position's cells are shared among the
// workgroup's four waves, each Fisher sum spread over the 64 lanes of its wave.
// FP64 throughout; compiled with -ffp-contract=off so every multiply/add/divide rounds exactly as the
// reference's scalar code does.
#include "msq_internal.h"

#include <new>
#include <string>
#include <vector>

struct msq_plan {
    int64_t P = 0;
    int32_t begin1 = 0, W = 0;
    bool has_reference = false;
    bool range_error = false;   // some position makes referenceSequence.substr(absPos, 3) throw
    int32_t* d_pos = nullptr;   // [P][4]: gene, i, aa_pos, ref codon from the reference sequence (-1 none)
};

static inline int codon_index_host(const char* s, int64_t n)
{
    if (n != 3) return -1;
    int idx = 0;
    for (int k = 0; k < 3; ++k) {
        int t;
        switch (s[k]) {
            case 'A': t = 0; break;
            case 'C': t = 1; break;
            case 'G': t = 2; break;
            case 'T': t = 3; break;
            default: return -1;
        }
        idx = idx * 4 + t;
    }
    return idx;
}

extern "C" int msq_plan_create(msq_ctx* ctx, int32_t n_genes, const int32_t* gene_begin, const int32_t* gene_end,
                               const char* ref_seq, int64_t ref_len, int32_t begin1, int32_t W, msq_plan** out)
{
    if (!ctx || !out) return msq_fail(MSQ_ERR_ARG, "msq_plan_create: NULL argument");
    *out = nullptr;
    if (n_genes < 0 || (n_genes > 0 && (!gene_begin || !gene_end)) || W < 0 || ref_len < 0)
        return msq_fail(MSQ_ERR_ARG, "msq_plan_create: bad argument");
    MSQ_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> gb, ge;
    if (n_genes == 0) {  // "Unnamed ORF" over the whole window, src/AminoAcidCaller.cpp:402-405
        gb.push_back(begin1);
        ge.push_back(begin1 + W);
    } else {
        gb.assign(gene_begin, gene_begin + n_genes);
        ge.assign(gene_end, gene_end + n_genes);
    }
    msq_plan* p = new (std::nothrow) msq_plan;
    if (!p) return msq_fail(MSQ_ERR_NOMEM, "msq_plan_create: out of memory");
    p->begin1 = begin1;
    p->W = W;
    p->has_reference = ref_seq != nullptr && ref_len > 0;  // !referenceSequence.empty(), :411
    std::vector<int32_t> pos;
    for (size_t g = 0; g < gb.size(); ++g) {
        for (int64_t i = gb[g]; i < (int64_t)ge[g] - 2; i += 3) {  // :417-423, relPos % 3 == 0
            int32_t ref = -1;
            if (p->has_reference) {
                const int64_t absPos = i - 1;  // :419
                if (absPos < 0 || absPos > ref_len) {
                    p->range_error = true;  // std::string::substr throws std::out_of_range, :446
                } else {
                    const int64_t n = ref_len - absPos < 3 ? ref_len - absPos : 3;
                    ref = codon_index_host(ref_seq + absPos, n);
                }
            }
            pos.push_back((int32_t)g);
            pos.push_back((int32_t)i);
            pos.push_back((int32_t)(1 + (i - gb[g]) / 3));  // aaPos, :427
            pos.push_back(ref);
        }
    }
    p->P = (int64_t)pos.size() / 4;
    if (p->P) {
        hipError_t e = hipMalloc(&p->d_pos, pos.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemcpy(p->d_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (p->d_pos) (void)hipFree(p->d_pos);
            delete p;
            return msq_fail(MSQ_ERR_HIP, "msq_plan_create: %s", hipGetErrorString(e));
        }
    }
    *out = p;
    return MSQ_OK;
}
extern "C" int64_t msq_plan_positions(const msq_plan* plan) { return plan ? plan->P : -1; }
extern "C" int msq_plan_destroy(msq_ctx* ctx, msq_plan* plan)
{
    if (!plan) return MSQ_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (plan->d_pos) (void)hipFree(plan->d_pos);
    delete plan;
    return MSQ_OK;
}

// ---------------------------------------------------------------------------------------------- Fisher
struct LFact {
    const double* tab;
    int64_t max_n;
};

// factln, src/Fisher.cpp:115-125 (table = lgamma(n + 1.0) from the host libm)
__device__ __forceinline__ double d_factln(const LFact& lf, int n, uint32_t* err)
{
    if (n <= 1) return 0.0;
    if (n > lf.max_n) {
        *err |= DEVERR_RANGE;
        return 0.0;
    }
    return lf.tab[n];
}
// binomialln, src/Fisher.cpp:127: (f(n) - f(k)) - f(n-k)
__device__ __forceinline__ double d_binomialln(const LFact& lf, int n, int k, uint32_t* err)
{
    return (d_factln(lf, n, err) - d_factln(lf, k, err) - d_factln(lf, n - k, err));
}
// factorInc / factorDec, src/Fisher.cpp:79-92
__device__ __forceinline__ double d_factor_inc(int a, int b, int c, int d)
{
    double f = (double)b * c;
    f /= (double)(a + 1) * (d + 1);
    return f;
}
__device__ __forceinline__ double d_factor_dec(int a, int b, int c, int d)
{
    double f = (double)a * d;
    f /= (double)(c + 1) * (b + 1);
    return f;
}
// fisher_exact_tiss, src/Fisher.cpp:12-77
__device__ double d_fisher(const LFact& lf, int a, int b, int c, int d, uint32_t* err)
{
    int co = a;
    const int ga = a + b, gb = a + c, total = a + b + c + d;
    int min_co = 0;
    if (ga + gb > total) min_co = ga + gb - total;
    const int max_co = ga < gb ? ga : gb;
    // calc_hypergeom, src/Fisher.cpp:129-143
    const double b1 = d_binomialln(lf, a + b, a, err);
    const double b2 = d_binomialln(lf, c + d, c, err);
    const double b3 = d_binomialln(lf, total, a + c, err);
    double base_p = exp(b1 + b2 - b3);
    if (!(co == max_co || co == min_co)) {
        double fi = d_factor_inc(a, b, c, d);
        const double fd = d_factor_dec(a, b, c, d);
        double curr = base_p;
        if (fi < fd) {
            do {
                curr *= fi;
                base_p += curr;
                co++;
                a++; d++; b--; c--;
                fi = d_factor_inc(a, b, c, d);
                // once curr is +0 every further term adds exactly 0: leaving early changes no bit
            } while (co < max_co && curr != 0.0);
        } else {
            base_p = 1.0;  // "saddle point", src/Fisher.cpp:71-74
        }
    }
    return base_p;
}

// fisher_exact_tiss for ONE table evaluated by a whole wave: all 64 lanes call it with the same (a, b, c, d).
// The reference's loop, p = base_p + sum_k base_p * prod_{j <= k} factorInc_j (src/Fisher.cpp:50-70), runs up to
// ceil(expected) ~ 1.7e-4 * coverage dependent steps (175 at 1e6 reads, 1400 at 8e6): the latency of the testing
// pass, and it grows with the node's read count.  Here lane L owns a block of the steps: block products, an
// exclusive wave prefix product, block sums in units of base_p, a wave sum.  Same terms, another association:
// a few thousand ulp at most from the sequential sum, far inside the 1e-9 relative the path allows; the rules
// before the loop (point probability at the table edge, "saddle point" -> 1) are the reference's, unchanged.
// base_p = the table's point probability (calc_hypergeom), computed beforehand by the cell's own lane so that the
// log-factorial lookups of all the cells of a position are in flight together.
// msq_fisher / msq_fisher_batch keep the sequential d_fisher (bit-exact unit door).
__device__ double d_hypergeom(const LFact& lf, int a, int b, int c, int d, uint32_t* err)  // src/Fisher.cpp:129-143
{
    const double b1 = d_binomialln(lf, a + b, a, err);
    const double b2 = d_binomialln(lf, c + d, c, err);
    const double b3 = d_binomialln(lf, a + b + c + d, a + c, err);
    return exp(b1 + b2 - b3);
}
__device__ double d_fisher_wave(double base_p, int a, int b, int c, int d, int lane)
{
    const int ga = a + b, gb = a + c, total = a + b + c + d;
    int min_co = 0;
    if (ga + gb > total) min_co = ga + gb - total;
    const int max_co = ga < gb ? ga : gb;
    if (a == max_co || a == min_co) return base_p;
    if (!(d_factor_inc(a, b, c, d) < d_factor_dec(a, b, c, d))) return 1.0;  // src/Fisher.cpp:71-74
    const int n = max_co - a;  // steps k = 0 .. n-1 multiply by factorInc(a+k, b-k, c-k, d+k); n = min(b, c) >= 1
    const int B = (n + 63) >> 6;
    const int k0 = min(n, lane * B), k1 = min(n, k0 + B);
    double prod = 1.0, lsum = 0.0;  // the block's product and its terms summed in units of the product before the block
    for (int k = k0; k < k1; ++k) {
        prod *= d_factor_inc(a + k, b - k, c - k, d + k);
        lsum += prod;
    }
    double incl = prod;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        const double t = __shfl_up(incl, dd);
        if (lane >= dd) incl *= t;
    }
    double before = __shfl_up(incl, 1);  // product of every step before this lane's block
    if (lane == 0) before = 1.0;
    double sum = before * lsum;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) sum += __shfl_xor(sum, dd);
    return base_p + base_p * sum;
}

__global__ __launch_bounds__(256) void fisher_batch_kernel(LFact lf, int64_t n, const int32_t* __restrict__ a,
                                                           const int32_t* __restrict__ b, const int32_t* __restrict__ c,
                                                           const int32_t* __restrict__ d, double* __restrict__ p,
                                                           uint32_t* __restrict__ err)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e = 0;
    p[i] = d_fisher(lf, a[i], b[i], c[i], d[i], &e);
    if (e) atomicOr(err, e);
}

// ------------------------------------------------------------------------------------------ cell kernels
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

// CodonsAt(winPos) as one lane per codon: zero outside 1 <= winPos <= W-3 (src/MSA.cpp:193,207)
__device__ __forceinline__ int codon_count(const uint32_t* __restrict__ codon, int win_pos, int W, int lane)
{
    return (win_pos >= 1 && win_pos + 2 < W) ? (int)codon[(size_t)win_pos * 64 + lane] : 0;
}

// CountNumberOfTests, src/AminoAcidCaller.cpp:86-103: sum over positions of the number of distinct codons; 0 -> 1
// (:102).  acc = the context's small accumulators (sum, ticket), all zero between kernels: the workgroup that
// draws the last ticket stores the finished number to n_tests and clears them.
__global__ __launch_bounds__(1024) void count_tests_kernel(const int32_t* __restrict__ pos, int64_t P, int32_t begin1,
                                                          int32_t W, const uint32_t* __restrict__ codon,
                                                          uint32_t* __restrict__ acc, int32_t* __restrict__ n_tests)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    int cnt = 0;
    if (p < P) cnt = codon_count(codon, pos[p * 4 + 1] - begin1, W, lane);
    const unsigned long long nz = __ballot(cnt > 0);
    __shared__ int s_n[16];
    if (lane == 0) s_n[threadIdx.x >> 6] = __popcll(nz);
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per block: same-address atomics serialise
        int n = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) n += s_n[k];
        if (n) atomicAdd(&acc[0], (uint32_t)n);
        __threadfence();
        if (atomicAdd(&acc[1], 1u) == gridDim.x - 1) {
            __threadfence();
            const uint32_t total = atomicExch(&acc[0], 0u);
            atomicExch(&acc[1], 0u);
            *n_tests = total == 0 ? 1 : (int32_t)total;
        }
    }
}

__global__ __launch_bounds__(256) void call_cells_kernel(const int32_t* __restrict__ pos, int64_t P, int32_t begin1,
                                                         int32_t W, int has_reference,
                                                         const uint32_t* __restrict__ codon, LFact lf,
                                                         msq_error_model em, double maximal_perc,
                                                         const int32_t* __restrict__ n_tests_raw,
                                                         msq_position* __restrict__ out_pos, double* __restrict__ out_p,
                                                         double* __restrict__ out_expected, uint32_t* __restrict__ err)
{
    // One workgroup per codon position; its four waves all work out the position's summary (cheap) and share the
    // position's cells -- the Fisher sums are a serial chain per cell, and a grid of one wave per position leaves
    // most SIMDs with nothing to overlap it with.
    const int lane = threadIdx.x & 63, wsub = threadIdx.x >> 6, nsub = blockDim.x >> 6;
    const int64_t p = blockIdx.x;
    if (p >= P) return;
    const int number_of_tests = *n_tests_raw;  // already 0 -> 1 (:102)
    const int gene = pos[p * 4 + 0], i = pos[p * 4 + 1], aa_pos = pos[p * 4 + 2], ref_from_seq = pos[p * 4 + 3];
    const int win_pos = i - begin1;  // :425
    const int cnt = codon_count(codon, win_pos, W, lane);
    const int coverage = wave_sum(cnt);  // :436-438
    // FindMajorityCodon, :380-395: strict '>' walking the map in codon order -> smallest codon among the maxima
    const int mx = wave_max(cnt);
    int maj = -1, maj_count = 0;
    if (mx > 0) {
        maj = __ffsll((long long)__ballot(cnt == mx)) - 1;
        maj_count = mx;
    }
    int ref = -1, alt = -1, skipped = 0;
    if (has_reference) {  // :444-458
        ref = ref_from_seq;
        if (ref < 0)
            skipped = 1;
        else if (maj_count == 0)
            skipped = 1;
        else if (maj_count * 100.0 / coverage > maximal_perc)
            alt = maj;
    } else {  // :459-463
        if (maj_count == 0)
            skipped = 1;
        else
            ref = maj;
    }
    if (lane == 0 && wsub == 0) {
        msq_position o;
        o.gene = gene; o.i = i; o.aa_pos = aa_pos; o.win_pos = win_pos; o.coverage = coverage;
        o.ref_codon = ref; o.alt_codon = alt; o.maj_codon = maj; o.maj_count = maj_count; o.skipped = skipped;
        out_pos[p] = o;
    }
    double pv = -1.0, expected = 0.0;
    const bool cell = !skipped && cnt > 0 && lane != ref && lane != alt;  // :465-472
    int tc = 0, td = 0;
    if (cell) {
        // Probability(refCodon, codon), :325-339: left-to-right product starting from 1
        double pr = 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int sh = 4 - 2 * k;
            pr *= (((ref >> sh) & 3) != ((lane >> sh) & 3)) ? em.substitution : em.match;
        }
        expected = coverage * pr;  // :477-478
        tc = (int)ceil(expected);  // :481-485; std::ceil of an int is the int
        td = (int)ceil(coverage - expected);
    }
    // every cell's point probability by its own lane (the table lookups of the position overlap), then the cells
    // one after the other, each summation spread over the wave's 64 lanes
    uint32_t e = 0;
    const unsigned long long cells = __ballot(cell);
    const bool mine = cell && (__popcll(cells & ((1ull << lane) - 1ull)) % nsub) == wsub;  // this wave's share
    const double base_p = mine ? d_hypergeom(lf, cnt, coverage - cnt, tc, td, &e) : 0.0;
    unsigned long long todo = __ballot(mine);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const double r = d_fisher_wave(__shfl(base_p, src), __shfl(cnt, src), __shfl(coverage - cnt, src), __shfl(tc, src),
                                       __shfl(td, src), lane);
        if (lane == src) {
            pv = r * number_of_tests;
            if (pv > 1) pv = 1;  // :488
        }
    }
    if (e) atomicOr(err, e);
    if (mine || (!cell && wsub == 0)) out_p[p * 64 + lane] = pv;  // a cell by the wave that summed it, -1 by wave 0
    if (out_expected && wsub == 0) out_expected[p * 64 + lane] = expected;
}

extern "C" int msq_call_cells(msq_ctx* ctx, const msq_plan* plan, const uint32_t* d_codon, const msq_error_model* em,
                              double maximal_perc, msq_position* d_positions, double* d_p, double* d_expected,
                              int32_t* d_n_tests)
{
    if (!ctx || !plan || !em || !d_n_tests) return msq_fail(MSQ_ERR_ARG, "msq_call_cells: NULL argument");
    if (plan->range_error)  // std::out_of_range from referenceSequence.substr, src/AminoAcidCaller.cpp:446
        return msq_fail(MSQ_ERR_RANGE, "basic_string::substr: a gene position lies outside the reference sequence");
    MSQ_HIP(hipSetDevice(ctx->device));
    if (plan->P == 0) {  // nothing to test: numberOfTests == 0 -> 1 (:102)
        const int32_t one = 1;
        MSQ_HIP(hipMemcpyAsync(d_n_tests, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
        return MSQ_OK;
    }
    if (!d_codon || !d_positions || !d_p) return msq_fail(MSQ_ERR_ARG, "msq_call_cells: NULL device buffer");
    if (!ctx->d_lfact) {
        int rc = msq_lfact_reserve(ctx, 4096);
        if (rc != MSQ_OK) return rc;
    }
    const unsigned blocks = (unsigned)plan->P;
    hipLaunchKernelGGL(count_tests_kernel, dim3((unsigned)((plan->P + 15) / 16)), dim3(1024), 0, ctx->stream, plan->d_pos, plan->P, plan->begin1,
                       plan->W, d_codon, ctx->d_small + 8, d_n_tests);
    MSQ_HIP(hipGetLastError());
    LFact lf{ctx->d_lfact, ctx->lfact_n};
    hipLaunchKernelGGL(call_cells_kernel, dim3(blocks), dim3(256), 0, ctx->stream, plan->d_pos, plan->P, plan->begin1,
                       plan->W, plan->has_reference ? 1 : 0, d_codon, lf, *em, maximal_perc, d_n_tests, d_positions, d_p,
                       d_expected, ctx->d_err);
    msq_timing_mark_path_end(ctx);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

extern "C" int msq_call_check(msq_ctx* ctx)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_call_check: ctx is NULL");
    uint32_t e = 0;
    MSQ_HIP(hipMemcpyAsync(&e, ctx->d_err, sizeof(e), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    if (e & DEVERR_RANGE) {
        const uint32_t keep = e & ~DEVERR_RANGE;
        MSQ_HIP(hipMemcpyAsync(ctx->d_err, &keep, sizeof(keep), hipMemcpyHostToDevice, ctx->stream));
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        return msq_fail(MSQ_ERR_RANGE, "log-factorial table too small (have n <= %lld): call msq_lfact_reserve(2*coverage+1)",
                        (long long)ctx->lfact_n);
    }
    return MSQ_OK;
}

extern "C" int msq_fisher_batch(msq_ctx* ctx, int64_t n, const int32_t* chi11, const int32_t* chi12,
                                const int32_t* chi21, const int32_t* chi22, double* p_out)
{
    if (!ctx || n < 0) return msq_fail(MSQ_ERR_ARG, "msq_fisher_batch: bad argument");
    if (n == 0) return MSQ_OK;
    if (!chi11 || !chi12 || !chi21 || !chi22 || !p_out) return msq_fail(MSQ_ERR_ARG, "msq_fisher_batch: NULL array");
    MSQ_HIP(hipSetDevice(ctx->device));
    int64_t max_total = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (chi11[i] < 0 || chi12[i] < 0 || chi21[i] < 0 || chi22[i] < 0)
            return msq_fail(MSQ_ERR_ARG, "msq_fisher_batch: negative cell in table %lld", (long long)i);
        const int64_t t = (int64_t)chi11[i] + chi12[i] + chi21[i] + chi22[i];
        if (t > INT32_MAX) return msq_fail(MSQ_ERR_ARG, "msq_fisher_batch: table %lld total overflows int", (long long)i);
        if (t > max_total) max_total = t;
    }
    int rc = msq_lfact_reserve(ctx, max_total);
    if (rc != MSQ_OK) return rc;
    void* s = nullptr;
    const size_t in_bytes = (size_t)n * sizeof(int32_t);
    const size_t in_pad = (in_bytes + 15) & ~(size_t)15;
    rc = msq_scratch(ctx, 4 * in_pad + (size_t)n * sizeof(double), &s);
    if (rc != MSQ_OK) return rc;
    char* base = (char*)s;
    const int32_t* src[4] = {chi11, chi12, chi21, chi22};
    for (int k = 0; k < 4; ++k)
        MSQ_HIP(hipMemcpyAsync(base + k * in_pad, src[k], in_bytes, hipMemcpyHostToDevice, ctx->stream));
    double* d_p = (double*)(base + 4 * in_pad);
    LFact lf{ctx->d_lfact, ctx->lfact_n};
    hipLaunchKernelGGL(fisher_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lf, n,
                       (const int32_t*)(base), (const int32_t*)(base + in_pad), (const int32_t*)(base + 2 * in_pad),
                       (const int32_t*)(base + 3 * in_pad), d_p, ctx->d_err);
    MSQ_HIP(hipGetLastError());
    MSQ_HIP(hipMemcpyAsync(p_out, d_p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return msq_call_check(ctx);
}

extern "C" int msq_fisher(msq_ctx* ctx, int32_t chi11, int32_t chi12, int32_t chi21, int32_t chi22, double* p_out)
{
    return msq_fisher_batch(ctx, 1, &chi11, &chi12, &chi21, &chi22, p_out);
}
