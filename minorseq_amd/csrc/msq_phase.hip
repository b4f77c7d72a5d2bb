// msq_phase.hip -- the per-read gather of AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:149-160):
// for every read, MSARow::CodonAt at each variant position.  The fast path is the streaming pileup kernel in its
// PHASE mode (msq_pileup_stream.hip); reads with exotic ops take the kernel below: one wave per read builds the
// read's row in LDS (msq_rowbuild.h) and reads the requested cells back.  Clustering the keys is the caller's.
#include "msq_internal.h"
#include "msq_rowbuild.h"

int msq_phase_stream_launch(msq_ctx* ctx, const msq_packed_reads& R, const msq_qv_thresholds& thr, int32_t win_begin0,
                            int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys, bool* handled);

namespace {

__global__ __launch_bounds__(256) void phase_codons_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                           const uint4* __restrict__ bop, QvPlanes qv, uint64_t n_reads,
                                                           int32_t win_begin0, int32_t W, int32_t row_bytes, int32_t waves_per_wg,
                                                           int32_t n_pos, const int32_t* __restrict__ pos, uint16_t* __restrict__ keys,
                                                           uint32_t* __restrict__ err)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t* row = lds + (size_t)wave * row_bytes;
    const uint64_t total_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint64_t iters = (n_reads + total_waves - 1) / total_waves;
    uint32_t my_err = 0;
    for (uint64_t it = 0; it < iters; ++it) {
        const uint64_t r = it * total_waves + (uint64_t)blockIdx.x * waves_per_wg + wave;
        const bool active = r < n_reads;
        for (int c = lane * 4; c < row_bytes; c += 256) *reinterpret_cast<uint32_t*>(row + c) = 0;  // MSARow(size): ' '
        __syncthreads();
        if (active) my_err |= msq_build_row(row, lane, ref_start, group_off, bop, qv, r, win_begin0, W);
        __syncthreads();
        if (active) {
            for (int k = lane; k < n_pos; k += 64) {
                const int p = pos[k];
                uint32_t key = 0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {  // MSARow::CodonAt: indices with i > 0 && i < length only (MSA.cpp:188-195)
                    const int i = p + j;
                    const uint32_t sym = (i > 0 && i < W) ? row[i] : 7u;
                    key |= sym << (3 * j);
                }
                keys[r * (uint64_t)n_pos + k] = (uint16_t)key;
            }
        }
        __syncthreads();
    }
    // a row the pileup accepted cannot fail here; keep the error word honest anyway
    if (my_err) atomicOr(err, my_err);
}

// ---- the gather by LOOKUP: with the group column index a read's symbol at a column is found without building its row.
// Thread = (read, position); the group that holds reference column c is at or behind group (c - ReferenceStart) / 32 of
// its read (a group consumes at most 32 columns), usually right there (only inserted bases push it back), and the
// index says where every group starts.  Touches K groups of a read instead of all of them, knows no column slabs:
// what pays when the positions are few against the read length (wide windows).  Sets without QV planes in the loop
// and without exotic symbols only; the symbols are those of the stream kernel's rows.
__device__ __forceinline__ uint32_t kth_set_bit(uint32_t m, uint32_t k)  // position of the k-th (0-based) set bit; k < popc(m)
{
    uint32_t pos = 0;
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const uint32_t low = __popc(m & ((1u << w) - 1u));
        if (k >= low) {
            k -= low;
            m >>= w;
            pos += w;
        }
    }
    return pos;
}

__global__ __launch_bounds__(256) void phase_lookup_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                           const uint4* __restrict__ bop, const uint32_t* __restrict__ gcol,
                                                           uint64_t n_reads, uint32_t normalized, int32_t win_begin0, int32_t W, int32_t n_pos,
                                                           const int32_t* __restrict__ pos, uint16_t* __restrict__ keys,
                                                           uint32_t* __restrict__ err)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = t / (uint32_t)n_pos;
    if (r >= n_reads) return;
    const int k = (int)(t - r * (uint32_t)n_pos);
    const int64_t rs = ref_start[r];
    const uint64_t g0 = group_off[r], g1 = group_off[r + 1];
    const int64_t ng = (int64_t)(g1 - g0);
    auto col_of = [&](int64_t i) -> int64_t { return (int64_t)(gcol[g0 + i] & 0x1fffffffu) - 1; };
    if (k == 0 && ng > 0) {  // a read that leaves the window: assert(pos >= 0) / the write past the row, MSA.cpp:149
        const int64_t end = col_of(ng - 1) + __popc(~bop[g1 - 1].w);
        if (rs < win_begin0 || end > (int64_t)win_begin0 + W) atomicOr(err, (uint32_t)DEVERR_WINDOW);
    }
    const int p = pos[k];
    uint32_t key = 0;
    int64_t gi = -1;   // group of the previous column, its first column and planes: the next column is mostly in it too
    int64_t gc = 0;
    uint4 w = make_uint4(0, 0, 0, ~0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // MSARow::CodonAt: indices with i > 0 && i < length only (MSA.cpp:188-195)
        const int i = p + j;
        uint32_t sym = 7u;
        if (i > 0 && i < W) {
            sym = 0u;  // ' ': the read does not reach the column
            const int64_t c = (int64_t)win_begin0 + i;
            if (c >= rs && ng > 0) {
                int64_t g = gi >= 0 ? gi : (c - rs) >> 5;
                if (g > ng - 1) g = ng - 1;
                // forward to the last group that starts at or before c: a few steps, then by bisection
                int steps = 0;
                while (g + 1 < ng && col_of(g + 1) <= c && steps < 4) {
                    ++g;
                    ++steps;
                }
                if (steps == 4 && g + 1 < ng && col_of(g + 1) <= c) {
                    int64_t lo = g + 1, hi = ng - 1;  // col_of(lo) <= c
                    while (lo < hi) {
                        const int64_t mid = (lo + hi + 1) >> 1;
                        if (col_of(mid) <= c) lo = mid;
                        else hi = mid - 1;
                    }
                    g = lo;
                }
                if (g != gi) {
                    gi = g;
                    gc = col_of(g);
                    w = bop[g0 + g];
                }
                const uint32_t cons = ~w.w;  // ops 0, 1 consume a column
                const int64_t kin = c - gc;
                if (kin >= 0 && kin < __popc(cons)) {
                    const uint32_t b = normalized ? (uint32_t)kin : kth_set_bit(cons, (uint32_t)kin);
                    const uint32_t xb = (w.x >> b) & 1u, yb = (w.y >> b) & 1u, zb = (w.z >> b) & 1u;
                    sym = zb ? 5u + xb : 1u + (xb | (yb << 1));  // op 1: '-' or a literal 'N'; op 0: the nucleotide
                }
            }
        }
        key |= sym << (3 * j);
    }
    keys[t] = (uint16_t)key;
}

}  // namespace

static int g_phase_lookups = 0;  // calls that took the lookup kernel (tests ask which path ran)
extern "C" int msq_debug_phase_lookups(void) { return g_phase_lookups; }

extern "C" int msq_phase_codons(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                                int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys)
{
    if (!ctx || !d_reads || !thr) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: NULL argument");
    if (W < 0 || n_pos < 0) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: negative size");
    if (d_reads->n_reads == 0 || n_pos == 0) return MSQ_OK;
    if (!pos || !d_keys) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: NULL positions / keys");
    if (int rc = msq_validate_reads(d_reads, thr, "msq_phase_codons")) return rc;
    MSQ_HIP(hipSetDevice(ctx->device));
    {  // few positions against the read length, a set with the column index and no QV plane to apply: the lookup
        const uint8_t* planes[4] = {d_reads->qual, d_reads->delqv, d_reads->subqv, d_reads->insqv};
        const int32_t thrs[4] = {thr->qual, thr->del, thr->sub, thr->ins};
        bool qv = false;
        for (int k = 0; k < 4; ++k) qv = qv || (thrs[k] >= 0 && planes[k]);
        // measured (tools/phase_ab.sh, picoseconds per read): the streaming gather 3.8 per group and column slab + 2.6 per
        // position; the lookup 2 per group (the read's index lines, its window check) + 11 per position
        const double groups_per_read = (double)d_reads->n_groups / (double)d_reads->n_reads;
        const int n_slabs = ((W + 31) / 32 + 127) / 128;
        const char* force = getenv("MSQ_PHASE_LOOKUP");  // 1: always (when applicable), 0: never
        // taken when it is at least 15 % ahead of a streaming pass priced without its per-position share (which hides
        // behind the decode for a few dozen positions): pol reads up to 10 positions, the 9.7 kb genome up to ~200
        const bool pays = force ? force[0] == '1' : 11.0 * (double)n_pos < groups_per_read * (0.85 * 3.8 * n_slabs - 2.0);
        if (d_reads->gcol && !qv && !(d_reads->flags & MSQ_READS_HAS_EXOTIC) && d_reads->n_groups && pays && win_begin0 >= 0 &&
            (int64_t)win_begin0 + W < 0x1fffff00 && (uint64_t)d_reads->n_reads * (uint64_t)n_pos < (1ull << 38)) {
            void* s = nullptr;
            int rc = msq_scratch(ctx, (size_t)n_pos * sizeof(int32_t), &s);
            if (rc != MSQ_OK) return rc;
            MSQ_HIP(hipMemcpyAsync(s, pos, (size_t)n_pos * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            const uint64_t threads = d_reads->n_reads * (uint64_t)n_pos;
            hipLaunchKernelGGL(phase_lookup_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_reads->ref_start,
                               d_reads->group_off, reinterpret_cast<const uint4*>(d_reads->bop), d_reads->gcol, d_reads->n_reads,
                               (d_reads->flags & MSQ_READS_NORMALIZED) ? 1u : 0u, win_begin0, W, n_pos, (const int32_t*)s, d_keys, ctx->d_err);
            MSQ_HIP(hipGetLastError());
            MSQ_HIP(hipStreamSynchronize(ctx->stream));  // the positions live in the context's scratch
            ++g_phase_lookups;
            return MSQ_OK;
        }
    }
    {  // fast path: phase 1 of the streaming pileup kernel + a gather from its LDS rows
        bool handled = false;
        int rc = msq_phase_stream_launch(ctx, *d_reads, *thr, win_begin0, W, n_pos, pos, d_keys, &handled);
        if (rc != MSQ_OK) return rc;
        if (handled) {
            MSQ_HIP(hipStreamSynchronize(ctx->stream));
            return MSQ_OK;
        }
    }
    // reads with exotic ops: one wave per read, byte rows (the simple kernel's row builder)
    const int32_t row_bytes = ((W + 2 + 15) / 16) * 16;
    int waves = 150 * 1024 / (row_bytes ? row_bytes : 16);
    if (waves < 1) return msq_fail(MSQ_ERR_ARG, "msq_phase_codons: window of %d columns exceeds one LDS row", W);
    if (waves > 4) waves = 4;
    const size_t lds = (size_t)row_bytes * waves;
    if (lds > 48 * 1024)
        MSQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(phase_codons_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    void* s = nullptr;
    int rc = msq_scratch(ctx, (size_t)n_pos * sizeof(int32_t), &s);
    if (rc != MSQ_OK) return rc;
    MSQ_HIP(hipMemcpyAsync(s, pos, (size_t)n_pos * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    uint64_t blocks = (d_reads->n_reads + waves - 1) / waves;
    const uint64_t cap = (uint64_t)ctx->n_cu * 8;
    if (blocks > cap) blocks = cap;
    QvPlanes q;
    q.plane[0] = d_reads->qual; q.plane[1] = d_reads->delqv; q.plane[2] = d_reads->subqv; q.plane[3] = d_reads->insqv;
    q.thr[0] = thr->qual; q.thr[1] = thr->del; q.thr[2] = thr->sub; q.thr[3] = thr->ins;
    hipLaunchKernelGGL(phase_codons_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, reinterpret_cast<const uint4*>(d_reads->bop), q, d_reads->n_reads, win_begin0, W,
                       row_bytes, waves, n_pos, (const int32_t*)s, d_keys, ctx->d_err);
    MSQ_HIP(hipGetLastError());
    // the positions live in the context's scratch: the next call that grows it must not race this kernel
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return MSQ_OK;
}
