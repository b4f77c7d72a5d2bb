// msq_insertions.hip -- the per-column insertion histogram's device half: which insertion string each
// read records at which window column.
//
// Reference: MSAByRow::AddRead collects consecutive 'I' bases and stores them at the column of the next op
// (`row.Insertions[pos] = insertion` at every '=', 'X', 'D', 'P', 'S'; src/MSA.cpp:151-156,158-184); a later run
// flushed at the same pos overwrites an earlier one (std::map assignment), a run at the very end of the read is
// never flushed.  MSAByColumn then does counts[pos].IncInsertion(string) per row (src/MSA.cpp:88-90,
// MSA.inl:101).  Inserted bases are not QV-filtered.
//
// Needs the PLAIN group layout (the genomic order of inserted and column-consuming bases inside a group) with
// marked padding (MSQ_READS_PAD_MARKED: op 3 + nucleotide code 3), so that the end of a read is not mistaken for a
// 'P' / 'S' op.  One wave per read; lane i owns group i of a 64-group block: bit tricks on the four planes find the
// run ends, a wave prefix sum of the consuming-base counts gives their columns.
#include "msq_internal.h"

namespace {
struct Masks {
    uint32_t isI, cons, stop, real;  // inserted base; column-consuming; a base that flushes a run; any base
};
__device__ inline Masks masks_of(const uint4 p)  // {nucleotide bit0, bit1, op bit0, bit1}
{
    const uint32_t pad = p.x & p.y & p.z & p.w;  // op 3 + code 3
    Masks m;
    m.isI = p.w & ~p.z;
    m.cons = ~p.w;
    m.real = ~pad;
    m.stop = m.cons | (p.w & p.z & ~pad);
    return m;
}
__device__ inline uint32_t from_bit(int b) { return b >= 32 ? 0u : (~0u << b); }

// Is the run that ends at bit j of group g overwritten by a later run flushed at the same column?  Called only when
// the base after the run is a 'P' / 'S' (rare): scan forward over P/S ops; a consuming base or the end of the read
// means no; an inserted base starts a run B at the same column, which overwrites iff B itself is flushed, i.e. is
// followed by a real base.
__device__ bool overwritten(const uint4* __restrict__ bop, uint64_t g, int j, uint64_t g_end)
{
    uint64_t gg = g;
    int bit = j + 1;
    for (;;) {
        if (bit >= 32) {
            ++gg;
            bit = 0;
        }
        if (gg >= g_end) return false;
        const Masks m = masks_of(bop[gg]);
        const uint32_t sig = (m.cons | m.isI | ~m.real) & from_bit(bit);
        if (!sig) {
            bit = 32;
            continue;
        }
        const int t = __ffs(sig) - 1;
        if (!((m.real >> t) & 1u) || ((m.cons >> t) & 1u)) return false;
        // run B starts at (gg, t): find the first base after it
        uint64_t g2 = gg;
        int b2 = t;
        for (;;) {
            const Masks m2 = masks_of(bop[g2]);
            const uint32_t non_i = ~m2.isI & from_bit(b2);
            if (non_i) return (m2.real >> (__ffs(non_i) - 1)) & 1u;
            ++g2;
            b2 = 0;
            if (g2 >= g_end) return false;
        }
    }
}

// mode 0: counts[col] += 1 per recorded insertion; mode 1: append a record (columns selected by `filter`, if given)
__global__ __launch_bounds__(256) void insertion_scan_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                             const uint4* __restrict__ bop, uint64_t n_reads, int32_t win_begin0, int32_t W,
                                                             int mode, uint32_t* __restrict__ counts, const uint32_t* __restrict__ filter,
                                                             msq_insertion* __restrict__ out, uint64_t capacity,
                                                             unsigned long long* __restrict__ n_out /* [0] records */)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t g_begin = group_off[r], g_end = group_off[r + 1];
        int32_t carry = ref_start[r] - win_begin0;  // window column of the next consuming base
        for (uint64_t blk = g_begin; blk < g_end; blk += 64) {
            const uint64_t g = blk + (uint64_t)lane;
            const bool valid = g < g_end;
            const uint4 p = valid ? bop[g] : make_uint4(~0u, ~0u, ~0u, ~0u);
            const Masks m = masks_of(p);
            uint32_t next_stop0 = __shfl_down(m.stop & 1u, 1);
            if (lane == 63) next_stop0 = (g + 1 < g_end) ? (masks_of(bop[g + 1]).stop & 1u) : 0u;
            const int ncons = __popc(m.cons);
            int incl = ncons;
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const int32_t col0 = carry + incl - ncons;
            uint32_t ends = m.isI & ((m.stop >> 1) | (next_stop0 << 31));  // an inserted base followed by a flushing op
            while (ends) {
                const int j = __ffs(ends) - 1;
                ends &= ends - 1;
                const int32_t col = col0 + __popc(m.cons & ((1u << j) - 1u));
                if (col < 0 || col >= W) continue;  // pos == row length: the reference writes out of bounds (UB); dropped
                const bool next_cons = j < 31 ? ((m.cons >> (j + 1)) & 1u) : (masks_of(bop[g + 1]).cons & 1u);
                if (!next_cons && overwritten(bop, g, j, g_end)) continue;
                if (mode == 0) {
                    atomicAdd(&counts[col], 1u);
                    continue;
                }
                if (filter && !((filter[col >> 5] >> (col & 31)) & 1u)) continue;
                // the run's first base: walk back over inserted bases
                uint64_t gs = g;
                int bs;
                {
                    int b = j;
                    Masks mm = m;
                    for (;;) {
                        const uint32_t below = b >= 31 ? ~0u : ((2u << b) - 1u);
                        const uint32_t z = ~mm.isI & below;
                        if (z) {
                            bs = 32 - __clz(z);
                            break;
                        }
                        if (gs == g_begin || !(masks_of(bop[gs - 1]).isI >> 31)) {
                            bs = 0;
                            break;
                        }
                        --gs;
                        b = 31;
                        mm = masks_of(bop[gs]);
                    }
                }
                uint64_t lo = 0, hi = 0;
                uint32_t len = 0;
                for (uint64_t gg = gs; gg <= g; ++gg) {
                    const uint4 q = gg == g ? p : bop[gg];
                    const int b_first = gg == gs ? bs : 0, b_last = gg == g ? j : 31;
                    for (int b = b_first; b <= b_last; ++b, ++len) {
                        const uint64_t code = ((q.x >> b) & 1u) | (((q.y >> b) & 1u) << 1);
                        if (len < 32)
                            lo |= code << (2 * len);
                        else if (len < 64)
                            hi |= code << (2 * (len - 32));
                    }
                }
                const unsigned long long at = atomicAdd(&n_out[0], 1ull);
                if (at < capacity) {
                    msq_insertion rec;
                    rec.col = col;
                    rec.len = len;  // beyond 64 bases the caller rebuilds the string from the read (read, first_base)
                    rec.seq_lo = lo;
                    rec.seq_hi = hi;
                    rec.read = (uint32_t)r;
                    rec.first_base = (uint32_t)((gs - g_begin) * 32 + (uint64_t)bs);  // plain layout: base k of a group in slot k
                    out[at] = rec;
                }
            }
            carry += __shfl(incl, 63);
        }
    }
}

int check_layout(const msq_packed_reads* R, const char* who)
{
    if ((R->flags & MSQ_READS_NORMALIZED) || !(R->flags & MSQ_READS_PAD_MARKED))
        return msq_fail(MSQ_ERR_ARG, "%s: needs the plain group layout with marked padding (pack with MSQ_PACK_PLAIN)", who);
    if (R->n_reads && (!R->ref_start || !R->group_off || !R->bop)) return msq_fail(MSQ_ERR_ARG, "%s: NULL read arrays", who);
    return MSQ_OK;
}

unsigned grid_for(const msq_ctx* ctx, uint64_t n_reads)
{
    const uint64_t blocks = (n_reads + 3) / 4;  // 4 waves per block, one read per wave and iteration
    const uint64_t most = (uint64_t)ctx->n_cu * 16;
    return (unsigned)(blocks < 1 ? 1 : blocks > most ? most : blocks);
}
}  // namespace

extern "C" int msq_insertion_counts(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W, uint32_t* d_counts)
{
    if (!ctx || !d_reads || (W > 0 && !d_counts) || W < 0) return msq_fail(MSQ_ERR_ARG, "msq_insertion_counts: bad argument");
    int rc = check_layout(d_reads, "msq_insertion_counts");
    if (rc != MSQ_OK) return rc;
    if (d_reads->n_reads == 0 || W == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(insertion_scan_kernel, dim3(grid_for(ctx, d_reads->n_reads)), dim3(256), 0, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, (const uint4*)d_reads->bop, d_reads->n_reads, win_begin0, W, 0, d_counts,
                       (const uint32_t*)nullptr, (msq_insertion*)nullptr, (uint64_t)0, (unsigned long long*)nullptr);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}

extern "C" int msq_insertion_records(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W,
                                     const uint32_t* d_col_filter, msq_insertion* d_out, uint64_t capacity, uint64_t* n_out)
{
    if (!ctx || !d_reads || !n_out || (capacity && !d_out) || W < 0) return msq_fail(MSQ_ERR_ARG, "msq_insertion_records: bad argument");
    *n_out = 0;
    int rc = check_layout(d_reads, "msq_insertion_records");
    if (rc != MSQ_OK) return rc;
    if (d_reads->n_reads == 0 || W == 0) return MSQ_OK;
    MSQ_HIP(hipSetDevice(ctx->device));
    void* scratch = nullptr;
    rc = msq_scratch(ctx, 64, &scratch);
    if (rc != MSQ_OK) return rc;
    MSQ_HIP(hipMemsetAsync(scratch, 0, 16, ctx->stream));
    hipLaunchKernelGGL(insertion_scan_kernel, dim3(grid_for(ctx, d_reads->n_reads)), dim3(256), 0, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, (const uint4*)d_reads->bop, d_reads->n_reads, win_begin0, W, 1, (uint32_t*)nullptr, d_col_filter,
                       d_out, capacity, (unsigned long long*)scratch);
    MSQ_HIP(hipGetLastError());
    unsigned long long res[2] = {0, 0};
    MSQ_HIP(hipMemcpyAsync(res, scratch, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    *n_out = res[0];
    if (res[0] > capacity) return msq_fail(MSQ_ERR_RANGE, "msq_insertion_records: %llu records, capacity %llu", res[0], (unsigned long long)capacity);
    return MSQ_OK;
}

// ------------------------------------------------------------------------------------------ group column index
// gcol[g] = (reference column of group g's first column-consuming base + 1) | (read index & 7) << 29: what phase 1 of the
// pileup otherwise works out per pass with a wave prefix sum and a walk over the read borders.  One wave per read,
// lane i = group i of a 64-group block (any layout: only the op planes are looked at).
namespace {
__global__ __launch_bounds__(256) void index_groups_kernel(const int32_t* __restrict__ ref_start, const uint64_t* __restrict__ group_off,
                                                           const uint4* __restrict__ bop, uint64_t n_reads, uint32_t* __restrict__ gcol)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t g_begin = group_off[r], g_end = group_off[r + 1];
        int64_t carry = ref_start[r];
        for (uint64_t blk = g_begin; blk < g_end; blk += 64) {
            const uint64_t g = blk + (uint64_t)lane;
            const int ncons = g < g_end ? __popc(~bop[g].w) : 0;
            int incl = ncons;
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            // column + 1: a valid word is never 0 (what a load past the end of a batch returns)
            // columns the 29 bits cannot hold -- beyond 2^29 - 2 or NEGATIVE (a read with ReferenceStart -1 must not
            // yield the word 0, "lane past the batch") -- saturate: such a group lies outside every window the kernels
            // take an index for (stream_launch: begin0 >= 0), and is reported there like any base outside the window
            if (g < g_end) {
                const int64_t c = carry + (incl - ncons);
                gcol[g] = (c < 0 || c + 1 > 0x1fffffff ? 0x1fffffffu : (uint32_t)(c + 1)) | ((uint32_t)(r & 7u) << 29);
            }
            carry += __shfl(incl, 63);
        }
    }
}
}  // namespace

extern "C" int msq_reads_index(msq_ctx* ctx, const msq_packed_reads* d_reads, uint32_t* d_gcol)
{
    if (!ctx || !d_reads || (d_reads->n_groups && !d_gcol)) return msq_fail(MSQ_ERR_ARG, "msq_reads_index: NULL argument");
    if (d_reads->n_reads == 0) return MSQ_OK;
    if (!d_reads->ref_start || !d_reads->group_off || !d_reads->bop) return msq_fail(MSQ_ERR_ARG, "msq_reads_index: NULL read arrays");
    MSQ_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(index_groups_kernel, dim3(grid_for(ctx, d_reads->n_reads)), dim3(256), 0, ctx->stream, d_reads->ref_start,
                       d_reads->group_off, (const uint4*)d_reads->bop, d_reads->n_reads, d_gcol);
    MSQ_HIP(hipGetLastError());
    return MSQ_OK;
}
int msq_packed_dev_index(msq_ctx* ctx, msq_packed_dev* d)
{
    if (d->view.n_groups == 0) return MSQ_OK;
    hipError_t e = hipMalloc(&d->ptrs[8], d->view.n_groups * sizeof(uint32_t));
    if (e != hipSuccess) return msq_fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "group column index: %s", hipGetErrorString(e));
    int rc = msq_reads_index(ctx, &d->view, (uint32_t*)d->ptrs[8]);
    if (rc != MSQ_OK) return rc;  // the owner frees ptrs[8] with the set
    d->view.gcol = (const uint32_t*)d->ptrs[8];
    return MSQ_OK;
}
