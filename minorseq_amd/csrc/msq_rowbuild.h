// msq_rowbuild.h -- one wave writes the MSARow of one read as bytes into LDS (device code shared by the simple
// pileup kernel and the phasing gather).  MSAByRow::AddRead, src/MSA.cpp:144-186.
#pragma once

#include "msq_internal.h"

struct QvPlanes {
    const uint8_t* plane[4];  // qual, del, sub, ins
    int32_t thr[4];           // -1 = unset
};

// ArrayBase::MeetQVThresholds (include/pacbio/data/internal/ArrayRead.inl:67-87)
__device__ __forceinline__ bool meets_qv(const QvPlanes& q, uint64_t base_index)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q.thr[k] >= 0 && q.plane[k]) ok = ok && (q.plane[k][base_index] >= (uint8_t)q.thr[k]);
    return ok;
}

// row[c] = SYM_* of window column c for read r (the row must be zeroed = all ' ' beforehand); every op of the
// packed format, both group layouts.  Returns deferred error bits (DEVERR_*).
__device__ __forceinline__ uint32_t msq_build_row(uint8_t* row, int lane, const int32_t* __restrict__ ref_start,
                                                  const uint64_t* __restrict__ group_off, const uint4* __restrict__ bop,
                                                  const QvPlanes& qv, uint64_t r, int32_t win_begin0, int32_t W)
{
    uint32_t my_err = 0;
    const uint64_t g0 = group_off[r], g1 = group_off[r + 1];
    int col = ref_start[r] - win_begin0;  // MSA.cpp:148
    for (uint64_t gb = g0; gb < g1; gb += 64) {
        const uint64_t g = gb + lane;
        uint4 w = make_uint4(0u, 0u, ~0u, ~0u);  // beyond the read: op 3
        if (g < g1) w = bop[g];
        const uint32_t cons = ~w.w;  // ops 0 and 1 consume a column
        const int n = __popc(cons);
        int incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        int c = col + incl - n;
        const int total = __shfl(incl, 63);
        for (int k = 0; k < 32; ++k) {
            const uint32_t op = ((w.z >> k) & 1u) | (((w.w >> k) & 1u) << 1);
            if (op >= 2) continue;  // 'I' (MSA.cpp:172-174), 'P'/'S'/padding (:175-180)
            const uint32_t code = ((w.x >> k) & 1u) | (((w.y >> k) & 1u) << 1);
            int sym;
            if (op == 0) {
                sym = meets_qv(qv, g * 32 + MSQ_QV_BYTE(k)) ? (int)(SYM_A + code) : (int)SYM_N;  // MSA.cpp:163-166
            } else if (code == 0) {
                sym = SYM_GAP;  // 'D', MSA.cpp:168-171
            } else if (code == 1) {
                sym = SYM_N;  // literal 'N' nucleotide: 'N' whether or not the QV passes
            } else {
                const bool ok = meets_qv(qv, g * 32 + MSQ_QV_BYTE(k));
                if (code == 2) {
                    if (ok) my_err |= DEVERR_BASE;  // "Unexpected base", MSA.cpp:85
                    sym = SYM_N;
                } else {
                    sym = ok ? (int)SYM_GAP : (int)SYM_N;  // literal '-' nucleotide at a match
                }
            }
            if (c < 0 || c >= W)
                my_err |= DEVERR_WINDOW;
            else
                row[c] = (uint8_t)sym;
            ++c;
        }
        col += total;
    }
    return my_err;
}
