// msq_synth_model.h -- the synthetic read model, shared verbatim by the host and device generators
// (integer arithmetic only, so both produce the same bits).  See include/msq_synth.h.
#pragma once

#include <cstdint>

#include "../../include/msq_synth.h"

#if defined(__HIPCC__)
#define SYN_HD __host__ __device__ inline
#else
#define SYN_HD inline
#endif

// 64 random bits as a pure function of (seed, read, column, stream): two rounds of the splitmix64 finaliser.
SYN_HD uint64_t syn_hash(uint64_t seed, uint64_t read, int64_t col, uint32_t stream)
{
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (read + 1);
    x ^= (uint64_t)(col + 16) * 0xBF58476D1CE4E5B9ull;
    x += (uint64_t)(stream + 1) * 0x94D049BB133111EBull;
    for (int round = 0; round < 2; ++round) {
        x ^= x >> 30;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27;
        x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
    }
    return x;
}

// Sum of four uniform bytes (Irwin-Hall, sd 147.8) scaled to the wanted mean / sd, clamped to 0..93.
// k = sd * 1024 / 147.8  (sd 10 -> 69, sd 15 -> 104).
SYN_HD uint8_t syn_qv(uint32_t bits, int mean, int k)
{
    const int s = (int)(bits & 255u) + (int)((bits >> 8) & 255u) + (int)((bits >> 16) & 255u) + (int)(bits >> 24);
    int q = mean + (((s - 510) * k + 512) >> 10);
    if (q < 0) q = 0;
    if (q > 93) q = 93;
    return (uint8_t)q;
}

struct SynRead {
    int clone;    // 0 major, m >= 1 minor m
    int c_begin;  // first window column the read covers
    int c_end;    // one past the last
};

SYN_HD SynRead syn_read_header(const msq_synth_params& p, uint64_t read)
{
    SynRead r;
    const uint64_t hm = syn_hash(p.seed, read, -1, 0);
    const uint32_t u = (uint32_t)hm;
    r.clone = 0;
    if (p.n_minors > 0 && p.thr_minor > 0) {
        const uint32_t m = u / p.thr_minor;
        if (m < (uint32_t)p.n_minors) r.clone = (int)m + 1;
    }
    r.c_begin = 0;
    r.c_end = p.W;
    if (p.max_jitter > 0) {
        const uint64_t hj = syn_hash(p.seed, read, -2, 0);
        if ((uint32_t)hj < p.thr_jitter) {
            const uint32_t hi = (uint32_t)(hj >> 32);
            int a = (int)((hi & 0xffffu) % (uint32_t)(p.max_jitter + 1));
            int b = (int)((hi >> 16) % (uint32_t)(p.max_jitter + 1));
            if (a + b >= p.W) { a = 0; b = 0; }
            r.c_begin = a;
            r.c_end = p.W - b;
        }
    }
    return r;
}

struct SynColumn {
    int has_ins;   // one inserted base before this column
    int ins_code;
    int kind;      // 0 match '=', 1 substitution 'X', 2 deletion 'D'
    int code;      // read nucleotide code (match / substitution)
};

SYN_HD SynColumn syn_column(const msq_synth_params& p, const uint8_t* clone_codes, const SynRead& hdr, uint64_t read, int c)
{
    SynColumn o;
    const uint64_t h2 = syn_hash(p.seed, read, c, 2);
    o.has_ins = (c > hdr.c_begin) && ((uint32_t)h2 < p.thr_ins);
    o.ins_code = (int)((h2 >> 32) & 3u);
    const uint64_t h1 = syn_hash(p.seed, read, c, 1);
    const int truth = clone_codes[(int64_t)hdr.clone * p.W + c] & 3;
    if ((uint32_t)h1 < p.thr_del) {
        o.kind = 2;
        o.code = 0;
    } else if ((uint32_t)(h1 >> 32) < p.thr_sub) {
        o.kind = 1;
        o.code = (truth + 1 + (int)((h1 >> 20) % 3u)) & 3;
    } else {
        o.kind = 0;
        o.code = truth;
    }
    return o;
}

// Only whether a column carries an insertion (the length pass needs nothing else).
SYN_HD int syn_column_has_ins(const msq_synth_params& p, const SynRead& hdr, uint64_t read, int c)
{
    return (c > hdr.c_begin) && ((uint32_t)syn_hash(p.seed, read, c, 2) < p.thr_ins);
}

struct SynQv {
    uint8_t qual, del, sub, ins;
};
// QVs of the base at column c (is_insert: of the base inserted before column c).  Deleted bases carry QV 0.
SYN_HD SynQv syn_qvs(const msq_synth_params& p, uint64_t read, int c, bool is_insert)
{
    SynQv q;
    const uint64_t a = syn_hash(p.seed, read, c, is_insert ? 5 : 3);
    q.qual = syn_qv((uint32_t)a, 80, 69);
    q.ins = syn_qv((uint32_t)(a >> 32), 60, 104);
    const uint64_t b = syn_hash(p.seed, read, c, is_insert ? 6 : 4);
    q.del = syn_qv((uint32_t)b, 60, 104);
    q.sub = syn_qv((uint32_t)(b >> 32), 60, 104);
    return q;
}
