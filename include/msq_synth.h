/* msq_synth.h -- deterministic synthetic CCS amplicon reads (measurement and test input).
 *
 * Not part of the reference's interface: the reference ships no test BAMs (its cram data submodule is
 * empty), so every workload of BASELINE.json is realised by this generator.  Mixture semantics follow
 * scripts/minorvariant/mixdata:8-9 (each minor clone at a fixed fraction, the rest is the major clone);
 * per-column error rates default to the reference's own priors (src/ErrorEstimates.cpp:46-48).
 *
 * Every random decision is a pure function of (seed, global read index, column, stream), so any shard of
 * the read set regenerates identically on any rank, and the host and device generators agree bit for bit.
 */
#ifndef MSQ_SYNTH_H
#define MSQ_SYNTH_H

#include "msq_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msq_synth_params {
    uint64_t seed;
    int32_t win_begin0;     /* 0-based reference position of window column 0 */
    int32_t W;              /* window width in reference columns */
    uint32_t thr_sub;       /* per-column substitution probability * 2^32 */
    uint32_t thr_del;       /* per-column deletion probability * 2^32 */
    uint32_t thr_ins;       /* probability * 2^32 of one inserted base before a column */
    uint32_t thr_minor;     /* fraction of reads per minor clone * 2^32 */
    int32_t n_minors;       /* number of minor clones (clone sequences 1..n_minors of clone_codes) */
    uint32_t thr_jitter;    /* probability * 2^32 that a read is shortened at both ends */
    int32_t max_jitter;     /* max columns trimmed per end */
    int32_t rich_qv;        /* 1: also emit Del/Sub/Ins QV planes */
} msq_synth_params;

/* clone_codes: [(1+n_minors)][W] nucleotide codes 0..3 (A C G T): row 0 the major clone, row m minor m. */

/* Host generator, unpacked (flat ArrayBase columns as msq_pack_reads takes them).  Two calls: lengths, fill. */
int msq_synth_host_lengths(const msq_synth_params* p, uint64_t first_read, uint64_t n_reads, int64_t* n_bases);
int msq_synth_host_fill(const msq_synth_params* p, const uint8_t* clone_codes, uint64_t first_read, uint64_t n_reads,
                        const int64_t* base_off, int32_t* ref_start, int32_t* ref_end, char* cigar, char* nuc,
                        uint8_t* qual, uint8_t* subqv, uint8_t* delqv, uint8_t* insqv);

/* Device generator, packed.  d_clone_codes on the device.  Step 1 writes d_n_groups[n_reads] (groups per
 * read); the caller turns it into d_group_off[n_reads+1] (exclusive prefix sum); step 2 fills d_reads' arrays
 * (all device pointers, sized from the prefix sum; QV planes required: qual always, the others iff rich_qv). */
int msq_synth_dev_group_counts(msq_ctx* ctx, const msq_synth_params* p, uint64_t first_read, uint64_t n_reads,
                               uint32_t* d_n_groups);
int msq_synth_dev_fill(msq_ctx* ctx, const msq_synth_params* p, const uint8_t* d_clone_codes, uint64_t first_read,
                       const msq_packed_reads* d_reads);

#ifdef __cplusplus
}
#endif
#endif
