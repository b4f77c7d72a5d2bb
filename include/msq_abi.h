/* msq_abi.h -- C ABI of the MI355X (gfx950) pileup + Fisher layer behind juliet's MSA / AminoAcidCaller
 * interface.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Every entry point returns an int status (0 = MSQ_OK, <0 = error) and never throws; msq_last_error()
 * gives the text for the calling thread.  One msq_ctx per (process, GPU); calls on one context are
 * serialised by the caller.  Unless a function says "host", pointer arguments named d_* are DEVICE
 * pointers (hipMalloc'd by the caller, e.g. a torch tensor's data_ptr()) and the work is enqueued on the
 * context's stream (msq_ctx_set_stream) without synchronising.
 *
 * All reference citations are file:line under the reference tree (PacBio MinorSeq 1.10.0).
 */
#ifndef MSQ_ABI_H
#define MSQ_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum msq_status {
    MSQ_OK = 0,
    MSQ_ERR_ARG = -1,      /* bad argument (NULL, negative size, misaligned offset ...) */
    MSQ_ERR_HIP = -2,      /* a HIP runtime call failed */
    MSQ_ERR_CIGAR = -3,    /* std::runtime_error("Unexpected cigar ...")  src/MSA.cpp:182 */
    MSQ_ERR_BASE = -4,     /* std::runtime_error("Unexpected base ...")   src/MSA.cpp:85  */
    MSQ_ERR_WINDOW = -5,   /* a read reaches outside [win_begin0, win_begin0+W): assert(pos >= 0) src/MSA.cpp:149 */
    MSQ_ERR_RANGE = -6,    /* std::out_of_range: referenceSequence.substr(absPos,3) src/AminoAcidCaller.cpp:446,
                              or a factorial argument beyond the reserved log-factorial table */
    MSQ_ERR_NODEVICE = -7, /* no usable GPU: the product path has no CPU fallback */
    MSQ_ERR_NOMEM = -8,
    MSQ_ERR_DATA = -9      /* the BAM front on the device: a BGZF block that does not inflate to its ISIZE, a record
                              chain or record the host front (cpp/msq_bam.hpp) would refuse */
};

typedef struct msq_ctx msq_ctx;

const char* msq_last_error(void);

/* ------------------------------------------------------------------------------------------------ context */
int msq_ctx_create(int device, msq_ctx** out);
int msq_ctx_destroy(msq_ctx* ctx);
/* hip_stream is a hipStream_t (NULL = the null stream).  The context does not own it. */
int msq_ctx_set_stream(msq_ctx* ctx, void* hip_stream);
int msq_ctx_sync(msq_ctx* ctx);
/* Small device-memory helpers so a C++ host that does not include HIP headers can drive the layer. */
int msq_dev_malloc(msq_ctx* ctx, size_t bytes, void** d_out);
int msq_dev_free(msq_ctx* ctx, void* d_ptr);
int msq_dev_memset(msq_ctx* ctx, void* d_ptr, int value, size_t bytes);           /* async on the stream */
int msq_memcpy_h2d(msq_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   /* async on the stream */
int msq_memcpy_d2h(msq_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);   /* synchronises */
int msq_memcpy_d2d(msq_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);   /* async on the stream */

/* ------------------------------------------------------------------------------------- packed read format
 * Replaces the in-memory type of the path, std::vector<ArrayBase> per ArrayRead
 * (include/pacbio/data/ArrayRead.h:60-81,96-119; 48 B per base) by 4 bits + 1 QV byte per base
 * (+3 QV bytes with rich QVs).
 *
 * A read is a run of 32-base GROUPS; reads start on a group boundary.  Group g holds four 32-bit planes
 * bop[4g..4g+3] = {nucleotide bit0, nucleotide bit1, op bit0, op bit1}; bit k of each plane belongs to
 * slot k of the group.  Group i of a read holds bases 32i .. 32i+31 of the read (genomic order, inserted bases
 * included); inside the group either base j sits in slot j (plain layout), or -- MSQ_READS_NORMALIZED, what
 * msq_pack_reads and the synthetic generator write -- the column-consuming bases (ops 0, 1) fill the slots from
 * 0 upwards in genomic order and the others (ops 2, 3) fill them from 31 downwards in genomic order, padding
 * in between.  Nothing on the path looks at where inside its group an inserted or padded base sits
 * (MSA.cpp:172-180), so the counts are the same; the pileup kernel then needs no bit compaction per group.
 *     nucleotide code  A=0 C=1 G=2 T=3                  (NucleotideConversion.h:69-88 order)
 *     op 0  '=' / 'X'   column-consuming, nucleotide subject to the QV mask   (MSA.cpp:160-167)
 *     op 1  column-consuming symbol that is not a QV-checked ACGT; the nucleotide code selects it:
 *           0 = 'D' -> '-' (MSA.cpp:168-171), 1 = a match whose nucleotide is a literal 'N',
 *           2 = a match whose nucleotide is none of ACGTN- (raises MSQ_ERR_BASE unless QV-masked to 'N'),
 *           3 = a match whose nucleotide is a literal '-'
 *     op 2  'I' inserted base, consumes no column (MSA.cpp:172-174)
 *     op 3  'P' / 'S' / padding after the last base of the read: consumes nothing (MSA.cpp:175-180)
 * QV planes hold one byte per base, 32 per group; a NULL plane means "QV absent for every base"
 * (boost::none in ArrayBase, ArrayRead.h:111-114), which always passes (ArrayRead.inl:72-87).
 * Inside a group the 32 QV bytes are stored transposed as a 4x8 matrix: base k of the group lives at byte
 * MSQ_QV_BYTE(k) = 4*(k & 7) + (k >> 3).  (Dword i of the group then holds bases i, 8+i, 16+i, 24+i, so a
 * 4-byte SIMD-within-a-register compare of the 8 dwords yields the 32-bit pass mask in base order.)
 */
#define MSQ_QV_BYTE(k) ((((k) & 7) << 2) | ((k) >> 3))

typedef struct msq_packed_reads {
    uint64_t n_reads;
    uint64_t n_groups;          /* total groups == group_off[n_reads] */
    const int32_t* ref_start;   /* [n_reads] ArrayRead::ReferenceStart(), 0-based */
    const int32_t* ref_end;     /* [n_reads] ArrayRead::ReferenceEnd(), exclusive */
    const uint64_t* group_off;  /* [n_reads+1] first group of each read */
    const uint32_t* bop;        /* [n_groups*4] */
    const uint8_t* qual;        /* [n_groups*32] or NULL */
    const uint8_t* delqv;       /* [n_groups*32] or NULL */
    const uint8_t* subqv;       /* [n_groups*32] or NULL */
    const uint8_t* insqv;       /* [n_groups*32] or NULL */
    uint32_t flags;             /* MSQ_READS_* */
    int32_t align_origin;       /* MSQ_READS_ALIGNED: the reference column the groups are aligned to (else 0) */
    const uint32_t* gcol;       /* [n_groups] or NULL: the group column index (msq_reads_index) */
    const uint32_t* guide;      /* [guide_words][8] or NULL: the set's guide / alt haplotype (msq_reads_align) */
    uint32_t guide_words;
    uint32_t hap_count;         /* 0 or 1: no haplotype groups; 2 .. MSQ_MAX_HAPS: the reads are grouped (msq_reads_align) */
    const uint32_t* hap_guides; /* [hap_count][guide_words][8] or NULL: a guide / alt haplotype per group */
    const uint32_t* hap_perm;   /* [n_reads] or NULL: the reads ordered by group (group 0 first), ascending inside a group */
    uint32_t hap_reads[8];      /* reads in each group; their sum is n_reads */
} msq_packed_reads;

#define MSQ_MAX_HAPS 8

#define MSQ_READS_HAS_EXOTIC 1u /* some base uses op 1 with nucleotide code 2 or 3 */
#define MSQ_READS_NORMALIZED 2u /* inside every group the column-consuming bases come first (see above) */
#define MSQ_READS_PAD_MARKED 4u /* plain layout whose padding slots carry op 3 + nucleotide code 3 ('P' / 'S' carry
                                   code 0): the end of a read can be told from a 'P' / 'S' op (msq_insertion_*) */
#define MSQ_READS_QV_BAKED 8u   /* made by msq_reads_bake_qv: QV-failing match bases already read 'N', no QV planes */
#define MSQ_READS_INS_EXOTIC 16u /* some inserted ('I') base is not ACGT: it was packed as 'A'; insertion strings taken from
                                   the packed planes differ from the reference's there (see msq_insertion.read / first_base) */

#define MSQ_READS_ALIGNED 32u    /* made by msq_reads_align: group k of a read holds exactly the reference columns
                                   [align_origin + 32 (a + k), + 32), a = (ReferenceStart - align_origin) >> 5, the base of
                                   column c in slot c & 31; slots without a base are padding (op 3); inserted bases are
                                   left out (nothing on the pileup / phasing path reads them, src/MSA.cpp:172-174).  Still a
                                   valid plain-layout set for every other call */

/* The group column index of a packed set in HBM: gcol[g] = (reference column of group g's first column-consuming
 * base + 1) | (index of its read & 7) << 29 -- `pos` of MSAByRow::AddRead (src/MSA.cpp:148,172-176) at the group's first
 * base, which every pass over the set otherwise recomputes with a prefix sum over the groups and a walk over the
 * read borders.  Derived data, 4 bytes per 32 bases, any group layout; reference columns < 2^29 - 1; a valid word is never 0.  A set whose view
 * carries it (`gcol` != NULL) is piled up and phased ~6 % faster; a view with gcol == NULL works as before.  Every
 * device set the library makes (msq_reads_upload, msq_pack_reads_device*, msq_reads_bake_qv, msq_reads_partition)
 * carries one; for a caller-made view: d_gcol = n_groups u32 of device memory, then view.gcol = d_gcol.  The index
 * belongs to the view it was built for: a view that starts at another read of the same arrays needs its own (the
 * row bits count reads from the view's read 0).  Async. */
int msq_reads_index(msq_ctx* ctx, const msq_packed_reads* d_reads, uint32_t* d_gcol);

/* QvThresholds (include/pacbio/data/QvThresholds.h:44-54, src/QvThresholds.cpp:45-56): -1 = boost::none. */
typedef struct msq_qv_thresholds {
    int32_t qual, del, sub, ins;
} msq_qv_thresholds;
/* QvThresholds::QvThresholds(): environment DELQV, SUBQV (default 42), INSQV, QUALQV.  Host. */
int msq_qv_thresholds_from_env(msq_qv_thresholds* out);

/* Host packer: flat ArrayBase columns -> packed host buffers owned by the returned object.
 * Replaces the per-base walk of MSAByRow::AddRead (src/MSA.cpp:144-186) over ArrayRead::Bases().
 * cigar[i] in "=XDIPS" (anything else: MSQ_ERR_CIGAR), nuc[i] the ArrayBase::Nucleotide char,
 * qual/subqv/delqv/insqv per-base QVs or NULL, base_off[n_reads+1] prefix offsets into those arrays. */
typedef struct msq_packed_host msq_packed_host;
int msq_pack_reads(uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end, const int64_t* base_off,
                   const char* cigar, const char* nuc, const uint8_t* qual, const uint8_t* subqv,
                   const uint8_t* delqv, const uint8_t* insqv, msq_packed_host** out);
const msq_packed_reads* msq_packed_host_view(const msq_packed_host* p);
int msq_packed_host_free(msq_packed_host* p);
/* Copy a host-resident packed set to freshly allocated device memory (async H2D on the stream). */
typedef struct msq_packed_dev msq_packed_dev;
int msq_reads_upload(msq_ctx* ctx, const msq_packed_reads* host_reads, msq_packed_dev** out);
const msq_packed_reads* msq_packed_dev_view(const msq_packed_dev* p);
int msq_packed_dev_free(msq_ctx* ctx, msq_packed_dev* p);
/* The QV thresholds applied once, at pack time: a new device set (the caller keeps and frees the source) in which
 * every '=' / 'X' base that misses a threshold -- ArrayBase::MeetQVThresholds false, ArrayRead.inl:67-87 -- is a
 * literal 'N', the row char MSAByRow::AddRead writes for it (src/MSA.cpp:163-166), and which carries no QV planes.
 * msq_pileup / msq_phase_codons / msq_reads_partition on the result with any thresholds give what the source gives
 * under `thr`, bit for bit, from 0.5 instead of 1.5 .. 4.5 bytes per base.  The thresholds of a juliet run are
 * fixed before the reads are loaded (JulietSettings), so the front bakes right after packing.  Asynchronous. */
int msq_reads_bake_qv(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, msq_packed_dev** out);

/* The column-aligned layout: MSAByRow::AddRead's placement (`row.Bases[pos++]`, src/MSA.cpp:148,158-171) done ONCE, when
 * the set is made, instead of in every pass over it.  In the other layouts a group holds 32 consecutive BASES of its
 * read, so where its bases land in the row depends on every insertion before it: each pass finds the columns again
 * (a prefix sum or the column index), shifts every group's planes to its bit offset and ORs them into a row in LDS.
 * Here a group IS a 32-column word of the read's row (still 2 bits of nucleotide + 2 bits of op per base, 16 bytes per
 * group), so the pileup needs no row building at all: a lane loads the word it counts.  `origin` fixes the word grid:
 * the pileup takes the fast kernel when its window begins a multiple of 32 columns from it -- pass the window's first
 * column (the reads' common ReferenceStart for an amplicon; on a sharded sample the GLOBAL window begin); origin <=
 * every ReferenceStart.  Thresholds are applied as in msq_reads_bake_qv (the result carries no QV planes).  The set
 * also carries a GUIDE: per column the majority base and -- where at least max(3, sample / 256) sampled reads agree on
 * another base -- a runner-up, learnt from up to 2048 reads spread evenly over the set (guide[w] = {valid, bit 0,
 * bit 1, 0, valid, alt bit 0, alt bit 1, has alt} for columns [origin + 32 w, + 32)).  A speed hint only: the pileup
 * counts cells that match the guide in registers, cells that match the alt haplotype in LDS and everything else with
 * global atomics; any guide gives the same counts.  Samples of several clones (a mixture of strains, minor populations
 * with many variant codons) are also GROUPED BY HAPLOTYPE: the sampled reads' bases at the columns with a runner-up are
 * clustered (on the host, <= MSQ_MAX_HAPS clusters), every read is assigned to the nearest cluster on the GPU, the
 * reads are listed group by group (hap_perm, hap_reads) and every group gets a guide of its own (hap_guides); the
 * pileup then counts every group against its own guide.  Again a speed hint only.  The new set is independent of the
 * source (the caller keeps and frees that).  Synchronises. */
int msq_reads_align(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t origin,
                    msq_packed_dev** out);

/* One ArrayBase (include/pacbio/data/ArrayRead.h:96-119) as an 8-byte record: the boost::optional QVs become a
 * presence byte + value.  has_qual covers qual; has_rich covers del, sub and ins (ArrayRead.h:98-101 sets all
 * three or none). */
typedef struct msq_array_base {
    char cigar;       /* ArrayBase::Cigar */
    char nucleotide;  /* ArrayBase::Nucleotide */
    uint8_t has_qual, has_rich;
    uint8_t qual, del, sub, ins;
} msq_array_base;
/* Device packer: MSAByRow::AddRead's per-base walk (src/MSA.cpp:144-186) over ArrayRead::Bases() on the GPU.
 * bases[r] points at the n_bases[r] records of read r (host memory, one contiguous array per read: a
 * std::vector<ArrayBase>'s data()).  The host only copies them into pinned staging (`threads` threads, <= 0 =
 * up to 16; chunks of `chunk_bases` records, <= 0 = 32 Mi) while the previous chunk crosses PCIe; translation and
 * bit-plane packing run on the device.  The result equals msq_pack_reads + msq_reads_upload of the same bases
 * with absent QVs written as 255: same planes, NULL QV planes when no base of the set carries that QV (the QV
 * bytes of slots that hold no base are unspecified).  Unknown cigar char: MSQ_ERR_CIGAR "Unexpected cigar <code>"
 * for the first one in read order (src/MSA.cpp:182).  Synchronises.
 * pack_flags: 0 = the normalized layout; MSQ_PACK_PLAIN = base k of a group in slot k and marked padding
 * (MSQ_READS_PAD_MARKED), which the insertion scan needs and the pileup accepts as well. */
#define MSQ_PACK_PLAIN 1u
int msq_pack_reads_device(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                          const msq_array_base* const* bases, const int64_t* n_bases, uint32_t pack_flags, int threads,
                          int64_t chunk_bases, msq_packed_dev** out);
/* The same packer for a producer that has no ArrayBase arrays yet (a BAM decoder: BAMArrayRead's constructor,
 * src/ArrayRead.cpp:53-99, writing straight into the staging buffer): instead of copying, the library calls
 * fill(user, r, dst) for every read r -- each exactly once, from several threads at a time, in no particular
 * order -- and the callee writes the read's n_bases[r] records to dst (pinned memory).  fill must not throw. */
typedef void (*msq_fill_read_fn)(void* user, uint64_t read_index, msq_array_base* dst);
int msq_pack_reads_device_fill(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                               const int64_t* n_bases, msq_fill_read_fn fill, void* user, uint32_t pack_flags,
                               int threads, int64_t chunk_bases, msq_packed_dev** out);

/* Both packers with the run's QV thresholds applied while packing: the result is, byte for byte, what
 * msq_reads_bake_qv(msq_pack_reads_device(...), thr) gives -- failing '=' / 'X' bases already read 'N'
 * (src/MSA.cpp:163-166), no QV planes, MSQ_READS_QV_BAKED when a threshold met a QV some base carries -- without
 * the QV planes ever being written to HBM and without the second pass over the set.  What MSAByRow(reads) of the C++
 * front calls: QvThresholds() is read from the environment before the reads are packed. */
int msq_pack_reads_device_baked(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                const msq_array_base* const* bases, const int64_t* n_bases, uint32_t pack_flags, int threads,
                                int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out);
int msq_pack_reads_device_fill_baked(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                     const int64_t* n_bases, msq_fill_read_fn fill, void* user, uint32_t pack_flags,
                                     int threads, int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out);

/* The thresholded packer for a producer that can TRANSLATE while it decodes (no counterpart in the reference: the walk
 * of MSAByRow::AddRead, src/MSA.cpp:158-184, and ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87, done by the callee
 * base by base): the callback writes ONE byte per base instead of an 8-byte record,
 *     byte = msq_compact_table()[(uint8_t)Cigar | (uint8_t)Nucleotide << 8]
 *          = nucleotide code | op << 2 | (exotic symbol) << 4 | (inserted base that is not ACGT) << 5, 0x80 = unknown cigar,
 * with msq_compact_fail() applied to the byte of a base that misses one of the run's thresholds (the caller knows them:
 * it passes them on in thr only so that the set can be flagged).  presence: OR 1 into it when some base carries a
 * QualQV, 2 when some base carries the rich QVs (that decides MSQ_READS_QV_BAKED as in the other packers).
 * Returns -1, or for the first base with an unknown cigar its index in the read << 8 | the cigar char
 * (the packer then fails with MSQ_ERR_CIGAR "Unexpected cigar <code>" for the first one in read order, src/MSA.cpp:182).
 * Same packed bytes as msq_pack_reads_device_fill_baked on the records the bytes stand for. */
typedef int64_t (*msq_fill_read_compact_fn)(void* user, uint64_t read_index, uint8_t* dst, uint32_t* presence);
int msq_pack_reads_device_fill_compact(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                                       const int64_t* n_bases, msq_fill_read_compact_fn fill, void* user, uint32_t pack_flags,
                                       int threads, int64_t chunk_bases, const msq_qv_thresholds* thr, msq_packed_dev** out);
const uint8_t* msq_compact_table(void); /* [65536], valid for the life of the process */
/* a '=' / 'X' base (op 0, or one of the exotic match symbols under op 1) that misses a threshold reads 'N' (MSA.cpp:163-166) */
uint8_t msq_compact_fail(uint8_t c); /* (c & 0x30) | 5 for those bases, c for every other */
/* the same rule for a decoder's inner loop (no call per base): op = c >> 2 & 3, code = c & 3 */
#define MSQ_COMPACT_FAIL(c) \
    ((uint8_t)((((c) & 0x0cu) == 0u || (((c) & 0x0cu) == 0x04u && ((c) & 2u))) ? (((c) & 0x30u) | 5u) : (c)))

/* ----------------------------------------------------------------------------------------------- pileup
 * MSAByRow::UpdateBoundaries over all reads (src/MSA.cpp:96-97,138-142): begin0 = min ReferenceStart,
 * end0 = max ReferenceEnd (0-based; MSAByRow::BeginPos()/EndPos() report begin0+1 / end0+1, MSA.cpp:107-108).
 * No reads: begin0 = INT32_MAX, end0 = 0.  d_reads fields are device pointers.  Synchronises. */
int msq_window_bounds(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* begin0, int32_t* end0);
/* The same bounds left in device memory, for callers that must not wait for the host (a stream of batches whose
 * window is known from the target configuration, steps captured in a HIP graph, shards merged by a collective):
 * d_bounds[0] = -begin0, d_bounds[1] = end0 -- negated so that ONE element-wise MAX all-reduce over the ranks
 * yields the global window.  No reads: {-INT32_MAX, 0}.  Enqueued on the stream, does not synchronise. */
int msq_window_bounds_device(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* d_bounds);

/* The first half of MSAByRow's constructor in one launch, for steps that run back to back on the stream:
 * msq_window_bounds_device (UpdateBoundaries over the reads, src/MSA.cpp:96-97,138-142) plus the count matrices
 * cleared (rows start blank and MSAColumn counts at 0, include/pacbio/data/MSA.h:99-101,194): d_counts[0 .. n_words)
 * = 0, typically the caller's [W][6] and [W][64] matrices in one buffer.  Async. */
int msq_pileup_begin(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t* d_bounds, uint32_t* d_counts, size_t n_words);

/* MSAByRow's constructor + MSAByColumn + CodonsAt (src/MSA.cpp:55-92,94-136,144-219) for a caller that knows its
 * window beforehand, as ONE call: what msq_pileup_begin followed by msq_pileup(.., MSQ_PILEUP_AUTO) leave --
 * d_bounds = {-begin0, end0} of these reads, d_counts = [W][6] nucleotide counts followed by [W][64] codon counts,
 * OVERWRITTEN (not added to) with the pileup inside [win_begin0, win_begin0 + W).  When the streaming kernel can
 * take the whole call as one launch (W <= 4096, no exotic ops, ~2 M reads on 256 CUs) there is no bounds and no
 * clearing launch: the bounds are a side product of the kernel and its fold stores.  Otherwise exactly the two
 * calls above.  The context keeps a [W][70] u32 matrix of its own for this (grown on demand).  Async. */
int msq_pileup_step(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                    int32_t W, int32_t* d_bounds, uint32_t* d_counts);

/* MSAByRow::MSAByRow + MSAByColumn::MSAByColumn + MSAByRow::CodonsAt for every column
 * (src/MSA.cpp:55-92, 94-136, 144-219) as two dense count matrices, ADDED into the caller's buffers:
 *   d_nuc  [W][6]  u32, bins A C G T - N            == MSAColumn::counts_ (MSA.h:194)
 *   d_codon[W][64] u32, bin 16*b0+4*b1+b2 (A0..T3)  == CodonsAt(c)[codon]; column 0 and columns >= W-2
 *                  stay zero (MSARow::CodonAt's i > 0 rule and CodingCodonAt's bound, MSA.cpp:193,207).
 * The caller zeroes the matrices (msq_dev_memset) before the first shard and may call repeatedly to
 * accumulate shards.  d_reads fields are device pointers.  Async. */
#define MSQ_PILEUP_AUTO 0     /* pick the kernel */
#define MSQ_PILEUP_SIMPLE 1   /* force the simple per-read kernel (any width, any op) */
#define MSQ_PILEUP_STREAM 2   /* force a streaming bit-plane kernel (error if not applicable) */
#define MSQ_PILEUP_ROWS 3     /* force the kernel over column-aligned sets (msq_reads_align; error if not applicable) */
int msq_pileup(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
               int32_t W, uint32_t* d_nuc, uint32_t* d_codon, int mode);
/* msq_pileup whose window comes from device memory, as msq_window_bounds_device (and a MAX all-reduce) left it:
 * window = [-d_bounds[0], d_bounds[1]).  The caller sized its count matrices (W_max columns) and its plan for a
 * window that starts at begin0: a window with another origin, or wider than W_max, raises the deferred
 * MSQ_ERR_WINDOW (so msq_call_cells never reads counts laid out for another window unnoticed); a narrower one
 * leaves the columns beyond it untouched.  No host synchronisation anywhere: bounds -> pileup -> msq_call_cells
 * run back to back on the stream.  Streaming kernel only (reads without MSQ_READS_HAS_EXOTIC); otherwise
 * MSQ_ERR_ARG.  Same reference lines as msq_pileup. */
int msq_pileup_device_window(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr,
                             const int32_t* d_bounds, int32_t begin0, int32_t W_max, uint32_t* d_nuc,
                             uint32_t* d_codon);
/* Column-slab partition, the HBM layout for windows wider than one slab of the streaming kernel (4096 columns:
 * the 9.7 kb HIV genome, 20 kb amplicons).  msq_pileup takes one launch per slab and, over the whole read set,
 * streams every read once per slab; the partition holds per slab a packed read set of its own with, of every
 * read, only the 32-base groups that reach into the slab's columns (and the two columns to its right, for the
 * slab's last codons).  Sub-reads are ordinary packed reads (group-aligned copies; ReferenceStart = the column of
 * the first copied group), so the counts are those of msq_pileup, while every launch streams only its slab's
 * share.  A pack-time step (one scan of the bit planes, one gather); d_reads stays valid and untouched.
 * max_slab_columns: the widest slab wanted, a multiple of 32 up to 4096 (<= 0: 4096, the fastest measured:
 * every sub-read costs the kernel a read's worth of bookkeeping, so fewer, wider slabs win).
 * Synchronises.  Same reference lines as msq_pileup. */
typedef struct msq_partition msq_partition;
int msq_reads_partition(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W,
                        int32_t max_slab_columns, msq_partition** out);
int msq_partition_slabs(const msq_partition* p);
const msq_packed_reads* msq_partition_view(const msq_partition* p, int slab); /* device pointers inside */
int msq_partition_free(msq_ctx* ctx, msq_partition* p);
/* msq_pileup over the partition's window: slab after slab, each over its own read set.  Async. */
int msq_pileup_partition(msq_ctx* ctx, const msq_partition* p, const msq_qv_thresholds* thr, uint32_t* d_nuc,
                         uint32_t* d_codon);
/* msq_pileup for callers that must not allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same result as msq_pileup(..., MSQ_PILEUP_STREAM), added into
 * the caller's matrices, but the one buffer the pileup needs beyond its inputs and outputs -- the per-workgroup
 * partial counts -- lives in a device array built into the library (32 MiB per GPU, held until the process ends)
 * instead of the context's scratch, and the call makes no hipMalloc / hipFree.  Limits: W <= 4096 columns (one
 * column slab), GPUs of up to 256 CUs, reads without MSQ_READS_HAS_EXOTIC; anything beyond is refused with
 * MSQ_ERR_ARG.  All calls on one GPU share that array: calls that overlap on one GPU (two contexts or two streams)
 * race -- serialise them.  Nothing in it has to be cleared between calls (every launch overwrites the partials it
 * then folds).  Async on the context's stream; deferred errors through msq_pileup_check as for msq_pileup. */
int msq_pileup_noalloc(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                       int32_t W, uint32_t* d_nuc, uint32_t* d_codon);
/* Deferred device-side error of the pileups enqueued since the last check (MSQ_ERR_BASE / MSQ_ERR_WINDOW).
 * Synchronises. */
int msq_pileup_check(msq_ctx* ctx);

/* The per-read gather of AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:149-160): for every read r of
 * d_reads and every k < n_pos, MSARow::CodonAt(pos[k]) (src/MSA.cpp:188-195) of the read's row in the window
 * [win_begin0, win_begin0 + W) under the QV thresholds, as d_keys[r * n_pos + k] = s0 | s1 << 3 | s2 << 6 with
 * s_j the row symbol at index pos[k] + j: 0 ' ' (read does not cover the column), 1..4 A C G T, 5 '-', 6 'N',
 * and 7 for an index CodonAt leaves out (i <= 0 or i >= W: the string is shorter there).  Two reads have the same
 * codon strings at every position exactly when their key rows are equal.  pos is a host array; d_keys is device
 * memory, [n_reads * n_pos].  Synchronises.  (Sets with the group column index and no QV plane to apply take a
 * per-(read, position) lookup when the positions are few against the read length; same keys.) */
int msq_phase_codons(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                     int32_t W, int32_t n_pos, const int32_t* pos, uint16_t* d_keys);

/* The collapse of reads with identical codon tuples in AminoAcidCaller::PhaseVariants
 * (src/AminoAcidCaller.cpp:162-196: every read is compared with the observations so far and joins the first equal
 * one, or opens a new one) on the key rows msq_phase_codons left in device memory: a hash + sort + segment pass.
 *   d_cluster_of [n_reads]   observation index of every read; observations are numbered in the order of their first
 *                            read, exactly the order the reference opens them in
 *   d_first_read [n_reads]   first read of observation j (j < *n_clusters)
 *   d_count      [n_reads]   reads in observation j
 *   d_rep_keys   [rep_capacity * n_pos] the key row of observation j (= of its first read); may be NULL
 * The names of an observation, in the reference's order, are those of the reads r = 0, 1, ... with cluster_of[r] = j.
 * Rows are compared in full, never by hash alone.  MSQ_ERR_RANGE: more observations than rep_capacity, or two
 * different rows share a 64-bit hash (then the caller clusters on the host).  Synchronises. */
int msq_phase_cluster(msq_ctx* ctx, const uint16_t* d_keys, uint64_t n_reads, int32_t n_pos, uint32_t* d_cluster_of,
                      uint32_t* d_first_read, uint32_t* d_count, uint16_t* d_rep_keys, uint64_t rep_capacity,
                      uint64_t* n_clusters);

/* ------------------------------------------------------------------------------------- insertion histogram
 * The insertion strings MSAByRow::AddRead records per row (`row.Insertions[pos] = insertion`, src/MSA.cpp:151-156:
 * consecutive 'I' bases, stored at the window column of the op that follows them; a later run flushed at the same
 * column overwrites; a run at the very end of the read is dropped) and MSAByColumn adds up per column
 * (counts[pos].IncInsertion, src/MSA.cpp:88-90, MSA.inl:101) -- what `fuse` reads (src/tools/Fuse.cpp:99-119).
 * d_reads must be in the plain layout with marked padding (MSQ_PACK_PLAIN).  Inserted bases are not QV-filtered.
 * A run flushed at column W (the reference then writes out of bounds) is dropped.
 * A record carries the first 64 bases of the string at 2 bits each and, for everything the 2-bit form cannot hold,
 * where the run sits in its read: a run longer than 64 bases, or -- in a set flagged MSQ_READS_INS_EXOTIC -- one with
 * a non-ACGT base (packed as 'A'), is rebuilt by the caller from the read's own ArrayBases
 * [first_base, first_base + len): the reference stores strings of any length with their literal chars
 * (src/MSA.cpp:151-156,172-174).  The C++ front does exactly that. */
typedef struct msq_insertion {
    int32_t col;         /* window column (0-based) */
    uint32_t len;        /* inserted bases (any length) */
    uint64_t seq_lo;     /* 2 bits per base (A=0 C=1 G=2 T=3): base k at bits 2k..2k+1 of seq_lo (k < 32) */
    uint64_t seq_hi;     /*                                     ... or bits 2(k-32).. of seq_hi (k < 64) */
    uint32_t read;       /* index of the read in the set */
    uint32_t first_base; /* index of the run's first base among the read's ArrayBases */
} msq_insertion;
/* d_counts[col] += number of rows that record an insertion at col (any string).  Async on the stream. */
int msq_insertion_counts(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W,
                         uint32_t* d_counts);
/* One record per (row, column) insertion, in no particular order, for the columns whose bit is set in
 * d_col_filter ([ceil(W/32)] words; NULL = all columns).  *n_out = records found; more than `capacity`:
 * MSQ_ERR_RANGE (d_out then holds the first `capacity`).  Synchronises. */
int msq_insertion_records(msq_ctx* ctx, const msq_packed_reads* d_reads, int32_t win_begin0, int32_t W,
                          const uint32_t* d_col_filter, msq_insertion* d_out, uint64_t capacity, uint64_t* n_out);

/* Measurement aid (no counterpart in the reference): while enabled, every msq_pileup records a HIP event pair
 * on the context's stream around its dominant kernel launch (the one that streams the packed reads), up to
 * max_records pairs; msq_timing_read synchronises, returns the elapsed ms of each pair and resets.
 * max_records == 0 disables.  bench.py uses this for the roofline figure. */
int msq_timing_enable(msq_ctx* ctx, int32_t max_records);
int msq_timing_read(msq_ctx* ctx, float* ms_out, int32_t capacity, int32_t* n_out);
/* The same with only every stride-th pileup timed (an event pair costs the stream two ~5 us bubbles, which is
 * visible in a 0.1 ms step), and, for the timed pileups, a third event behind the msq_call_cells that follows:
 * msq_timing_read_path returns begin-of-pileup-kernel .. end-of-testing-pass (K1 + fold [+ the count all-reduce]
 * + K2, what SURVEY 8d calls t_kernel(K1 + K2)).  Call it before msq_timing_read, which resets the records. */
int msq_timing_enable_strided(msq_ctx* ctx, int32_t max_records, int32_t stride);
int msq_timing_read_path(msq_ctx* ctx, float* ms_out, int32_t capacity, int32_t* n_out);

/* ---------------------------------------------------------------------------------- read shards over the GPUs
 * No counterpart in the reference (a single process): the exchange step the north star prescribes when the reads
 * of one sample are sharded over the GPUs of a node, one process (and one context) per GPU -- RCCL collectives
 * over xGMI on the context's stream.  Counts are additive integers and the window is a min / max, so every rank
 * ends up with bit-identical matrices and the testing pass (msq_call_cells) runs replicated, with no broadcast.
 * Rehearsal transport: with MSQ_COMM_TRANSPORT=shm in the environment (when the id is made and the communicator
 * created) the same entry points exchange through a POSIX shared-memory segment between processes of one host that may
 * share ONE GPU -- which RCCL refuses -- host-synchronously.  It exists so that a one-GPU box can run world > 1 (the
 * sharded C++ front, the bench's ranks, tests/test_gpu_comm.py); it is no data path and nothing else selects it.
 *   rank 0:      msq_comm_unique_id(id)            -> hand the 128 bytes to every rank (MPI, a file, any store)
 *   every rank:  msq_comm_create(ctx, rank, world, id)
 *   per step:    msq_window_bounds_device -> msq_comm_allreduce_bounds -> msq_pileup_device_window (own shard)
 *                -> msq_comm_allreduce_counts -> msq_call_cells
 *   or, when the window is known beforehand (a target region):
 *                msq_pileup_begin -> msq_pileup(begin0, W) -> msq_comm_allreduce_step (ONE RCCL group: bounds MAX +
 *                counts SUM) -> msq_call_cells, and the caller compares the reduced bounds with its window.
 * A context without a communicator treats every msq_comm_allreduce_* as a no-op (one GPU: the shard is the sample).
 * RCCL (librccl.so.1) is loaded when the first msq_comm_* call needs it; the single-GPU path does not depend on it. */
#define MSQ_COMM_ID_BYTES 128
int msq_comm_available(void);  /* MSQ_OK when RCCL can be loaded here: local, NOT collective -- ranks agree on it before msq_comm_create */
int msq_comm_unique_id(uint8_t* id_out /* [MSQ_COMM_ID_BYTES] */);   /* host */
int msq_comm_create(msq_ctx* ctx, int rank, int world, const uint8_t* id);  /* collective over the ranks; synchronises */
int msq_comm_destroy(msq_ctx* ctx);
int msq_comm_active(const msq_ctx* ctx); /* 1 when the context has a communicator (even of one rank) */
int msq_comm_rank(const msq_ctx* ctx);   /* 0 without a communicator */
int msq_comm_world(const msq_ctx* ctx);  /* 1 without a communicator */
/* Contiguous read range of a rank, [rank * N / world, (rank + 1) * N / world).  Host. */
int msq_comm_shard(uint64_t n_reads, int rank, int world, uint64_t* first, uint64_t* count);
/* The same, balanced by base count (SURVEY.md 8e "balanced by base count via base_offset") for reads that do not all
 * span the window: prefix[i] = bases (or 32-base groups: group_off) of the reads before read i, n_reads + 1 entries,
 * host; rank r starts at the first read i with prefix[i] >= ceil(r * prefix[n] / world).  Host. */
int msq_comm_shard_bases(const uint64_t* prefix, uint64_t n_reads, int rank, int world, uint64_t* first, uint64_t* count);
/* Collective: every rank hands in its local return code (its text: local_text, or msq_last_error() when NULL) and gets back the code of
 * the LOWEST rank that failed, with that rank's text in msq_last_error() -- what a one-process run over the same reads
 * reports first (the "Unexpected cigar" / "Unexpected base" of src/MSA.cpp:85,182 raised in one rank's share) -- or
 * MSQ_OK.  A rank must call it before it throws, or the others wait for it in the next collective.  Synchronises.
 * Without a communicator: *agreed_rc = local_rc. */
int msq_comm_agree_status(msq_ctx* ctx, int local_rc, const char* local_text, int* agreed_rc);
/* d_bounds = {-begin0, end0} as msq_window_bounds_device left it: element-wise MAX over the ranks, in place. */
int msq_comm_allreduce_bounds(msq_ctx* ctx, int32_t* d_bounds);
/* u32 SUM over the ranks, in place: the [W][6] + [W][64] count matrices (global read count < 2^32). */
int msq_comm_allreduce_counts(msq_ctx* ctx, uint32_t* d_counts, size_t n);
/* Both of the above as one RCCL group (one launch). */
int msq_comm_allreduce_step(msq_ctx* ctx, int32_t* d_bounds, uint32_t* d_counts, size_t n);
/* Every rank's bytes_per_rank bytes, concatenated in rank order into d_recv [world * bytes_per_rank] (the phasing
 * keys of the shards, when PhaseVariants runs on a sharded sample).  Without a communicator: a copy. */
int msq_comm_allgather(msq_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);

/* ------------------------------------------------------------------------------------------ cell testing
 * A plan fixes which (gene, codon position) pairs AminoAcidCaller::CallVariants visits
 * (src/AminoAcidCaller.cpp:413-427): for each gene, i = begin; i < end-2; (i-begin)%3==0, 1-based.
 * n_genes == 0 -> one gene [begin1, end1) ("Unnamed ORF", :402-405).  ref_seq NULL/ref_len 0 -> no
 * reference (:411).  begin1 = MSAByRow::BeginPos() = win_begin0 + 1.  Host arrays. */
typedef struct msq_plan msq_plan;
int msq_plan_create(msq_ctx* ctx, int32_t n_genes, const int32_t* gene_begin, const int32_t* gene_end,
                    const char* ref_seq, int64_t ref_len, int32_t begin1, int32_t W, msq_plan** out);
int64_t msq_plan_positions(const msq_plan* plan);
int msq_plan_destroy(msq_ctx* ctx, msq_plan* plan);

/* Per visited codon position (same field meaning as the locals of CallVariants, :417-463). */
typedef struct msq_position {
    int32_t gene;       /* index into the plan's gene list */
    int32_t i;          /* 1-based reference position of the codon start */
    int32_t aa_pos;     /* 1 + relPos/3 */
    int32_t win_pos;    /* i - begin1 */
    int32_t coverage;   /* sum of the codon counts at win_pos */
    int32_t ref_codon;  /* refCodon index, -1 if unset / not a codon */
    int32_t alt_codon;  /* altRefCodon index, -1 if unset (:455-458) */
    int32_t maj_codon;  /* FindMajorityCodon (:380-395), -1 if none */
    int32_t maj_count;
    int32_t skipped;    /* 1 if CallVariants 'continue's before the cell loop */
} msq_position;

/* ErrorEstimates (include/pacbio/juliet/ErrorEstimates.h:57-60). */
typedef struct msq_error_model {
    double match, substitution, deletion;
} msq_error_model;
/* ErrorEstimates(chemistry) src/ErrorEstimates.cpp:44-49 / ErrorEstimates(sub, del) :61-67.  Host. */
int msq_error_model_default(msq_error_model* out);
int msq_error_model_from_rates(double substitution_rate, double deletion_rate, msq_error_model* out);

/* MSARow::Bases (include/pacbio/data/MSA.h:96-101; MSAByRow::AddRead, src/MSA.cpp:144-186) of reads
 * [first_read, first_read + n_rows): d_rows[k * W + c] = the character AddRead leaves at window column c of read
 * first_read + k -- one of "ACGT-N", or ' ' where the read has no base -- with the thresholds applied as in msq_pileup.
 * What MSAByRow::Rows() of the C++ front materialises on first use; the counts never need it.  Async; deferred errors
 * (a base outside the window, "Unexpected base") via msq_pileup_check. */
int msq_rows_decode(msq_ctx* ctx, const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, int32_t win_begin0,
                    int32_t W, uint64_t first_read, uint64_t n_rows, char* d_rows /* [n_rows][W] */);

/* ------------------------------------------------------------------------------- the BAM front on the device
 * IO::BamUtils::BamToArrayReads (src/BamUtils.cpp:56-78) + BAMArrayRead's constructor (src/ArrayRead.cpp:53-99)
 * without pbbam AND without the host in the data path (cpp/msq_bam.hpp is the host front these calls
 * replace stage by stage): the file image crosses PCIe once, compressed; BGZF members are inflated one wave each
 * (csrc/msq_bgzf.hip), the record chain is found and checked, records are filtered (primary alignments that overlap
 * the region, BamUtils.cpp:69-71), measured (reference interval, ArrayBases inside the region) and decoded straight
 * into the packed set the pileup reads (csrc/msq_bamdev.hip); only the small per-record table and the names come back.
 * A file the host front would refuse or that needs one of its exceptions (a block that does not inflate, a malformed
 * record, a bad cigar op, a cigar the MSA throws on) makes the call return MSQ_ERR_DATA: the caller then runs the
 * host front, which reports the error in file order as before.  Parity unpinned at the pbbam boundary, as for the
 * host front; the tests check byte equality with zlib and with the host front's packed sets. */
typedef struct msq_bgzf_member {
    uint64_t in_off;  /* of the member's DEFLATE payload in the file image */
    uint64_t out_off; /* where it inflates to: members follow each other in the output */
    uint32_t in_len;  /* payload bytes (BSIZE + 1 - header - 8) */
    uint32_t out_len; /* ISIZE, <= 65536 */
} msq_bgzf_member;
typedef struct msq_bam_dev msq_bam_dev; /* a BAM file's inflated stream (and, after msq_bam_index_device, its record tables) in HBM */
/* host arrays in (the file image may be a pageable mapping); synchronises.  zlib's inflate of every member, bit for bit.
 * A file image of 16 GiB or more is refused (MSQ_ERR_ARG: the kernel counts the image's dwords in 32 bits). */
int msq_bgzf_inflate_device(msq_ctx* ctx, const uint8_t* file, uint64_t file_bytes, const msq_bgzf_member* members,
                            uint64_t n_members, msq_bam_dev** out);
/* Optional, ahead of msq_bgzf_inflate_device on the same (file, file_bytes): start the file image's way to the device
 * and return at once (the copies are issued by a thread of the library's own: a copy from a pageable mapping blocks its
 * thread).  The caller scans the BGZF headers meanwhile; the inflate call finds the image staged and launches every
 * piece's members as that piece gets on its way.  The file must stay mapped until the inflate call has returned or
 * msq_bgzf_stage_file(ctx, NULL, 0) has forgotten the image.  No other call on the context in between.
 * The upload: four threads of the library (MSQ_BAM_UPLOAD_THREADS; 0 = one plain hipMemcpyAsync per piece) copy slices of
 * the image into a ring of eight pinned 8 MB slots that the context keeps for its life, and every slot crosses by DMA
 * as soon as it is full (50 instead of 17 GB/s from a pageable mapping); images of less than 1 MB take the plain copy. */
int msq_bgzf_stage_file(msq_ctx* ctx, const uint8_t* file, uint64_t file_bytes);
uint64_t msq_bam_dev_bytes(const msq_bam_dev* bam);
/* bytes [off, off + len) of the inflated stream to the host (the BAM header; tests); synchronises */
int msq_bam_dev_read(msq_ctx* ctx, const msq_bam_dev* bam, uint64_t off, uint64_t len, void* host_dst);
int msq_bam_dev_free(msq_ctx* ctx, msq_bam_dev* bam);

enum msq_bam_rec_status {
    MSQ_BAM_REC_SKIP = 0,      /* secondary, supplementary, unmapped or outside the region (BamUtils.cpp:69-71) */
    MSQ_BAM_REC_KEEP = 1,
    MSQ_BAM_REC_MALFORMED = 2, /* fields that do not fit the record's block_size */
    MSQ_BAM_REC_CIGAR = 3,     /* a cigar op > 8, or a cigar longer than SEQ */
    MSQ_BAM_REC_MSA = 4        /* an 'M' or 'N' run inside the region: MSAByRow::AddRead throws on it (src/MSA.cpp:182) */
};
typedef struct msq_bam_record {
    uint64_t body_off;  /* of the record's body (behind block_size) in the inflated stream */
    int64_t n_bases;    /* ArrayBases BAMArrayRead would hold: the alignment columns inside the region, clips excised */
    uint32_t block_size;
    int32_t ref_start, ref_end; /* pos, pos + the cigar's reference length (unclipped) */
    uint32_t l_seq;
    uint32_t sq_off, dq_off, iq_off; /* the sq / dq / iq values from the body's start when all three are there with
                                        l_seq chars (rich QVs, ArrayRead.cpp:74-79), else all 0 */
    uint16_t n_cigar, flag;
    uint8_t l_read_name;
    uint8_t status;   /* msq_bam_rec_status */
    uint8_t rg_len;   /* length of the RG:Z value, 255 without one, 254 when longer than rg[] */
    uint8_t has_qual; /* QUAL present (first byte != 0xFF); QualQV 0 for every base otherwise (ArrayRead.cpp:94-96) */
    char rg[16];
    uint32_t pad_;
} msq_bam_record;
/* The record chain from first_record_off (the byte behind the BAM header) to the end of the stream, every record's
 * consistency, the filters and measurements above; region as BamToArrayReads computes it (0-based, [start, end)).
 * MSQ_ERR_DATA unless the chain ends exactly at the end of the stream and no record is malformed.  Synchronises. */
int msq_bam_index_device(msq_ctx* ctx, msq_bam_dev* bam, uint64_t first_record_off, int32_t n_ref, int32_t region_start0,
                         int32_t region_end0, uint64_t* n_records, uint64_t* n_kept);
/* the kept records' table, and their names back to back (l_read_name bytes each, NUL included), to the host */
int msq_bam_records(msq_ctx* ctx, const msq_bam_dev* bam, msq_bam_record* host_kept /* [n_kept] */);
/* ... or the library's own host copy of that table, [n_kept], valid until msq_bam_dev_free or the next
 * msq_bam_index_device on `bam` (NULL before the index, or when nothing was kept): no second copy of 72 bytes a record */
const msq_bam_record* msq_bam_records_view(const msq_bam_dev* bam);
int msq_bam_names(msq_ctx* ctx, const msq_bam_dev* bam, char* host_buf, uint64_t buf_bytes);
/* Kept records [first, first + count) decoded and packed on the device: the same packed set, bit for bit, as
 * msq_pack_reads_device_fill_compact fed by the host front's decoder (thresholds applied while decoding; pack_flags as
 * there).  MSQ_ERR_DATA when one of them has a status other than KEEP.  Synchronises. */
int msq_bam_pack_device(msq_ctx* ctx, const msq_bam_dev* bam, uint64_t first, uint64_t count, uint32_t pack_flags,
                        const msq_qv_thresholds* thr, msq_packed_dev** out);

/* Diagnostics (no counterpart in the reference; tests ask which path a call took): pileup calls of this process that
 * ran the slabs of a wide window as one merged launch; msq_phase_codons calls that took the lookup gather.  (An
 * experiment build with -DMSQ_WGT also exports msq_debug_wgt, per-workgroup clocks for tools/wgt.py; the shipped
 * library does not.) */
int msq_debug_merged_launches(void);
int msq_debug_phase_lookups(void);
/* tests and tools: the two launcher switches the library otherwise reads from the environment once, when the context is
 * made -- MSQ_NO_MERGE (a launch per column slab instead of one merged launch) and MSQ_CARRIER_MIN_BATCHES (from how many
 * batches per wave on a launch of the read-space kernel takes the carrier rule; < 0: the default).  No counterpart in the
 * reference. */
#define MSQ_TUNE_NO_MERGE 1
#define MSQ_TUNE_CARRIER_MIN_BATCHES 2
#define MSQ_TUNE_NO_HAPS 3 /* msq_reads_align makes no haplotype groups (MSQ_NO_HAPS in the environment) */
#define MSQ_TUNE_BAM_PIECES 5 /* msq_bgzf_inflate_device: pieces a large file crosses PCIe in, each inflated behind its copy (default 3) */
#define MSQ_TUNE_BAM_PIECE_MIN_BYTES 6 /* ... and from which file size on (default 32 MB; tests set 0) */
#define MSQ_TUNE_BAM_SEGMENT 4 /* msq_bam_index_device: bytes per segment of the record-start search (default 16384, >= 64) */
int msq_debug_tune(msq_ctx* ctx, int key, long long value);
/* tools: event counts of the aligned-set pileup kernel since the last call -- {queue entries, drains, nucleotide atomics,
 * codon atomics, alt-haplotype LDS adds, isolated-substitution LDS adds, 0, 0}; all zero unless the library was built
 * with -DMSQ_ROWS_EXP=4.  No counterpart in the reference. */
int msq_debug_rows_counters(unsigned long long* out8);
/* Device memory the context has allocated so far and still holds (scratch, msq_pileup_step's deviation matrix, the
 * log-factorial table): a call that promises not to allocate (msq_pileup_noalloc) leaves it unchanged. */
uint64_t msq_debug_ctx_device_bytes(const msq_ctx* ctx);

/* Reserve the log-factorial table factln(n) = lgamma(n+1.0), n <= max_n (src/Fisher.cpp:115-125), built on
 * the host with the same libm lgamma the reference calls and kept in HBM by the context.  A table that is
 * too small for a call makes that call fail with MSQ_ERR_RANGE.  msq_call_cells needs max_n >= 2*coverage+1. */
int msq_lfact_reserve(msq_ctx* ctx, int64_t max_n);

/* CountNumberOfTests (src/AminoAcidCaller.cpp:86-103) then the per-(position, codon) cell evaluation of
 * CallVariants (:435-488): coverage, majority/ref/altRef codon, expected = coverage * Probability(ref, codon)
 * (:325-339), p = fisher_exact_tiss(ceil(cnt), ceil(cov-cnt), ceil(exp), ceil(cov-exp)) * numberOfTests
 * clipped to 1.  Outputs (device):
 *   d_positions [P]      msq_position
 *   d_p         [P][64]  f64; -1.0 where the reference evaluates no cell (count 0, ref/altRef codon, skipped)
 *   d_expected  [P][64]  f64 expected count (0 where no cell); may be NULL
 *   d_n_tests   [1]      i32 numberOfTests
 * Async; device-side failures are reported by msq_call_check. */
int msq_call_cells(msq_ctx* ctx, const msq_plan* plan, const uint32_t* d_codon, const msq_error_model* err,
                   double maximal_perc, msq_position* d_positions, double* d_p, double* d_expected,
                   int32_t* d_n_tests);
int msq_call_check(msq_ctx* ctx); /* synchronises */

/* Statistics::Fisher::fisher_exact_tiss (include/pacbio/statistics/Fisher.h:15, src/Fisher.cpp:12-77)
 * evaluated on the GPU for n tables; host arrays in, host array out.  Synchronises. */
int msq_fisher_batch(msq_ctx* ctx, int64_t n, const int32_t* chi11, const int32_t* chi12, const int32_t* chi21,
                     const int32_t* chi22, double* p_out);
int msq_fisher(msq_ctx* ctx, int32_t chi11, int32_t chi12, int32_t chi21, int32_t chi22, double* p_out);

#ifdef __cplusplus
}
#endif
#endif /* MSQ_ABI_H */
