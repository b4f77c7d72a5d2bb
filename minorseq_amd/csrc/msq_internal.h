// msq_internal.h -- shared by the translation units of libmsq_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/msq_abi.h"

// Symbol codes of one MSA cell as the kernels see it (MSARow::Bases chars, include/pacbio/data/MSA.h:99-101).
// 1..6 minus one is the NucleotideToTag order A C G T - N (NucleotideConversion.h:69-88).
enum : int { SYM_BLANK = 0, SYM_A = 1, SYM_C = 2, SYM_G = 3, SYM_T = 4, SYM_GAP = 5, SYM_N = 6 };

// Device-side deferred error word (bit set): reported by msq_pileup_check / msq_call_check.
enum : uint32_t { DEVERR_BASE = 1u, DEVERR_WINDOW = 2u, DEVERR_RANGE = 4u, DEVERR_EXOTIC = 8u };

struct msq_comm;  // msq_comm.hip: the RCCL communicator of a context that is one rank of a read-sharded run

struct msq_ctx {
    int device = 0;
    msq_comm* comm = nullptr;
    hipStream_t stream = nullptr;
    uint32_t* d_err = nullptr;      // [1] deferred error bits
    // [48] small accumulators of kernels that finish their own reduction: all zero between kernels -- the last
    // workgroup of a kernel (atomic ticket) reads the result, hands it on and clears what it used, so no memset
    // and no finishing kernel is launched.  [0..2] window bounds (keyed maxima, ticket), [4..5] their decoded
    // result for the host, [8..9] number of tests (sum, ticket), [16..18] the bounds inside msq_pileup_step's kernel
    uint32_t* d_small = nullptr;
    double* d_lfact = nullptr;      // [lfact_n + 1] lgamma(n + 1.0)
    int64_t lfact_n = -1;
    int n_cu = 256;
    // scratch owned by the context, grown on demand
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // pinned host staging of the device packer (msq_pack_reads_device): two buffers, reused across calls
    void* h_stage[2] = {nullptr, nullptr};
    size_t h_stage_bytes = 0;
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    // msq_pileup_noalloc: this GPU's copy of the built-in scratch array (resolved once, at context creation)
    uint32_t* d_noalloc = nullptr;
    size_t noalloc_bytes = 0;
    // msq_pileup_step: the matrix its kernel's deviation path adds into ([W * 70], grown on demand); all zero between
    // calls (the fold stores matrix + partials into the caller's buffer and clears it), dev_clean says so
    uint32_t* d_dev = nullptr;
    size_t dev_words = 0;
    bool dev_clean = false;
    // measurement aid: event pairs around the dominant pileup kernel (msq_timing_enable)
    // and, for the same records, the end of the step's last kernel (msq_call_cells): K1 .. K2 as SURVEY 8d prices it.
    // stride: only every stride-th pileup is timed (an event costs the stream a ~5 us bubble)
    std::vector<hipEvent_t> ev_begin, ev_end, ev_path_end;
    int ev_used = 0, ev_stride = 1, ev_calls = 0;
    bool ev_open = false, ev_path_open = false;
    // switches read from the environment ONCE, when the context is made (MSQ_NO_MERGE, MSQ_CARRIER_MIN_BATCHES), or set
    // with msq_debug_tune: the launchers look here, never at the environment
    bool no_merge = false;
    int64_t carrier_min_batches = -1;  // < 0: the kernel's default
    bool no_haps = false;              // MSQ_NO_HAPS / MSQ_TUNE_NO_HAPS: msq_reads_align makes no haplotype groups
    // msq_bgzf_inflate_device: the file image's pieces cross PCIe on copy_stream beside the inflate kernels of the pieces before
    hipStream_t copy_stream = nullptr;
    // ... and the pieces' kernels in turn on the context's stream and the two side streams: a piece's kernel takes as long
    // as one wave needs for one member however few its members, so the pieces' kernels overlap instead of queueing
    // msq_bgzf_stage_file: the ring of pinned slots the copy threads fill from the file's mapping (kept for the context's life)
    void* pinned_ring = nullptr;
    uint64_t pinned_ring_bytes = 0;
    uint64_t bam_upload_threads = 4;     // 0: one hipMemcpyAsync per piece straight from the pageable mapping
    hipStream_t side_stream[2] = {nullptr, nullptr};
    hipEvent_t side_event[3] = {nullptr, nullptr, nullptr};  // the tables are in place; side stream 0 / 1 has done its kernels
    // msq_bgzf_stage_file: the file image now in the scratch (host pointer and size as its name), the pieces' ends and events
    const uint8_t* staged_file = nullptr;
    uint64_t staged_bytes = 0, staged_upto[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t staged_event[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int staged_total = 0;                // pieces of the staged image; staged_done of them have their event recorded
    std::atomic<int> staged_done{0}, staged_err{0};
    std::thread* stage_thread = nullptr; // issues the copies (a pageable copy blocks its thread)
    void* d_bam_tables = nullptr;        // msq_bgzf_inflate_device: member table + status words
    size_t bam_tables_bytes = 0;
    uint64_t bam_pieces = 3;             // msq_bgzf_inflate_device: pieces a file of >= bam_piece_min_bytes is uploaded and inflated in
    uint64_t bam_piece_min_bytes = (uint64_t)32 << 20;
    uint64_t bam_first_share_pct = 40;   // two pieces: the first one's share of the file
    uint64_t bam_segment_bytes = 16384;  // msq_bam_index_device: the stream is searched for record starts in segments of this size
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when `lds` exceeds what was last set for this kernel on this device
int msq_ensure_dynamic_lds(int device, const void* kernel, size_t lds);

// Packed reads resident in HBM and owned by the library: the eight arrays of msq_packed_reads (ref_start, ref_end,
// group_off, bop, qual, delqv, subqv, insqv; absent planes NULL) and the view over them.
struct msq_packed_dev {
    // [8]: the group column index, [9]: the guide, [10]: the haplotype groups' guides, [11]: the reads ordered by group
    void* ptrs[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    msq_packed_reads view = {};
};

// Record the begin / end event of the next timing slot (no-ops when timing is off or the slots are used up).
void msq_timing_mark_begin(msq_ctx* ctx);
void msq_timing_mark_end(msq_ctx* ctx);
void msq_timing_mark_path_end(msq_ctx* ctx);

void msq_set_error(const char* fmt, ...);
int msq_fail(int code, const char* fmt, ...);

#define MSQ_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return msq_fail(MSQ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                                  \
    } while (0)

int msq_scratch(msq_ctx* ctx, size_t bytes, void** out);
void msq_bgzf_drop_staged(msq_ctx* ctx);  // msq_bgzf.hip: forget a staged file image (joins its copy thread)
int msq_noalloc_init(msq_ctx* ctx);
// allocate and build the group column index of a library-owned device set (ptrs[8], view.gcol); async
int msq_packed_dev_index(msq_ctx* ctx, msq_packed_dev* d);
// The checks every entry point that streams packed reads shares: read arrays present, bit planes and QV planes
// 16-byte aligned (the kernels load uint4), thresholds in -1..255 (qv_pass_mask's byte arithmetic).
// The device packer over bytes that are already in HBM (msq_bam_pack_device): d_compact holds one translated byte per
// base (msq_pack_reads_device_fill_compact's staging format), read r at base_off[r]; presence as a fill callback reports it.
int msq_pack_compact_resident(msq_ctx* ctx, uint64_t n_reads, const int32_t* ref_start, const int32_t* ref_end, const uint64_t* base_off,
                              const uint64_t* d_base_off, const uint8_t* d_compact, const uint32_t* d_presence, uint32_t pack_flags,
                              const msq_qv_thresholds* thr, msq_packed_dev** out);
int msq_validate_reads(const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, const char* who);
