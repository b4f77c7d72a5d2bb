"""ctypes binding of include/msq_abi.h and include/msq_synth.h (libmsq_hip.so).

Harness plumbing only (tests, bench.py, smoke): the product is the C-ABI library and the C++ front in
minorseq_amd/cpp.  There is no CPU fallback: loading fails loudly when the HIP library is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSQ_LIB_DIR: an experiment build of the library (tools/exp_run.sh); the shipped one lives in lib/
LIB_PATH = os.path.join(_HERE, os.environ.get("MSQ_LIB_DIR", "lib"), "libmsq_hip.so")

MSQ_OK = 0
MSQ_ERR_ARG, MSQ_ERR_HIP, MSQ_ERR_CIGAR, MSQ_ERR_BASE = -1, -2, -3, -4
MSQ_ERR_WINDOW, MSQ_ERR_RANGE, MSQ_ERR_NODEVICE, MSQ_ERR_NOMEM = -5, -6, -7, -8
MSQ_ERR_DATA = -9
MSQ_PILEUP_AUTO, MSQ_PILEUP_SIMPLE, MSQ_PILEUP_STREAM, MSQ_PILEUP_ROWS = 0, 1, 2, 3
MSQ_READS_HAS_EXOTIC = 1
MSQ_READS_NORMALIZED = 2
MSQ_READS_PAD_MARKED = 4
MSQ_READS_QV_BAKED = 8
MSQ_READS_INS_EXOTIC = 16
MSQ_READS_ALIGNED = 32
MSQ_PACK_PLAIN = 1
MSQ_COMM_ID_BYTES = 128
MSQ_TUNE_NO_MERGE, MSQ_TUNE_CARRIER_MIN_BATCHES, MSQ_TUNE_NO_HAPS, MSQ_TUNE_BAM_SEGMENT, MSQ_TUNE_BAM_PIECES = 1, 2, 3, 4, 5
MSQ_TUNE_BAM_PIECE_MIN_BYTES = 6
MSQ_BAM_REC_SKIP, MSQ_BAM_REC_KEEP, MSQ_BAM_REC_MALFORMED, MSQ_BAM_REC_CIGAR, MSQ_BAM_REC_MSA = 0, 1, 2, 3, 4


class MsqError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("msq error %d: %s" % (code, text))
        self.code = code


class PackedReads(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("n_groups", C.c_uint64),
        ("ref_start", C.c_void_p),
        ("ref_end", C.c_void_p),
        ("group_off", C.c_void_p),
        ("bop", C.c_void_p),
        ("qual", C.c_void_p),
        ("delqv", C.c_void_p),
        ("subqv", C.c_void_p),
        ("insqv", C.c_void_p),
        ("flags", C.c_uint32),
        ("align_origin", C.c_int32),
        ("gcol", C.c_void_p),
        ("guide", C.c_void_p),
        ("guide_words", C.c_uint32),
        ("hap_count", C.c_uint32),
        ("hap_guides", C.c_void_p),
        ("hap_perm", C.c_void_p),
        ("hap_reads", C.c_uint32 * 8),
    ]


class QvThresholds(C.Structure):
    _fields_ = [("qual", C.c_int32), ("del_", C.c_int32), ("sub", C.c_int32), ("ins", C.c_int32)]

    @classmethod
    def make(cls, qual=-1, del_=-1, sub=-1, ins=-1):
        return cls(qual, del_, sub, ins)


class Position(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "gene", "i", "aa_pos", "win_pos", "coverage", "ref_codon", "alt_codon", "maj_codon", "maj_count", "skipped")]


class ErrorModel(C.Structure):
    _fields_ = [("match", C.c_double), ("substitution", C.c_double), ("deletion", C.c_double)]


class BgzfMember(C.Structure):
    _fields_ = [("in_off", C.c_uint64), ("out_off", C.c_uint64), ("in_len", C.c_uint32), ("out_len", C.c_uint32)]


class BamRecord(C.Structure):
    _fields_ = [
        ("body_off", C.c_uint64),
        ("n_bases", C.c_int64),
        ("block_size", C.c_uint32),
        ("ref_start", C.c_int32),
        ("ref_end", C.c_int32),
        ("l_seq", C.c_uint32),
        ("sq_off", C.c_uint32),
        ("dq_off", C.c_uint32),
        ("iq_off", C.c_uint32),
        ("n_cigar", C.c_uint16),
        ("flag", C.c_uint16),
        ("l_read_name", C.c_uint8),
        ("status", C.c_uint8),
        ("rg_len", C.c_uint8),
        ("has_qual", C.c_uint8),
        ("rg", C.c_char * 16),
        ("pad_", C.c_uint32),
    ]


class SynthParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("win_begin0", C.c_int32),
        ("W", C.c_int32),
        ("thr_sub", C.c_uint32),
        ("thr_del", C.c_uint32),
        ("thr_ins", C.c_uint32),
        ("thr_minor", C.c_uint32),
        ("n_minors", C.c_int32),
        ("thr_jitter", C.c_uint32),
        ("max_jitter", C.c_int32),
        ("rich_qv", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/*.h declares
_P = C.c_void_p
_SIGNATURES = {
    "msq_last_error": (C.c_char_p, []),
    "msq_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "msq_ctx_destroy": (C.c_int, [_P]),
    "msq_ctx_set_stream": (C.c_int, [_P, _P]),
    "msq_ctx_sync": (C.c_int, [_P]),
    "msq_dev_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "msq_dev_free": (C.c_int, [_P, _P]),
    "msq_dev_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "msq_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msq_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msq_memcpy_d2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msq_qv_thresholds_from_env": (C.c_int, [C.POINTER(QvThresholds)]),
    "msq_pack_reads": (C.c_int, [C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    "msq_packed_host_view": (C.POINTER(PackedReads), [_P]),
    "msq_packed_host_free": (C.c_int, [_P]),
    "msq_reads_upload": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(_P)]),
    "msq_packed_dev_view": (C.POINTER(PackedReads), [_P]),
    "msq_reads_bake_qv": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.POINTER(_P)]),
    "msq_packed_dev_free": (C.c_int, [_P, _P]),
    "msq_reads_align": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.POINTER(_P)]),
    "msq_pack_reads_device": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, C.c_uint32, C.c_int, C.c_int64, C.POINTER(_P)]),
    "msq_pack_reads_device_fill": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, C.c_uint32, C.c_int, C.c_int64, C.POINTER(_P)]),
    "msq_pack_reads_device_baked": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, C.c_uint32, C.c_int, C.c_int64, C.POINTER(QvThresholds), C.POINTER(_P)]),
    "msq_pack_reads_device_fill_baked": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, C.c_uint32, C.c_int, C.c_int64, C.POINTER(QvThresholds), C.POINTER(_P)]),
    "msq_pack_reads_device_fill_compact": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, C.c_uint32, C.c_int, C.c_int64, C.POINTER(QvThresholds), C.POINTER(_P)]),
    "msq_compact_table": (C.POINTER(C.c_uint8), []),
    "msq_compact_fail": (C.c_uint8, [C.c_uint8]),
    "msq_insertion_counts": (C.c_int, [_P, C.POINTER(PackedReads), C.c_int32, C.c_int32, _P]),
    "msq_insertion_records": (C.c_int, [_P, C.POINTER(PackedReads), C.c_int32, C.c_int32, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msq_window_bounds": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "msq_window_bounds_device": (C.c_int, [_P, C.POINTER(PackedReads), _P]),
    "msq_pileup_device_window": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), _P, C.c_int32, C.c_int32, _P, _P]),
    "msq_reads_partition": (C.c_int, [_P, C.POINTER(PackedReads), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "msq_partition_slabs": (C.c_int, [_P]),
    "msq_partition_view": (C.POINTER(PackedReads), [_P, C.c_int]),
    "msq_partition_free": (C.c_int, [_P, _P]),
    "msq_pileup_partition": (C.c_int, [_P, _P, C.POINTER(QvThresholds), _P, _P]),
    "msq_pileup": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.c_int32, _P, _P, C.c_int]),
    "msq_pileup_noalloc": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.c_int32, _P, _P]),
    "msq_pileup_check": (C.c_int, [_P]),
    "msq_rows_decode": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _P]),
    "msq_bgzf_inflate_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(_P)]),
    "msq_bgzf_stage_file": (C.c_int, [_P, _P, C.c_uint64]),
    "msq_bam_dev_bytes": (C.c_uint64, [_P]),
    "msq_bam_dev_read": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P]),
    "msq_bam_dev_free": (C.c_int, [_P, _P]),
    "msq_bam_index_device": (C.c_int, [_P, _P, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "msq_bam_records": (C.c_int, [_P, _P, _P]),
    "msq_bam_records_view": (C.c_void_p, [_P]),
    "msq_bam_names": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "msq_bam_pack_device": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(QvThresholds), C.POINTER(_P)]),
    "msq_debug_merged_launches": (C.c_int, []),
    "msq_debug_phase_lookups": (C.c_int, []),
    "msq_debug_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong]),
    "msq_debug_rows_counters": (C.c_int, [C.POINTER(C.c_ulonglong)]),
    "msq_debug_ctx_device_bytes": (C.c_uint64, [_P]),
    "msq_phase_codons": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "msq_pileup_begin": (C.c_int, [_P, C.POINTER(PackedReads), _P, _P, C.c_size_t]),
    "msq_reads_index": (C.c_int, [_P, C.POINTER(PackedReads), _P]),
    "msq_pileup_step": (C.c_int, [_P, C.POINTER(PackedReads), C.POINTER(QvThresholds), C.c_int32, C.c_int32, _P, _P]),
    "msq_timing_enable_strided": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "msq_timing_read_path": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "msq_comm_available": (C.c_int, []),
    "msq_comm_shard_bases": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "msq_comm_agree_status": (C.c_int, [_P, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "msq_comm_unique_id": (C.c_int, [_P]),
    "msq_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "msq_comm_destroy": (C.c_int, [_P]),
    "msq_comm_active": (C.c_int, [_P]),
    "msq_comm_rank": (C.c_int, [_P]),
    "msq_comm_world": (C.c_int, [_P]),
    "msq_comm_shard": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "msq_comm_allreduce_bounds": (C.c_int, [_P, _P]),
    "msq_comm_allreduce_counts": (C.c_int, [_P, _P, C.c_size_t]),
    "msq_comm_allreduce_step": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msq_comm_allgather": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "msq_phase_cluster": (C.c_int, [_P, _P, C.c_uint64, C.c_int32, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "msq_timing_enable": (C.c_int, [_P, C.c_int32]),
    "msq_timing_read": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "msq_plan_create": (C.c_int, [_P, C.c_int32, _P, _P, C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "msq_plan_positions": (C.c_int64, [_P]),
    "msq_plan_destroy": (C.c_int, [_P, _P]),
    "msq_error_model_default": (C.c_int, [C.POINTER(ErrorModel)]),
    "msq_error_model_from_rates": (C.c_int, [C.c_double, C.c_double, C.POINTER(ErrorModel)]),
    "msq_lfact_reserve": (C.c_int, [_P, C.c_int64]),
    "msq_call_cells": (C.c_int, [_P, _P, _P, C.POINTER(ErrorModel), C.c_double, _P, _P, _P, _P]),
    "msq_call_check": (C.c_int, [_P]),
    "msq_fisher_batch": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "msq_fisher": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    # include/msq_synth.h
    "msq_synth_host_lengths": (C.c_int, [C.POINTER(SynthParams), C.c_uint64, C.c_uint64, _P]),
    "msq_synth_host_fill": (C.c_int, [C.POINTER(SynthParams), _P, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "msq_synth_dev_group_counts": (C.c_int, [_P, C.POINTER(SynthParams), C.c_uint64, C.c_uint64, _P]),
    "msq_synth_dev_fill": (C.c_int, [_P, C.POINTER(SynthParams), _P, C.c_uint64, C.POINTER(PackedReads)]),
}

_lib = None


def declared_symbols():
    return sorted(_SIGNATURES)


def lib():
    """The loaded libmsq_hip.so.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C minorseq_amd/csrc)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != MSQ_OK:
        raise MsqError(rc, lib().msq_last_error().decode("utf-8", "replace"))
    return rc


def ptr(a):
    """Address of a numpy array / torch tensor / int / None as a void*."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data
