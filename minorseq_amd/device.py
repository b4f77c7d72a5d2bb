"""Harness-side driver of the C ABI with torch as the device allocator (tests, bench.py, smoke()).

torch is plumbing here (HBM allocation, streams, torch.distributed for the count all-reduce); every
computation goes through libmsq_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import abi


class Ctx:
    def __init__(self, device=0):
        self.device = device
        self._h = C.c_void_p()
        abi.check(abi.lib().msq_ctx_create(device, C.byref(self._h)))
        self.use_current_stream()

    def use_current_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        abi.check(abi.lib().msq_ctx_set_stream(self._h, s))

    @property
    def h(self):
        return self._h

    def sync(self):
        abi.check(abi.lib().msq_ctx_sync(self._h))

    def lfact_reserve(self, max_n):
        abi.check(abi.lib().msq_lfact_reserve(self._h, int(max_n)))

    def close(self):
        if self._h:
            abi.lib().msq_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _view(addr, dtype, n):
    if not addr or n == 0:
        return None
    buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def pack_host(reads):
    """msq_pack_reads on unpacked reads (dict as synth.Workload.host_reads returns) -> dict of numpy arrays."""
    L = abi.lib()
    h = C.c_void_p()
    abi.check(L.msq_pack_reads(reads["n_reads"], abi.ptr(reads["ref_start"]), abi.ptr(reads["ref_end"]),
                               abi.ptr(reads["base_off"]), abi.ptr(reads["cigar"]), abi.ptr(reads["nuc"]),
                               abi.ptr(reads.get("qual")), abi.ptr(reads.get("subqv")), abi.ptr(reads.get("delqv")),
                               abi.ptr(reads.get("insqv")), C.byref(h)))
    try:
        v = L.msq_packed_host_view(h).contents
        n, g = int(v.n_reads), int(v.n_groups)
        out = {
            "n_reads": n,
            "n_groups": g,
            "ref_start": _view(v.ref_start, np.int32, n) if n else np.zeros(0, np.int32),
            "ref_end": _view(v.ref_end, np.int32, n) if n else np.zeros(0, np.int32),
            "group_off": _view(v.group_off, np.uint64, n + 1),
            "bop": _view(v.bop, np.uint32, g * 4) if g else np.zeros(0, np.uint32),
            "qual": _view(v.qual, np.uint8, g * 32),
            "delqv": _view(v.delqv, np.uint8, g * 32),
            "subqv": _view(v.subqv, np.uint8, g * 32),
            "insqv": _view(v.insqv, np.uint8, g * 32),
            "flags": int(v.flags),
        }
    finally:
        L.msq_packed_host_free(h)
    return out


def pack_device(ctx, reads, has_qual=None, has_rich=None, threads=0, chunk_bases=0, plain=False, thr=None, keep=False):
    """msq_pack_reads_device on unpacked reads (the dict pack_host takes) -> the packed arrays copied back from HBM,
    in pack_host()'s layout.  has_qual / has_rich: per-base presence bytes (default: present wherever the dict
    holds the QV array).  thr: msq_pack_reads_device_baked with these thresholds instead.  keep: return the
    library-owned device set (LibReads) instead."""
    L = abi.lib()
    n = int(reads["n_reads"])
    off = np.asarray(reads["base_off"], dtype=np.int64)
    total = int(off[-1]) if n else 0
    rec = np.zeros((max(total, 1), 8), dtype=np.uint8)  # msq_array_base records
    rec[:total, 0] = reads["cigar"][:total]
    rec[:total, 1] = reads["nuc"][:total]
    if reads.get("qual") is not None:
        rec[:total, 2] = 1 if has_qual is None else has_qual
        rec[:total, 4] = reads["qual"][:total]
    if reads.get("subqv") is not None:
        rec[:total, 3] = 1 if has_rich is None else has_rich
        rec[:total, 5] = reads["delqv"][:total]
        rec[:total, 6] = reads["subqv"][:total]
        rec[:total, 7] = reads["insqv"][:total]
    ptrs = (rec.ctypes.data + off[:-1] * 8).astype(np.uint64)
    lens = np.diff(off).astype(np.int64)
    h = C.c_void_p()
    if thr is None:
        abi.check(L.msq_pack_reads_device(ctx.h, n, abi.ptr(reads["ref_start"]), abi.ptr(reads["ref_end"]), abi.ptr(ptrs), abi.ptr(lens),
                                          abi.MSQ_PACK_PLAIN if plain else 0, threads, chunk_bases, C.byref(h)))
    else:
        abi.check(L.msq_pack_reads_device_baked(ctx.h, n, abi.ptr(reads["ref_start"]), abi.ptr(reads["ref_end"]), abi.ptr(ptrs),
                                                abi.ptr(lens), abi.MSQ_PACK_PLAIN if plain else 0, threads, chunk_bases, thr, C.byref(h)))
    if keep:  # the library-owned set itself (LibReads), not a copy of its arrays
        return LibReads(ctx, h)
    try:
        v = L.msq_packed_dev_view(h).contents
        g = int(v.n_groups)

        def back(addr, dtype, count):
            if not addr:
                return None
            a = np.zeros(count, dtype=dtype)
            if count:
                abi.check(L.msq_memcpy_d2h(ctx.h, abi.ptr(a), addr, a.nbytes))
            return a
        out = {"n_reads": int(v.n_reads), "n_groups": g, "ref_start": back(v.ref_start, np.int32, n), "ref_end": back(v.ref_end, np.int32, n),
               "group_off": back(v.group_off, np.uint64, n + 1), "bop": back(v.bop, np.uint32, g * 4),
               "qual": back(v.qual, np.uint8, g * 32), "delqv": back(v.delqv, np.uint8, g * 32),
               "subqv": back(v.subqv, np.uint8, g * 32), "insqv": back(v.insqv, np.uint8, g * 32), "flags": int(v.flags),
               "gcol": back(v.gcol, np.uint32, g)}
    finally:
        L.msq_packed_dev_free(ctx.h, h)
    return out


class DeviceReads:
    """Packed reads resident in HBM (torch tensors own the memory) + the msq_packed_reads view of them."""

    FIELDS = ("ref_start", "ref_end", "group_off", "bop", "qual", "delqv", "subqv", "insqv")

    def __init__(self, n_reads, n_groups, tensors, flags=0):
        self.n_reads, self.n_groups, self.t, self.flags = int(n_reads), int(n_groups), tensors, int(flags)
        self.view = abi.PackedReads(self.n_reads, self.n_groups,
                                    *[abi.ptr(tensors.get(f)) for f in self.FIELDS], self.flags, 0, abi.ptr(tensors.get("gcol")))

    def index(self, ctx):
        """Build the group column index (msq_reads_index) and hang it into the view: the pileup / phasing kernels then
        skip phase 1's prefix sum.  Without it (plain `upload`) the same kernels take the sum-and-walk path."""
        if self.n_groups and self.t.get("gcol") is None:
            g = torch.zeros(self.n_groups, dtype=torch.int32, device=self.t["bop"].device)
            abi.check(abi.lib().msq_reads_index(ctx.h, self.view, g.data_ptr()))
            self.t["gcol"] = g
            self.view.gcol = g.data_ptr()
        return self

    def n_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.t.values() if t is not None)


class LibReads:
    """A packed set in HBM that the library owns (msq_packed_dev): the attributes of DeviceReads the harness uses."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle
        v = abi.lib().msq_packed_dev_view(handle).contents
        self.view = abi.PackedReads.from_buffer_copy(v)
        self.n_reads, self.n_groups, self.flags = int(v.n_reads), int(v.n_groups), int(v.flags)

    def n_bytes(self):
        planes = sum(1 for f in ("qual", "delqv", "subqv", "insqv") if getattr(self.view, f))
        return self.n_reads * 8 + (self.n_reads + 1) * 8 + self.n_groups * (16 + 32 * planes + (4 if self.view.gcol else 0))

    def download(self):
        """-> dict of numpy arrays in the pack_host() layout."""
        L, v, n, g = abi.lib(), self.view, self.n_reads, self.n_groups

        def back(addr, dtype, count):
            if not addr:
                return None
            a = np.zeros(count, dtype=dtype)
            if count:
                abi.check(L.msq_memcpy_d2h(self.ctx.h, abi.ptr(a), addr, a.nbytes))
            return a
        return {"n_reads": n, "n_groups": g, "ref_start": back(v.ref_start, np.int32, n), "ref_end": back(v.ref_end, np.int32, n),
                "group_off": back(v.group_off, np.uint64, n + 1), "bop": back(v.bop, np.uint32, g * 4),
                "qual": back(v.qual, np.uint8, g * 32), "delqv": back(v.delqv, np.uint8, g * 32),
                "subqv": back(v.subqv, np.uint8, g * 32), "insqv": back(v.insqv, np.uint8, g * 32), "flags": self.flags,
                "align_origin": int(v.align_origin), "guide": back(v.guide, np.uint32, int(v.guide_words) * 8),
                "hap_count": int(v.hap_count), "hap_reads": [int(x) for x in v.hap_reads],
                "hap_guides": back(v.hap_guides, np.uint32, int(v.hap_count) * int(v.guide_words) * 8),
                "hap_perm": back(v.hap_perm, np.uint32, n if v.hap_perm else 0)}

    def close(self):
        if self.handle is not None:
            abi.lib().msq_packed_dev_free(self.ctx.h, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bake_qv(ctx, dr, thr):
    """msq_reads_bake_qv: the set with the thresholds applied (QV-failing match bases read 'N', no QV planes)."""
    h = C.c_void_p()
    abi.check(abi.lib().msq_reads_bake_qv(ctx.h, dr.view, thr, C.byref(h)))
    return LibReads(ctx, h)


def align(ctx, dr, thr=None, origin=None):
    """msq_reads_align: the set re-laid in column space (MSQ_READS_ALIGNED) on the word grid of `origin` (default: the
    reads' window begin), thresholds applied, with its guide."""
    if origin is None:
        origin = window_bounds(ctx, dr)[0] if dr.n_reads else 0
    h = C.c_void_p()
    abi.check(abi.lib().msq_reads_align(ctx.h, dr.view, thr if thr is not None else abi.QvThresholds.make(), int(origin), C.byref(h)))
    return LibReads(ctx, h)


def upload(packed, device=0, ctx=None):
    """Host arrays of pack_host() -> DeviceReads; with a context the set also gets its group column index."""
    dev = torch.device("cuda", device)
    t = {}
    for f in DeviceReads.FIELDS:
        a = packed.get(f)
        if a is None:
            t[f] = None
            continue
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        if a.size == 0:
            a = np.zeros(4, dtype=a.dtype)
        t[f] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    dr = DeviceReads(packed["n_reads"], packed["n_groups"], t, packed.get("flags", 0))
    return dr.index(ctx) if ctx is not None else dr


def download(dr):
    """DeviceReads -> dict of numpy arrays in the pack_host() layout."""
    out = {"n_reads": dr.n_reads, "n_groups": dr.n_groups, "flags": dr.flags}
    n, g = dr.n_reads, dr.n_groups
    sizes = {"ref_start": n, "ref_end": n, "group_off": n + 1, "bop": 4 * g, "qual": 32 * g, "delqv": 32 * g,
             "subqv": 32 * g, "insqv": 32 * g}
    for f in DeviceReads.FIELDS:
        t = dr.t.get(f)
        if t is None:
            out[f] = None
            continue
        a = t.cpu().numpy()[:sizes[f]]
        if f == "group_off":
            a = a.view(np.uint64)
        elif f == "bop":
            a = a.view(np.uint32)
        out[f] = a
    return out


def synth_device(ctx, wl, first=0, n=None):
    """Generate reads [first, first+n) of a synth.Workload directly in HBM (packed)."""
    n = wl.n_reads - first if n is None else n
    L = abi.lib()
    dev = torch.device("cuda", ctx.device)
    clones = torch.from_numpy(wl.clone_codes).to(dev)
    ng = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    abi.check(L.msq_synth_dev_group_counts(ctx.h, wl.params, first, n, ng.data_ptr()))
    group_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    if n:
        group_off[1:] = torch.cumsum(ng[:n].to(torch.int64), 0)
    g = int(group_off[-1].item())
    t = {
        "ref_start": torch.zeros(max(n, 1), dtype=torch.int32, device=dev),
        "ref_end": torch.zeros(max(n, 1), dtype=torch.int32, device=dev),
        "group_off": group_off,
        "bop": torch.zeros(max(4 * g, 4), dtype=torch.int32, device=dev),
        "qual": torch.zeros(max(32 * g, 32), dtype=torch.uint8, device=dev),
        "delqv": torch.zeros(max(32 * g, 32), dtype=torch.uint8, device=dev) if wl.rich_qv else None,
        "subqv": torch.zeros(max(32 * g, 32), dtype=torch.uint8, device=dev) if wl.rich_qv else None,
        "insqv": torch.zeros(max(32 * g, 32), dtype=torch.uint8, device=dev) if wl.rich_qv else None,
    }
    dr = DeviceReads(n, g, t, abi.MSQ_READS_NORMALIZED)  # the generator writes normalized groups
    abi.check(L.msq_synth_dev_fill(ctx.h, wl.params, clones.data_ptr(), first, dr.view))
    dr.index(ctx)
    ctx.sync()
    return dr


def window_bounds(ctx, dr):
    b, e = C.c_int32(), C.c_int32()
    abi.check(abi.lib().msq_window_bounds(ctx.h, dr.view, C.byref(b), C.byref(e)))
    return b.value, e.value


def window_bounds_device(ctx, dr, out=None):
    """msq_window_bounds_device -> int32 tensor {-begin0, end0} in HBM (no host synchronisation)."""
    if out is None:
        out = torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", ctx.device))
    abi.check(abi.lib().msq_window_bounds_device(ctx.h, dr.view, out.data_ptr()))
    return out


def pileup_device_window(ctx, dr, thr, bounds, begin0, W_max, nuc=None, codon=None, check=True):
    """msq_pileup_device_window: the window is read from `bounds` (device); the caller expects it to start at
    begin0 and its matrices hold W_max columns."""
    if nuc is None:
        nuc, codon = alloc_counts(ctx, W_max)
    abi.check(abi.lib().msq_pileup_device_window(ctx.h, dr.view, thr, bounds.data_ptr(), begin0, W_max, nuc.data_ptr(), codon.data_ptr()))
    if check:
        abi.check(abi.lib().msq_pileup_check(ctx.h))
    return nuc, codon


class Partition:
    """msq_reads_partition: per column slab a packed read set of its own (library-owned HBM)."""

    def __init__(self, ctx, dr, win_begin0, W, max_slab_columns=0):
        self.ctx, self.win_begin0, self.W = ctx, win_begin0, W
        h = C.c_void_p()
        abi.check(abi.lib().msq_reads_partition(ctx.h, dr.view, win_begin0, W, max_slab_columns, C.byref(h)))
        self._h = h
        self.n_slabs = abi.lib().msq_partition_slabs(h)

    def view(self, slab):
        return abi.lib().msq_partition_view(self._h, slab).contents

    def n_groups(self):
        return [int(self.view(s).n_groups) for s in range(self.n_slabs)]

    def tables(self, slab):
        """(ref_start, ref_end, group_off) of a slab's sub-reads, copied back."""
        v = self.view(slab)
        n = int(v.n_reads)
        out = []
        for addr, dtype, count in ((v.ref_start, np.int32, n), (v.ref_end, np.int32, n), (v.group_off, np.uint64, n + 1)):
            a = np.zeros(count, dtype=dtype)
            if count and addr:
                abi.check(abi.lib().msq_memcpy_d2h(self.ctx.h, abi.ptr(a), addr, a.nbytes))
            out.append(a)
        return out

    def pileup(self, thr, nuc=None, codon=None, check=True):
        if nuc is None:
            nuc, codon = alloc_counts(self.ctx, self.W)
        abi.check(abi.lib().msq_pileup_partition(self.ctx.h, self._h, thr, nuc.data_ptr(), codon.data_ptr()))
        if check:
            abi.check(abi.lib().msq_pileup_check(self.ctx.h))
        return nuc, codon

    def close(self):
        if self._h:
            abi.lib().msq_partition_free(self.ctx.h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def alloc_counts(ctx, W):
    dev = torch.device("cuda", ctx.device)
    return (torch.zeros(max(W, 1) * 6, dtype=torch.int32, device=dev),
            torch.zeros(max(W, 1) * 64, dtype=torch.int32, device=dev))


def pileup(ctx, dr, thr, win_begin0, W, nuc=None, codon=None, mode=abi.MSQ_PILEUP_AUTO, check=True):
    """msq_pileup into (new or given) count matrices; returns (nuc [W,6], codon [W,64]) int32 tensors."""
    if nuc is None:
        nuc, codon = alloc_counts(ctx, W)
    abi.check(abi.lib().msq_pileup(ctx.h, dr.view, thr, win_begin0, W, nuc.data_ptr(), codon.data_ptr(), mode))
    if check:
        abi.check(abi.lib().msq_pileup_check(ctx.h))
    return nuc, codon


def insertion_counts(ctx, dr, win_begin0, W):
    """msq_insertion_counts -> numpy [W]: rows recording an insertion at each window column."""
    counts = torch.zeros(max(W, 1), dtype=torch.int32, device="cuda")
    abi.check(abi.lib().msq_insertion_counts(ctx.h, dr.view, win_begin0, W, counts.data_ptr()))
    return counts.cpu().numpy()[:W].astype(np.int64)


def insertion_records(ctx, dr, win_begin0, W, cols=None, capacity=None, reads=None):
    """msq_insertion_records -> list of (col, string); cols = iterable of window columns to keep (None = all).
    reads: the unpacked reads the set was packed from; with them a run longer than 64 bases, or any run of a set
    flagged MSQ_READS_INS_EXOTIC, is rebuilt from the read's own bases (record.read / record.first_base), as the
    C++ front does; without them only the 2-bit prefix of such a run is returned."""
    filt = None
    if cols is not None:
        bits = np.zeros((W + 31) // 32 + 1, dtype=np.uint32)
        for c in cols:
            bits[c >> 5] |= np.uint32(1 << (c & 31))
        filt = torch.from_numpy(bits.view(np.int32)).cuda()
    if capacity is None:
        capacity = int(insertion_counts(ctx, dr, win_begin0, W).sum())
    rec = torch.zeros(max(capacity, 1) * 4, dtype=torch.int64, device="cuda")  # msq_insertion = 32 bytes
    n = C.c_uint64(0)
    abi.check(abi.lib().msq_insertion_records(ctx.h, dr.view, win_begin0, W, None if filt is None else filt.data_ptr(),
                                              rec.data_ptr(), capacity, C.byref(n)))
    raw = rec.cpu().numpy()[:n.value * 4].reshape(-1, 4)
    out = []
    for head, lo, hi, loc in raw:
        col, ln = int(head & 0xFFFFFFFF), int((head >> 32) & 0xFFFFFFFF)
        if col >= 1 << 31:
            col -= 1 << 32
        bits = (int(lo) & 0xFFFFFFFFFFFFFFFF) | ((int(hi) & 0xFFFFFFFFFFFFFFFF) << 64)
        s = "".join("ACGT"[(bits >> (2 * k)) & 3] for k in range(min(ln, 64)))
        if reads is not None and (ln > 64 or dr.flags & abi.MSQ_READS_INS_EXOTIC):
            r, first = int(loc & 0xFFFFFFFF), int((loc >> 32) & 0xFFFFFFFF)
            at = int(reads["base_off"][r]) + first
            assert bytes(reads["cigar"][at:at + ln]) == b"I" * ln
            whole = bytes(reads["nuc"][at:at + ln]).decode()
            assert "".join(c if c in "ACGT" else "A" for c in whole[:64]) == s  # the packed prefix is the same run
            s = whole
        out.append((col, s))
    return out


class Plan:
    def __init__(self, ctx, gene_begin, gene_end, ref_seq, begin1, W):
        self.ctx = ctx
        self._h = C.c_void_p()
        gb = np.ascontiguousarray(gene_begin, dtype=np.int32)
        ge = np.ascontiguousarray(gene_end, dtype=np.int32)
        ref = ref_seq.encode() if isinstance(ref_seq, str) else ref_seq
        abi.check(abi.lib().msq_plan_create(ctx.h, len(gb), abi.ptr(gb) if len(gb) else None,
                                            abi.ptr(ge) if len(ge) else None, ref, len(ref) if ref else 0, begin1, W,
                                            C.byref(self._h)))
        self.P = int(abi.lib().msq_plan_positions(self._h))
        dev = torch.device("cuda", ctx.device)
        self.positions = torch.zeros(max(self.P, 1) * 10, dtype=torch.int32, device=dev)
        self.p = torch.zeros(max(self.P, 1) * 64, dtype=torch.float64, device=dev)
        self.expected = torch.zeros(max(self.P, 1) * 64, dtype=torch.float64, device=dev)
        self.n_tests = torch.zeros(1, dtype=torch.int32, device=dev)

    @property
    def h(self):
        return self._h

    def close(self):
        if self._h:
            abi.lib().msq_plan_destroy(self.ctx.h, self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def call_cells(ctx, plan, codon, err=None, maximal_perc=100.0, check=True):
    """msq_call_cells into the plan's output buffers (async unless check)."""
    if err is None:
        err = abi.ErrorModel()
        abi.check(abi.lib().msq_error_model_default(C.byref(err)))
    abi.check(abi.lib().msq_call_cells(ctx.h, plan.h, codon.data_ptr(), C.byref(err), float(maximal_perc),
                                       plan.positions.data_ptr(), plan.p.data_ptr(), plan.expected.data_ptr(),
                                       plan.n_tests.data_ptr()))
    if check:
        abi.check(abi.lib().msq_call_check(ctx.h))


def results(plan):
    """Host copies: positions [P,10] int32, p [P,64], expected [P,64], n_tests."""
    P = plan.P
    return (plan.positions.cpu().numpy()[:P * 10].reshape(P, 10), plan.p.cpu().numpy()[:P * 64].reshape(P, 64),
            plan.expected.cpu().numpy()[:P * 64].reshape(P, 64), int(plan.n_tests.item()))


def fisher_batch(ctx, a, b, c, d):
    a, b, c, d = (np.ascontiguousarray(x, dtype=np.int32) for x in (a, b, c, d))
    p = np.zeros(len(a), dtype=np.float64)
    abi.check(abi.lib().msq_fisher_batch(ctx.h, len(a), abi.ptr(a), abi.ptr(b), abi.ptr(c), abi.ptr(d), abi.ptr(p)))
    return p


def phase_cluster(ctx, keys, n_reads, K, with_rows=True):
    """msq_phase_cluster on a device tensor of key rows [n_reads * K] (int16 view of u16) ->
    (cluster_of [n_reads], first_read [C], count [C], rep_rows [C, K] or None), numpy."""
    dev = torch.device("cuda", ctx.device)
    cluster_of = torch.zeros(max(n_reads, 1), dtype=torch.int32, device=dev)
    first = torch.zeros(max(n_reads, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(max(n_reads, 1), dtype=torch.int32, device=dev)
    rep = torch.zeros(max(n_reads * K, 1), dtype=torch.int16, device=dev) if with_rows else None
    n = C.c_uint64(0)
    abi.check(abi.lib().msq_phase_cluster(ctx.h, keys.data_ptr(), n_reads, K, cluster_of.data_ptr(), first.data_ptr(), count.data_ptr(),
                                          rep.data_ptr() if with_rows else None, n_reads, C.byref(n)))
    c = int(n.value)
    return (cluster_of.cpu().numpy()[:n_reads].astype(np.int64), first.cpu().numpy()[:c].astype(np.int64), count.cpu().numpy()[:c].astype(np.int64),
            rep.cpu().numpy().view(np.uint16)[:c * K].reshape(c, K) if with_rows else None)
