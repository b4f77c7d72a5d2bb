"""The BAM front on the device behind the inflate (csrc/msq_bamdev.hip): record chain, filters, measurements, names and
the decode into the packed set -- against expectations worked out here from the records the test wrote (the walker of
tests/test_bam_front.py, which is not the parser's walk) and against the device packer fed with those expectations.
Segments of 64 bytes to 16 KiB (records spanning several segments, segments without a record start), regions that clip,
reverse-strand rich QVs under thresholds, files the front must decline (MSQ_ERR_DATA -> the caller runs the host front)."""
import ctypes as C
import struct

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device
from test_bam_front import CHEM, INT_MAX, encode_record, expected_reads, make_records, write_bam, bgzf_block
from test_gpu_bgzf import scan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    abi.lib().msq_debug_tune(c.h, abi.MSQ_TUNE_BAM_SEGMENT, 16384)
    c.close()


@pytest.fixture(scope="module")
def ref():
    rng = np.random.default_rng(5)
    return "".join("ACGT"[i] for i in rng.integers(0, 4, 420))


class DevBam:
    """A BAM file inflated into HBM, its header parsed here."""

    def __init__(self, ctx, path):
        self.ctx, self.L = ctx, abi.lib()
        blob = open(path, "rb").read()
        members = scan(blob)
        arr = (abi.BgzfMember * max(len(members), 1))()
        for i, m in enumerate(members):
            arr[i] = abi.BgzfMember(*m)
        self.h = C.c_void_p()
        abi.check(self.L.msq_bgzf_inflate_device(ctx.h, abi.ptr(np.frombuffer(blob, dtype=np.uint8)), len(blob), C.cast(arr, C.c_void_p), len(members),
                                                  C.byref(self.h)))
        n = self.L.msq_bam_dev_bytes(self.h)
        head = np.zeros(n, dtype=np.uint8)
        abi.check(self.L.msq_bam_dev_read(ctx.h, self.h, 0, n, abi.ptr(head)))
        self.data = head.tobytes()
        assert self.data[:4] == b"BAM\1"
        l_text = struct.unpack_from("<I", self.data, 4)[0]
        off = 8 + l_text
        self.n_ref = struct.unpack_from("<I", self.data, off)[0]
        off += 4
        for _ in range(self.n_ref):
            off += 4 + struct.unpack_from("<I", self.data, off)[0] + 4
        self.first = off

    def index(self, region):
        lo, hi = max(region[0] - 1, 0), max(region[1] - 1, 0)
        n_rec, n_kept = C.c_uint64(), C.c_uint64()
        abi.check(self.L.msq_bam_index_device(self.ctx.h, self.h, self.first, self.n_ref, lo, hi, C.byref(n_rec), C.byref(n_kept)))
        recs = (abi.BamRecord * max(n_kept.value, 1))()
        abi.check(self.L.msq_bam_records(self.ctx.h, self.h, C.cast(recs, C.c_void_p)))
        view = self.L.msq_bam_records_view(self.h)  # the library's own copy of the table: the same bytes
        if n_kept.value:
            assert C.string_at(view, n_kept.value * C.sizeof(abi.BamRecord)) == bytes(recs)[:n_kept.value * C.sizeof(abi.BamRecord)]
        else:
            assert not view
        total = sum(recs[i].l_read_name for i in range(n_kept.value))
        buf = C.create_string_buffer(max(total, 1))
        abi.check(self.L.msq_bam_names(self.ctx.h, self.h, buf, total))
        names, at = [], 0
        for i in range(n_kept.value):
            names.append(buf.raw[at:at + recs[i].l_read_name].split(b"\0")[0].decode())
            at += recs[i].l_read_name
        return n_rec.value, [recs[i] for i in range(n_kept.value)], names, (lo, hi)

    def pack(self, first, count, thr, plain=False):
        h = C.c_void_p()
        abi.check(self.L.msq_bam_pack_device(self.ctx.h, self.h, first, count, abi.MSQ_PACK_PLAIN if plain else 0, thr, C.byref(h)))
        return device.LibReads(self.ctx, h)

    def close(self):
        self.L.msq_bam_dev_free(self.ctx.h, self.h)


def check_tables(recs, names, clip, want):
    lo, hi = clip
    assert len(recs) == len(want)
    for r, name, w in zip(recs, names, want):
        assert (max(r.ref_start, lo), min(r.ref_end, hi), r.n_bases, name) == (w[0], w[1], len(w[2]), w[5])
        rg = r.rg[:r.rg_len].decode() if r.rg_len < 254 else None
        assert (CHEM[rg] if rg is not None else ".") == w[6]
        assert r.status == abi.MSQ_BAM_REC_KEEP
        assert (r.sq_off != 0) == (len(w[4]) == 4 or (not w[2] and r.sq_off != 0))


@pytest.mark.parametrize("segment", [64, 300, 4096, 16384])
@pytest.mark.parametrize("rich", [False, True])
def test_record_tables(ctx, tmp_path, ref, segment, rich):
    abi.check(abi.lib().msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_SEGMENT, segment))
    rng = np.random.default_rng(41 + rich)
    recs = make_records(rng, ref, 400, clean=True, rich=rich)
    # secondary / supplementary / unmapped records are filtered before anything else is looked at
    for k, r in enumerate(recs):
        if k % 17 == 3:
            r["flag"] |= (0x100, 0x800, 0x4)[k % 3]
    path = str(tmp_path / "t.bam")
    write_bam(path, recs, len(ref), block=5000)
    bam = DevBam(ctx, path)
    try:
        for region in [(0, INT_MAX), (100, 200), (150, 151), (400, 5000), (0, 1), (2, 2)]:
            n_rec, got, names, clip = bam.index(region)
            assert n_rec == len(recs)
            kept = [r for r in recs if not (r["flag"] & 0x4)]
            check_tables(got, names, clip, expected_reads(kept, *region))
    finally:
        bam.close()


@pytest.mark.parametrize("rich,thr", [(False, (-1, -1, -1, -1)), (False, (40, -1, -1, -1)), (True, (-1, -1, 42, -1)), (True, (30, 40, 20, 50)), (True, (1, 1, 1, 1))])
@pytest.mark.parametrize("plain", [False, True])
def test_packed_set_equals_the_device_packer(ctx, tmp_path, ref, rich, thr, plain):
    """msq_bam_pack_device == msq_pack_reads_device_baked on the ArrayBases the walker expects, bit for bit."""
    abi.check(abi.lib().msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_SEGMENT, 2048))
    rng = np.random.default_rng(51 + rich)
    recs = make_records(rng, ref, 700, clean=True, rich=rich)
    path = str(tmp_path / "p.bam")
    write_bam(path, recs, len(ref), block=20000)
    q = abi.QvThresholds.make(*thr)
    bam = DevBam(ctx, path)
    try:
        for region in [(0, INT_MAX), (31, 391), (200, 260)]:
            _, got, names, clip = bam.index(region)
            want = expected_reads(recs, *region)
            check_tables(got, names, clip, want)
            rows = [(w[0], w[2], w[3]) + tuple(w[4]) for w in want]
            reads = cases.make_reads(rows, rich=rich)
            reads["ref_end"] = np.array([w[1] for w in want], dtype=np.int32)
            ref_set = device.pack_device(ctx, reads, plain=plain, thr=q)
            for first, count in [(0, len(want)), (len(want) // 3, len(want) // 2)]:
                dev = bam.pack(first, count, q, plain=plain)
                g = dev.download()
                if (first, count) == (0, len(want)):
                    assert dev.flags == ref_set["flags"], (dev.flags, ref_set["flags"])
                    for f in ("ref_start", "ref_end", "group_off", "bop"):
                        assert (np.asarray(g[f]) == np.asarray(ref_set[f])).all(), (region, f)
                else:  # a share of the reads (a sharded front packs only its own)
                    g0 = int(ref_set["group_off"][first])
                    assert (np.asarray(g["ref_start"]) == np.asarray(ref_set["ref_start"][first:first + count])).all()
                    assert (np.asarray(g["group_off"]) == np.asarray(ref_set["group_off"][first:first + count + 1]) - g0).all()
                    n_g = int(g["group_off"][-1])
                    assert (np.asarray(g["bop"]) == np.asarray(ref_set["bop"][4 * g0:4 * (g0 + n_g)])).all()
                for f in ("qual", "delqv", "subqv", "insqv"):
                    assert g[f] is None
                dev.close()
    finally:
        bam.close()


def test_counts_from_the_device_front_match_the_oracle(ctx, tmp_path, ref):
    from oracle import orc
    rng = np.random.default_rng(61)
    recs = make_records(rng, ref, 1500, clean=True, rich=True)
    path = str(tmp_path / "c.bam")
    write_bam(path, recs, len(ref), block=60000)
    region = (31, 391)
    bam = DevBam(ctx, path)
    try:
        _, got, names, clip = bam.index(region)
        want = expected_reads(recs, *region)
        q = abi.QvThresholds.make(-1, -1, 42, -1)
        dev = bam.pack(0, len(got), q)
        b0, e0 = device.window_bounds(ctx, dev)
        nuc, codon = device.pileup(ctx, dev, q, b0, e0 - b0)
        rows = [(w[0], w[2], w[3]) + tuple(w[4]) for w in want]
        reads = cases.make_reads(rows, rich=True)
        reads["ref_end"] = np.array([w[1] for w in want], dtype=np.int32)
        m = orc.Msa(reads, (-1, -1, 42, -1))
        W = e0 - b0
        assert (b0 + 1, e0 + 1) == (m.begin1, m.end1)
        assert (nuc.cpu().numpy()[:W * 6].reshape(W, 6) == m.nuc_counts()).all()
        assert (codon.cpu().numpy()[:W * 64].reshape(W, 64) == m.codon_counts()).all()
        dev.close()
    finally:
        bam.close()


def test_files_the_device_front_declines(ctx, tmp_path, ref):
    """Anything the host front would throw on -- or whose exception text it alone knows -- is MSQ_ERR_DATA here."""
    abi.check(abi.lib().msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_SEGMENT, 512))
    rng = np.random.default_rng(71)
    L = abi.lib()

    def outcome(path, region=(0, INT_MAX)):
        bam = DevBam(ctx, path)
        try:
            bam.index(region)
            return abi.MSQ_OK
        except abi.MsqError as e:
            return e.code
        finally:
            bam.close()

    # 'M' / 'N' cigars inside the region (the MSA's "Unexpected cigar"), fine outside it
    recs = make_records(rng, ref, 60, clean=True)
    recs[40]["cigar"] = [("=", 5), ("N", 3)] + recs[40]["cigar"][1:]
    recs[40]["pos"] = 2
    path = str(tmp_path / "n.bam")
    write_bam(path, recs, len(ref), block=3000)
    assert outcome(path) == abi.MSQ_ERR_DATA
    assert outcome(path, (100, 300)) in (abi.MSQ_OK, abi.MSQ_ERR_DATA)
    ops = "".join(op * n for op, n in recs[40]["cigar"])
    if recs[40]["pos"] + sum(ch in "MDN=X" for ch in ops) > 99:
        assert outcome(path, (12, 300)) == abi.MSQ_OK
    # a cigar longer than SEQ, and a cigar op that does not exist
    recs = make_records(rng, ref, 30, clean=True)
    recs[7]["cigar"] = recs[7]["cigar"] + [("=", 9)]
    write_bam(path, recs, len(ref), block=3000)
    assert outcome(path) == abi.MSQ_ERR_DATA
    # a record whose fields do not fit its block_size; a stream cut inside a record; trailing bytes
    recs = make_records(rng, ref, 30, clean=True)
    text = b"@HD\tVN:1.5\n"
    head = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 4) + b"ref\0" + struct.pack("<I", len(ref))
    body = [encode_record(r) for r in recs]
    bad = bytearray(body[11])
    bad[4 + 16:4 + 20] = struct.pack("<I", 1 << 20)  # l_seq
    for raw in (head + b"".join(body[:11]) + bytes(bad) + b"".join(body[12:]), head + b"".join(body)[:-17], head + b"".join(body) + b"\1\2\3",
                head + b"".join(body[:5]) + struct.pack("<I", 7) + b"x" * 7 + b"".join(body[5:])):
        with open(path, "wb") as f:
            for o in range(0, len(raw), 4000):
                f.write(bgzf_block(raw[o:o + 4000]))
            f.write(bgzf_block(b""))
        assert outcome(path) == abi.MSQ_ERR_DATA
    # and the clean file is taken
    with open(path, "wb") as f:
        f.write(bgzf_block(head + b"".join(body)) + bgzf_block(b""))
    assert outcome(path) == abi.MSQ_OK
    # no records at all
    with open(path, "wb") as f:
        f.write(bgzf_block(head) + bgzf_block(b""))
    bam = DevBam(ctx, path)
    n_rec, got, names, _ = bam.index((0, INT_MAX))
    assert n_rec == 0 and got == []
    dev = bam.pack(0, 0, abi.QvThresholds.make())
    assert dev.download()["n_groups"] == 0
    dev.close()
    bam.close()
