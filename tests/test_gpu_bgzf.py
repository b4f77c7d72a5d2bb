"""msq_bgzf_inflate_device: BGZF members inflated on the GPU, one wave per member (csrc/msq_bgzf.hip).  Byte work: the
test is equality with zlib's output -- stored, fixed and dynamic blocks, several blocks per member (sync / full flushes:
empty stored blocks in between), code lengths beyond the primary tables (a skewed alphabet), matches that overlap
themselves (runs, short periods), members of 0 bytes to 64 KiB, thousands of members per call -- and refusal of every
member zlib refuses (random corruptions, truncations), one member per call so that each outcome is compared."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from minorseq_amd import abi, device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def member(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush_mode=zlib.Z_SYNC_FLUSH, mem_level=8):
    """One BGZF block (SAMv1 4.1) around `payload`, deflated the way the arguments say."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    if flush_every:
        comp = b""
        for o in range(0, len(payload), flush_every):
            comp += co.compress(payload[o:o + flush_every]) + co.flush(flush_mode)
        comp += co.flush()
    else:
        comp = co.compress(payload) + co.flush()
    bsize = 18 + len(comp) + 8
    assert bsize <= 65536, bsize
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def scan(blob):
    """The member table the host front builds (InflateBgzf in cpp/msq_bam.hpp)."""
    members, off, out = [], 0, 0
    while off < len(blob):
        xlen = struct.unpack_from("<H", blob, off + 10)[0]
        bsize = struct.unpack_from("<H", blob, off + 16)[0] + 1
        payload = off + 12 + xlen
        isize = struct.unpack_from("<I", blob, off + bsize - 4)[0]
        members.append((payload, out, off + bsize - 8 - payload, isize))
        out += isize
        off += bsize
    return members


def inflate_device(ctx, blob, members):
    """-> the inflated stream (bytes); raises abi.MsqError as the call does."""
    L = abi.lib()
    arr = (abi.BgzfMember * max(len(members), 1))()
    for i, m in enumerate(members):
        arr[i] = abi.BgzfMember(m[0], m[1], m[2], m[3])
    buf = np.frombuffer(blob, dtype=np.uint8) if len(blob) else np.zeros(1, dtype=np.uint8)
    h = C.c_void_p()
    abi.check(L.msq_bgzf_inflate_device(ctx.h, abi.ptr(buf), len(blob), C.cast(arr, C.c_void_p), len(members), C.byref(h)))
    try:
        n = L.msq_bam_dev_bytes(h)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        abi.check(L.msq_bam_dev_read(ctx.h, h, 0, n, abi.ptr(out)))
        return out[:n].tobytes()
    finally:
        L.msq_bam_dev_free(ctx.h, h)


def payloads(rng):
    skew = np.minimum(rng.geometric(0.08, 60000), 255).astype(np.uint8).tobytes()      # code lengths up to 15 bits
    very_skew = np.minimum(rng.geometric(0.35, 65000), 200).astype(np.uint8).tobytes()
    text = (b"@RG\tID:abcd1234\tPL:PACBIO\tDS:READTYPE=CCS;BINDINGKIT=100-862-200\n" * 700)[:65280]
    quals = rng.integers(0, 94, 65280).astype(np.uint8).tobytes()
    nibbles = rng.integers(0, 256, 65280).astype(np.uint8).tobytes()
    period = (bytes(rng.integers(0, 256, 7).astype(np.uint8)) * 9400)[:65000]               # matches with distance < length
    mixed = b"".join([quals[:9000], b"\0" * 5000, text[:9000], nibbles[:9000], period[:9000], skew[:9000], b"A" * 300, quals[:3000]])
    return {"skew": skew, "very_skew": very_skew, "text": text, "quals": quals, "random": nibbles, "period7": period,
            "zeros": b"\0" * 65536, "one_byte_run": (b"ab" + b"c" * 70000)[:60000], "mixed": mixed, "full": quals + quals[:256]}


def test_every_block_kind_equals_zlib(ctx):
    rng = np.random.default_rng(17)
    P = payloads(rng)
    blocks = []
    for name, p in P.items():
        for kw in (dict(level=6), dict(level=1), dict(level=9), dict(strategy=zlib.Z_FIXED), dict(strategy=zlib.Z_HUFFMAN_ONLY),
                   dict(strategy=zlib.Z_RLE), dict(level=6, flush_every=5000), dict(level=6, flush_every=777, flush_mode=zlib.Z_FULL_FLUSH),
                   dict(level=9, mem_level=1), dict(strategy=zlib.Z_FIXED, flush_every=3001)):
            try:
                blocks.append((name, kw, member(p, **kw)))
            except AssertionError:  # did not compress into 64 KiB this way
                pass
        blocks.append((name, "stored", member(p[:65000], level=0)))
    for size in (0, 1, 2, 61, 255, 256, 257, 4095, 65279, 65280):
        blocks.append(("size%d" % size, "default", member(P["mixed"][:size] if size <= len(P["mixed"]) else P["quals"][:size])))
    blob = b"".join(b for _, _, b in blocks) + member(b"")
    members = scan(blob)
    assert len(members) == len(blocks) + 1 and len(blocks) > 100
    want = b"".join(zlib.decompress(blob[m[0]:m[0] + m[2]], -15) for m in members)
    got = inflate_device(ctx, blob, members)
    assert len(got) == len(want)
    if got != want:  # name the first member that differs
        for (name, kw, _), m in zip(blocks, members):
            assert got[m[1]:m[1] + m[3]] == want[m[1]:m[1] + m[3]], (name, kw)
    assert got == want


def test_many_members_and_odd_payload_offsets(ctx):
    """A BAM-sized run of members (one wave each, thousands of workgroups), payloads at every byte alignment."""
    rng = np.random.default_rng(18)
    parts = []
    for i in range(3000):
        n = int(rng.integers(1, 3000))
        kind = i % 3
        p = (rng.integers(0, 94, n).astype(np.uint8).tobytes() if kind == 0 else
             np.minimum(rng.geometric(0.2, n), 255).astype(np.uint8).tobytes() if kind == 1 else (b"movie/%d/ccs\0" % i) * (n // 12 + 1))
        parts.append(member(p, level=int(rng.integers(1, 10))))
    blob = b"".join(parts) + member(b"")
    members = scan(blob)
    want = b"".join(zlib.decompress(blob[m[0]:m[0] + m[2]], -15) for m in members)
    assert inflate_device(ctx, blob, members) == want
    assert len({m[0] & 3 for m in members}) == 4
    # ... and with the file crossing PCIe in pieces, each piece's members inflated behind its copy (what files of >= 32 MB get)
    L = abi.lib()
    try:
        abi.check(L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECE_MIN_BYTES, 0))
        for pieces in (2, 3, 7):
            abi.check(L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECES, pieces))
            assert inflate_device(ctx, blob, members) == want
    finally:
        L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECE_MIN_BYTES, 32 << 20)
        L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECES, 3)


def test_large_image_crosses_in_ring_slots(ctx):
    """A file image of more than the upload ring (8 pinned slots of 8 MB, filled by the library's copy threads and reused
    as their DMAs complete; msq_bgzf_stage_file): every byte must arrive, in pieces and in one."""
    rng = np.random.default_rng(23)
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    parts, want = [], []
    for i in range(1700):  # ~90 MB: stored blocks of noise (nothing to compress), every 7th member text at level 6
        off = (i * 40093) % ((1 << 20) - 65280)
        p = noise[off:off + 65280 - (i % 5)] if i % 7 else (b"record %d\t" % i) * 4000
        parts.append(member(p, level=0 if i % 7 else 6))
        want.append(p)
    blob = b"".join(parts) + member(b"")
    assert len(blob) > 80 << 20
    members = scan(blob)
    want = b"".join(want)
    L = abi.lib()
    try:
        for pieces in (2, 1, 3):
            abi.check(L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECES, pieces))
            got = inflate_device(ctx, blob, members)
            assert len(got) == len(want)
            assert got == want, pieces
    finally:
        L.msq_debug_tune(ctx.h, abi.MSQ_TUNE_BAM_PIECES, 3)


def test_plain_upload_path(monkeypatch):
    """MSQ_BAM_UPLOAD_THREADS=0: the image crosses in one hipMemcpyAsync per piece straight from the caller's pages (what a
    context does that cannot have pinned memory), the same bytes."""
    monkeypatch.setenv("MSQ_BAM_UPLOAD_THREADS", "0")
    c = device.Ctx(0)
    try:
        rng = np.random.default_rng(29)
        parts = [member(rng.integers(0, 64, int(rng.integers(1000, 60000))).astype(np.uint8).tobytes(), level=int(rng.integers(1, 10))) for _ in range(120)]
        blob = b"".join(parts) + member(b"")
        assert len(blob) > 2 << 20
        members = scan(blob)
        want = b"".join(zlib.decompress(blob[m[0]:m[0] + m[2]], -15) for m in members)
        L = abi.lib()
        abi.check(L.msq_debug_tune(c.h, abi.MSQ_TUNE_BAM_PIECE_MIN_BYTES, 0))
        for pieces in (1, 3):
            abi.check(L.msq_debug_tune(c.h, abi.MSQ_TUNE_BAM_PIECES, pieces))
            assert inflate_device(c, blob, members) == want
    finally:
        c.close()


def test_bam_shaped_members(ctx, tmp_path):
    """Members the way a BAM writer makes them: 65280-byte payloads of records (packed bases, quality bytes, names, tags)
    at level 1 -- long literal runs that end in a length code with extra bits right at the edge of the speculative window
    (the case a first version of the literal decode lost: it left too few bits behind the window)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_bam
    for rich, reads in ((False, 1500), (True, 500)):
        path = str(tmp_path / "b.bam")
        bench_bam.write(path, reads, 2843, rich)
        blob = open(path, "rb").read()
        members = scan(blob)
        assert len(members) > 90
        want = b"".join(zlib.decompress(blob[m[0]:m[0] + m[2]], -15) for m in members)
        assert inflate_device(ctx, blob, members) == want


def zlib_outcome(payload, isize):
    """What the host front's inflate makes of a member: its bytes, or None when it refuses (not Z_STREAM_END within isize)."""
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(payload, isize + 1)
    except zlib.error:
        return None
    if not d.eof or len(out) != isize:
        return None
    return out


def test_corrupt_members_are_refused_like_zlib(ctx):
    rng = np.random.default_rng(19)
    P = payloads(rng)
    base = [member(P["mixed"][:20000]), member(P["skew"][:30000], level=9), member(P["text"][:8000], strategy=zlib.Z_FIXED),
            member(P["quals"][:5000], level=0), member(P["period7"][:30000], flush_every=4000)]
    refused = accepted = 0
    for trial in range(120):
        blob = bytearray(base[trial % len(base)])
        (payload, _, n, isize), = scan(bytes(blob))
        how = trial % 4
        if how == 0:    # a flipped bit somewhere in the stream
            at = payload + int(rng.integers(0, n))
            blob[at] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:  # in the block header / code lengths
            at = payload + int(rng.integers(0, min(n, 40)))
            blob[at] ^= 1 << int(rng.integers(0, 8))
        elif how == 2:  # the stream cut short: the member's payload ends early
            n = int(rng.integers(0, n))
        else:           # ISIZE wrong
            isize = max(0, isize + int(rng.integers(-3, 4)))
        blob = bytes(blob)
        want = zlib_outcome(blob[payload:payload + n], isize) if isize else b""
        try:
            got = inflate_device(ctx, blob, [(payload, 0, n, isize)])
        except abi.MsqError as e:
            assert e.code == abi.MSQ_ERR_DATA and "does not inflate to its ISIZE" in str(e)
            got = None
        assert got == want, (trial, how)
        refused += want is None
        accepted += want is not None
    assert refused > 40 and accepted > 5


def test_arguments(ctx):
    blob = member(b"hello, world") + member(b"")
    m = scan(blob)
    assert inflate_device(ctx, blob, m) == b"hello, world"
    assert inflate_device(ctx, b"", []) == b""
    with pytest.raises(abi.MsqError) as e:  # a member outside the file
        inflate_device(ctx, blob, [(m[0][0], 0, len(blob), 12)])
    assert e.value.code == abi.MSQ_ERR_ARG
    with pytest.raises(abi.MsqError) as e:  # more than a BGZF block can hold
        inflate_device(ctx, blob, [(m[0][0], 0, m[0][2], 70000)])
    assert e.value.code == abi.MSQ_ERR_DATA
    with pytest.raises(abi.MsqError) as e:  # members must follow each other in the output
        inflate_device(ctx, blob, [(m[0][0], 5, m[0][2], 12)])
    assert e.value.code == abi.MSQ_ERR_ARG
