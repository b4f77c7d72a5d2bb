"""msq_pack_reads_device_fill_compact: the packer fed with one translated byte per base by the caller's callback.  The
packed set must be, byte for byte, what msq_pack_reads_device_baked makes of the 8-byte records the bytes stand for --
tables, bit planes, flags, no QV planes -- on every edge set, with thresholds that mask bases, both layouts, several
chunks and threads; the table itself against the translation rule; the first unknown cigar in read order."""
import ctypes as C

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device

pytestmark = pytest.mark.gpu
FILL = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32))


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def table():
    t = abi.lib().msq_compact_table()
    return np.ctypeslib.as_array(t, shape=(65536,)).copy()


def test_table_and_fail_rule():
    T = table()
    at = lambda cig, nuc: int(T[ord(cig) | (ord(nuc) << 8)])
    for k, nuc in enumerate("ACGT"):
        assert at("=", nuc) == k and at("X", nuc) == k          # op 0
        assert at("I", nuc) == (k | 2 << 2)                      # op 2 keeps the code
    assert at("=", "N") == (1 | 1 << 2) and at("D", "-") == (0 | 1 << 2) and at("D", "A") == (1 << 2)
    assert at("=", "-") == (3 | 1 << 2 | 16) and at("X", "Z") == (2 | 1 << 2 | 16)   # exotic symbols
    assert at("I", "N") == (2 << 2 | 32)                                            # inserted non-ACGT
    assert at("P", "*") == at("S", "A") == 3 << 2
    for cig in "MNH*B":
        assert at(cig, "A") == 0x80
    f = abi.lib().msq_compact_fail
    for c in range(64):
        op, code = (c >> 2) & 3, c & 3
        want = ((c & 0x30) | 5) if (op == 0 or (op == 1 and code >= 2)) else c
        assert f(c) == want
        # the header's macro form of the rule (what a decoder's inner loop uses)
        macro = ((c & 0x30) | 5) if ((c & 0x0c) == 0 or ((c & 0x0c) == 0x04 and (c & 2))) else c
        assert macro == want


def compact_pack(ctx, reads, thr, q, plain=False, chunk_bases=0, threads=0, has_qual=None, has_rich=None):
    """The reads through the compact callback (the bytes worked out here, in numpy, per read) -> downloaded packed set."""
    T = table()
    n = int(reads["n_reads"])
    off = np.asarray(reads["base_off"], dtype=np.int64)
    cig, nuc = np.asarray(reads["cigar"], dtype=np.uint8), np.asarray(reads["nuc"], dtype=np.uint8)
    total = int(off[-1]) if n else 0
    hq = np.zeros(total, dtype=bool) if reads.get("qual") is None else (np.ones(total, dtype=bool) if has_qual is None else has_qual.astype(bool))
    hr = np.zeros(total, dtype=bool) if reads.get("subqv") is None else (np.ones(total, dtype=bool) if has_rich is None else has_rich.astype(bool))
    ok = np.ones(total, dtype=bool)
    for t, present, plane in ((thr[0], hq, "qual"), (thr[1], hr, "delqv"), (thr[2], hr, "subqv"), (thr[3], hr, "insqv")):
        if t >= 0 and reads.get(plane) is not None:
            ok &= ~present | (np.asarray(reads[plane][:total]).astype(np.int64) >= t)
    c = T[cig[:total].astype(np.int64) | (nuc[:total].astype(np.int64) << 8)]
    bad = (c & 0x80) != 0
    c = np.where(bad, 12, c).astype(np.uint8)
    op, code = (c >> 2) & 3, c & 3
    fail = ~ok & ~bad & ((op == 0) | ((op == 1) & (code >= 2)))
    c = np.where(fail, (c & 0x30) | 5, c).astype(np.uint8)

    def fill(user, r, dst, presence):
        a, b = int(off[r]), int(off[r + 1])
        if b > a:
            C.memmove(dst, c[a:b].ctypes.data, b - a)
        presence[0] |= (1 if hq[a:b].any() else 0) | (2 if hr[a:b].any() else 0)
        w = np.flatnonzero(bad[a:b])
        return int((int(w[0]) << 8) | int(cig[a + int(w[0])])) if len(w) else -1
    cb = FILL(fill)
    lens = np.diff(off).astype(np.int64)
    h = C.c_void_p()
    L = abi.lib()
    abi.check(L.msq_pack_reads_device_fill_compact(ctx.h, n, abi.ptr(reads["ref_start"]), abi.ptr(reads["ref_end"]), abi.ptr(lens), cb, None,
                                                   abi.MSQ_PACK_PLAIN if plain else 0, threads, chunk_bases, q, C.byref(h)))
    return device.LibReads(ctx, h)


@pytest.mark.parametrize("plain", [False, True])
def test_same_packed_set_as_the_baked_packer(ctx, plain):
    sets = dict(cases.edge_sets())
    sets["random_rich"] = (cases.random_reads(61, 120, 500, rich=True, p_ins=0.05), (40, 25, 42, 30))
    sets["random_qual"] = (cases.random_reads(62, 90, 1300, p_ins=0.03, p_del=0.04), (60, -1, -1, -1))
    for name, (reads, thr) in sets.items():
        thr = tuple(thr[:4])
        q = abi.QvThresholds.make(*thr)
        want = device.pack_device(ctx, reads, plain=plain, thr=q)
        for chunk_bases, threads in ((0, 0), (600, 3)):
            got_set = compact_pack(ctx, reads, thr, q, plain=plain, chunk_bases=chunk_bases, threads=threads)
            got = got_set.download()
            assert got_set.flags == want["flags"], (name, got_set.flags, want["flags"])
            assert got["n_groups"] == want["n_groups"]
            for f in ("qual", "delqv", "subqv", "insqv"):
                assert got[f] is None
            for f in ("ref_start", "ref_end", "group_off", "bop"):
                assert (np.asarray(got[f]) == np.asarray(want[f])).all(), (name, f)
            got_set.close()


def test_presence_decides_the_baked_flag_and_the_unknown_cigar(ctx):
    reads = cases.random_reads(63, 60, 300, rich=True)
    total = int(reads["base_off"][-1])
    thr = (50, -1, 42, -1)
    q = abi.QvThresholds.make(*thr)
    none = np.zeros(total, dtype=np.uint8)
    want = device.pack_device(ctx, reads, has_qual=none, has_rich=none, thr=q)  # no base carries a QV: nothing was baked
    got_set = compact_pack(ctx, reads, thr, q, has_qual=none, has_rich=none)
    assert got_set.flags == want["flags"] and not (got_set.flags & abi.MSQ_READS_QV_BAKED)
    assert (np.asarray(got_set.download()["bop"]) == np.asarray(want["bop"])).all()
    got_set.close()
    off = reads["base_off"]
    reads["cigar"][int(off[50]) + 3] = ord("N")
    reads["cigar"][int(off[20]) + 9] = ord("M")  # the first in read order
    with pytest.raises(abi.MsqError) as e:
        device.pack_device(ctx, reads, thr=q)
    for threads in (1, 4):
        with pytest.raises(abi.MsqError) as ec:
            compact_pack(ctx, reads, thr, q, threads=threads, chunk_bases=2000)
        assert ec.value.code == abi.MSQ_ERR_CIGAR and str(ec.value) == str(e.value)
