"""The BAM decode front (minorseq_amd/cpp/msq_bam.hpp = BamUtils::BamToArrayReads, src/BamUtils.cpp:56-78, and
BAMArrayRead, src/ArrayRead.cpp:53-99, without pbbam).  The test writes its own BAM files (BGZF blocks + records,
from the SAM/BAM specification), lets tests/cpp/bam_test parse them, and compares every ArrayRead with expectations
computed here from the same records by a separate walker.  Parity is unpinned at this boundary: the reference holds
no BAM fixture and pbbam is not vendored; what is checked is the behaviour BamUtils.cpp / ArrayRead.cpp
spell out (primary filter, region overlap test, clip to the region, excised clips, QV tags reversed for reverse-strand
records, QualQV 0 without QUAL).  The GPU test feeds a parsed file through AminoAcidCaller and compares the counts
with the oracle."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import cases
from oracle import orc, pymodel

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "bam_test")
OPS = "MIDNSHP=X"
SEQ16 = "=ACMGRSVTWYHKDBN"
INT_MAX = 2147483647


# ------------------------------------------------------------------------------------------------ BAM writer
def bgzf_block(payload):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(payload) + co.flush()
    bsize = 18 + len(comp) + 8
    assert bsize <= 65536
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def encode_record(rec):
    name = rec["name"].encode() + b"\0"
    seq = rec["seq"]
    packed = bytearray((len(seq) + 1) // 2)
    for i, c in enumerate(seq):
        packed[i >> 1] |= SEQ16.index(c) << (0 if i & 1 else 4)
    qual = bytes(rec["qual"]) if rec["qual"] is not None else b"\xff" * len(seq)
    cigar = b"".join(struct.pack("<I", (n << 4) | OPS.index(op)) for op, n in rec["cigar"])
    tags = b""
    for key, (typ, val) in rec.get("tags", {}).items():
        tags += key.encode() + typ.encode()
        if typ == "Z":
            tags += val.encode() + b"\0"
        elif typ == "i":
            tags += struct.pack("<i", val)
        elif typ == "f":
            tags += struct.pack("<f", val)
        elif typ == "C":
            tags += struct.pack("<B", val)
        elif typ == "B":  # ('S', [..])
            sub, arr = val
            tags += sub.encode() + struct.pack("<I", len(arr)) + b"".join(struct.pack("<H", x) for x in arr)
    body = struct.pack("<iiBBHHHIiii", 0, rec["pos"], len(name), 255, 4680, len(rec["cigar"]), rec["flag"], len(seq), -1, -1, 0)
    body += name + cigar + bytes(packed) + qual + tags
    return struct.pack("<I", len(body)) + body


RG_LINES = ("@RG\tID:abcd1234\tPL:PACBIO\tDS:READTYPE=CCS;BINDINGKIT=100-862-200;SEQUENCINGKIT=100-861-800;BASECALLERVERSION=5.0.0;"
            "FRAMERATEHZ=80.000000\tPU:movie\n@RG\tID:ffff0000\tPL:PACBIO\tDS:READTYPE=CCS\tPU:other\n")
CHEM = {"abcd1234": "100-862-200/100-861-800/5.0.0", "ffff0000": "."}


def write_bam(path, records, ref_len, block=7000):
    text = b"@HD\tVN:1.5\tSO:coordinate\n@SQ\tSN:ref\tLN:%d\n" % ref_len + RG_LINES.encode()
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 4) + b"ref\0" + struct.pack("<I", ref_len)
    raw += b"".join(encode_record(r) for r in records)
    with open(path, "wb") as f:
        for o in range(0, len(raw), block):  # records straddle block borders on purpose
            f.write(bgzf_block(raw[o:o + block]))
        f.write(bgzf_block(b""))  # the BGZF end-of-file marker


# ------------------------------------------------------------------------------------------------ generator
def make_records(rng, ref, n, clean=False, rich=False):
    recs = []
    L = len(ref)
    for r in range(n):
        a = int(rng.integers(0, L - 30))
        b = int(rng.integers(a + 20, L + 1))
        if not clean and rng.random() < 0.5:
            a, b = 0, L
        cols = []  # (op, nt)
        for c in range(a, b):
            if c > a and rng.random() < 0.02:
                for _ in range(int(rng.integers(1, 4))):
                    cols.append(("I", "ACGT"[rng.integers(4)]))
            u = rng.random()
            if u < 0.02 and c > a and c < b - 1:
                cols.append(("D", None))
            elif not clean and u < 0.025 and c > a and c < b - 1:
                cols.append(("N", None))
            elif u < 0.05:
                cols.append(("X", "ACGT".replace(ref[c], "")[rng.integers(3)]))
            elif not clean and u < 0.06:
                cols.append(("M", "N" if rng.random() < 0.3 else ref[c]))
            else:
                cols.append(("=", ref[c]))
            if not clean and rng.random() < 0.004 and c < b - 1:
                cols.append(("P", None))
        if not clean and rng.random() < 0.4:
            cols = [("S", "ACGT"[rng.integers(4)]) for _ in range(int(rng.integers(1, 9)))] + cols
        if not clean and rng.random() < 0.4:
            cols = cols + [("S", "ACGT"[rng.integers(4)]) for _ in range(int(rng.integers(1, 9)))]
        cigar = []
        for op, _ in cols:
            if cigar and cigar[-1][0] == op:
                cigar[-1][1] += 1
            else:
                cigar.append([op, 1])
        if not clean and rng.random() < 0.2:
            cigar = [["H", int(rng.integers(1, 50))]] + cigar + [["H", 3]]
        seq = "".join(nt for _, nt in cols if nt is not None)
        flag = 0
        if rng.random() < 0.5:
            flag |= 0x10
        if not clean:
            u = rng.random()
            flag |= 0x100 if u < 0.05 else 0x800 if u < 0.10 else 0
        qual = None if (not clean and rng.random() < 0.15) else [int(x) for x in rng.integers(0, 94, len(seq))]
        tags = {"np": ("i", int(rng.integers(3, 40))), "rq": ("f", 0.999), "sn": ("B", ("S", [1, 2, 3, 4])), "zm": ("i", r)}
        if rich and (clean or rng.random() < 0.8):
            for t in ("dq", "iq", "sq"):
                tags[t] = ("Z", "".join(chr(33 + int(x)) for x in rng.integers(0, 94, len(seq))))
        if clean or rng.random() < 0.9:
            tags["RG"] = ("Z", "abcd1234" if clean or rng.random() < 0.8 else "ffff0000")
        recs.append({"name": "movie/%d/ccs" % r, "flag": flag, "pos": a, "cigar": [tuple(x) for x in cigar], "seq": seq, "qual": qual,
                     "tags": tags})
    return recs


# ------------------------------------------------------------------------------------------------ expectations
def expected_reads(records, region_start, region_end):
    """What BamToArrayReads makes of the records, from their per-column expansion (not the parser's walk)."""
    lo, hi = max(region_start - 1, 0), max(region_end - 1, 0)
    out = []
    for rec in records:
        if rec["flag"] & 0x900:
            continue
        ops = "".join(op * n for op, n in rec["cigar"])
        ref_len = sum(ch in "MDN=X" for ch in ops)
        start, end = rec["pos"], rec["pos"] + ref_len
        if not (start < hi and end > lo):
            continue
        n = len(rec["seq"])
        rich = all(t in rec["tags"] for t in ("sq", "dq", "iq"))
        rev = bool(rec["flag"] & 0x10)

        def tag(t):
            v = [ord(ch) - 33 for ch in rec["tags"][t][1]]
            return v[::-1] if rev else v
        lanes = [rec["qual"] if rec["qual"] is not None else [0] * n]
        if rich:
            lanes += [tag("sq"), tag("dq"), tag("iq")]
        c_lo, c_hi = max(start, lo), min(end, hi)
        rp, qp = start, 0
        cig, nuc, qs = [], [], [[] for _ in lanes]
        for ch in ops:
            ref_c, qry_c = ch in "MDN=X", ch in "MIS=X"
            keep = ch not in "SH" and ((c_lo <= rp < c_hi) if ref_c else (c_lo <= rp <= c_hi))
            if keep:
                cig.append(ch)
                nuc.append(rec["seq"][qp] if qry_c else ("*" if ch == "P" else "-"))
                for k, lane in enumerate(lanes):
                    qs[k].append(lane[qp] if qry_c else 0)
            rp += ref_c
            qp += qry_c
        chem = CHEM[rec["tags"]["RG"][1]] if "RG" in rec["tags"] else "."
        out.append((c_lo, c_hi, "".join(cig), "".join(nuc), qs, rec["name"], chem))
    return out


def run_bam_test(path, region_start, region_end, *extra):
    assert os.path.exists(BIN), "tests/cpp/bam_test not built: run __graft_entry__.build()"
    out = subprocess.run([BIN, path, str(region_start), str(region_end), *extra], capture_output=True, text=True, timeout=300)
    reads, rest = [], []
    for line in out.stdout.splitlines():
        t = line.split(" ")
        if t[0] == "READ":
            k = t.index("NAME")
            reads.append((int(t[1]), int(t[2]), t[3], t[4], [[int(x) for x in f.split(",")] if f else [] for f in t[5:k]], t[k + 1], t[k + 3]))
        else:
            rest.append(line)
    return out.returncode, reads, rest


@pytest.fixture(scope="module")
def ref():
    rng = np.random.default_rng(5)
    return "".join("ACGT"[i] for i in rng.integers(0, 4, 420))


@pytest.mark.parametrize("region", [(0, INT_MAX), (1, 421), (100, 200), (150, 151), (400, 5000), (0, 1), (2, 2)])
@pytest.mark.parametrize("rich", [False, True])
def test_records_become_array_reads(tmp_path, ref, region, rich):
    rng = np.random.default_rng(11 + rich)
    recs = make_records(rng, ref, 300, rich=rich)
    path = str(tmp_path / "a.bam")
    write_bam(path, recs, len(ref))
    rc, got, _ = run_bam_test(path, *region)
    assert rc == 0
    want = expected_reads(recs, *region)
    assert len(got) == len(want)
    if region == (0, INT_MAX):
        assert len(want) > 200 and any(len(w[4]) == 4 for w in want) == rich
    for g, w in zip(got, want):
        if not w[2]:  # nothing left inside the region: the dump cannot show which QV lanes the (absent) bases had
            g, w = g[:4] + g[5:], w[:4] + w[5:]
        assert g == w


def test_block_sizes_and_threads(tmp_path, ref):
    """Records straddling BGZF blocks of any size, one or many inflate threads: the same reads."""
    rng = np.random.default_rng(3)
    recs = make_records(rng, ref, 120, rich=True)
    want = expected_reads(recs, 0, INT_MAX)
    for block, threads in [(61, 1), (997, 3), (65000, 8), (4096, 16)]:
        path = str(tmp_path / ("b%d.bam" % block))
        write_bam(path, recs, len(ref), block=block)
        rc, got, _ = run_bam_test(path, 0, INT_MAX, "--threads", str(threads))
        assert rc == 0 and got == want


def _member(payload, level=6, extra=b""):
    """A BGZF block; `extra`: further subfields in the gzip extra field, before the BC one."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(payload) + co.flush()
    bsize = 12 + len(extra) + 6 + len(comp) + 8
    assert bsize <= 65536
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6 + len(extra)) + extra + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def _scan_check(path, threads):
    out = subprocess.run([BIN, path, "0", "0", "--scan-check", "--threads", str(threads)], capture_output=True, text=True, timeout=300)
    t = out.stdout.split()
    assert t[0] == "SCAN", out.stdout + out.stderr
    return out.returncode, dict(members=int(t[2]), total=int(t[4]), same=int(t[6]), in_ranges=int(t[8]), error=out.stdout.split("error ")[1].strip())


def test_member_scan_in_ranges_equals_the_serial_walk(tmp_path):
    """ScanBgzf with threads cuts the file into ranges that start at look-alikes of a member header and walks them side by
    side; its table must be the serial walk's, block for block, whatever the file holds: look-alikes inside stored data
    (the range whose walk does not land on the next start sends the scan back to the serial walk), members with further
    extra subfields (no look-alike to be found), a file cut short (the serial walk's error), small files (serial)."""
    rng = np.random.default_rng(31)
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    head = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0]) + b"BC" + bytes([2, 0])

    def build(n, planted=False, extra_every=0):
        parts = []
        for i in range(n):
            off = (i * 52361) % ((1 << 20) - 65000)
            p = noise[off:off + 60000 + (i * 37) % 5000]
            if planted and i % 3 == 0:  # stored blocks full of member headers that are none: BSIZE fields pointing anywhere
                p = (head + struct.pack("<H", (i * 977) % 65536) + noise[off:off + 46]) * 900
            parts.append(_member(p, level=0 if (planted and i % 3 == 0) else 1, extra=b"XY\x04\x00abcd" if extra_every and i % extra_every == 0 else b""))
        return b"".join(parts) + _member(b"")

    path = str(tmp_path / "scan.bgzf")
    # plain: every range start is a member; walked in ranges
    blob = build(400)
    assert len(blob) > 16 << 20
    open(path, "wb").write(blob)
    for threads in (1, 2, 3, 8, 16):
        rc, r = _scan_check(path, threads)
        assert rc == 0 and r["same"] == 1 and r["members"] == 401 and r["in_ranges"] == (threads > 1), (threads, r)
    # look-alikes: whatever the ranges find, the table is the serial walk's
    blob = build(400, planted=True)
    open(path, "wb").write(blob)
    seen = set()
    for threads in (2, 3, 4, 5, 6, 7, 8):
        rc, r = _scan_check(path, threads)
        assert rc == 0 and r["same"] == 1 and r["members"] == 401, (threads, r)
        seen.add(r["in_ranges"])
    assert 0 in seen  # some cut did start on a look-alike
    # members with another subfield in front of BC: nothing looks like the usual header, the serial walk does it
    open(path, "wb").write(build(400, extra_every=1))
    rc, r = _scan_check(path, 8)
    assert rc == 0 and r["same"] == 1 and r["in_ranges"] == 0 and r["members"] == 401
    # ... or only some of them do
    open(path, "wb").write(build(400, extra_every=5))
    rc, r = _scan_check(path, 8)
    assert rc == 0 and r["same"] == 1 and r["members"] == 401
    # a file cut short: the same error either way
    open(path, "wb").write(build(400)[:-40000])
    rc, r = _scan_check(path, 8)
    assert rc == 0 and r["same"] == 1 and r["error"] == "'truncated BGZF block'" and r["in_ranges"] == 0
    # a small file: serial
    open(path, "wb").write(build(40))
    rc, r = _scan_check(path, 8)
    assert rc == 0 and r["same"] == 1 and r["in_ranges"] == 0 and r["members"] == 41


def test_empty_and_broken_files(tmp_path, ref):
    path = str(tmp_path / "e.bam")
    write_bam(path, [], len(ref))
    rc, got, _ = run_bam_test(path, 0, INT_MAX)
    assert rc == 0 and got == []  # juliet then reports "Empty input." (src/tools/JulietWorkflow.cpp:139-142)
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(b"this is not a BGZF file at all........")
    rc, got, rest = run_bam_test(bad, 0, INT_MAX)
    assert rc == 1 and rest and rest[0].startswith("ERR ")
    rng = np.random.default_rng(4)
    write_bam(path, make_records(rng, ref, 20), len(ref))
    data = open(path, "rb").read()
    open(bad, "wb").write(data[:len(data) // 2])
    rc, got, rest = run_bam_test(bad, 0, INT_MAX)
    assert rc == 1 and rest[0].startswith("ERR ")
    rc, got, rest = run_bam_test(str(tmp_path / "missing.bam"), 0, INT_MAX)
    assert rc == 1 and "cannot open" in rest[0]


def test_crafted_bgzf_and_bam_headers_are_refused(tmp_path, ref):
    """Nothing in a BGZF / BAM header is trusted: an extra field or subfield that runs past the file, a BSIZE smaller
    than the block's own header + trailer (the payload length would wrap), a block cut inside its payload, a reference
    name length past the data -- each is a clean 'ERR', never a read outside the mapping."""
    rng = np.random.default_rng(5)
    good = str(tmp_path / "g.bam")
    write_bam(good, make_records(rng, ref, 12), len(ref))
    data = bytearray(open(good, "rb").read())
    bad = str(tmp_path / "bad.bam")

    def refused(blob, what):
        open(bad, "wb").write(bytes(blob))
        rc, got, rest = run_bam_test(bad, 0, INT_MAX)
        assert rc == 1 and rest and rest[0].startswith("ERR "), what

    d = bytearray(data)
    d[10:12] = struct.pack("<H", 0xFFFF)           # XLEN far past the block (and, for a short file, past the file)
    refused(d[:2000], "over-long xlen in a short file")
    refused(d, "over-long xlen")
    d = bytearray(data)
    d[14:16] = struct.pack("<H", 0xFFF0)           # the BC subfield claims 65520 bytes
    refused(d, "subfield past the extra field")
    d = bytearray(data)
    d[16:18] = struct.pack("<H", 10)               # BSIZE 11: smaller than header + trailer
    refused(d, "bsize below the header size")
    bsize = struct.unpack("<H", data[16:18])[0] + 1
    refused(data[:bsize - 5], "block cut inside its trailer")
    refused(data[:30], "block cut inside its payload")
    # a reference name length that runs past the inflated data
    text = b"@HD\tVN:1.5\n"
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 0x7FFFFFF0) + b"ref\0"
    refused(bgzf_block(raw) + bgzf_block(b""), "l_name past the data")
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 3) + struct.pack("<I", 4) + b"ref\0" + struct.pack("<I", 9)
    refused(bgzf_block(raw) + bgzf_block(b""), "n_ref past the data")


@pytest.mark.gpu
@pytest.mark.parametrize("rich", [False, True])
def test_bam_to_counts_matches_oracle(tmp_path, ref, rich):
    """BAM -> ArrayReads -> MSAByRow/AminoAcidCaller on the GPU, against the oracle on the records themselves."""
    rng = np.random.default_rng(21 + rich)
    recs = make_records(rng, ref, 2500, clean=True, rich=rich)
    path = str(tmp_path / "c.bam")
    write_bam(path, recs, len(ref), block=60000)
    region = (31, 391)
    rc, got, rest = run_bam_test(path, *region, "--call")
    assert rc == 0, rest[:3]
    want = expected_reads(recs, *region)
    assert got == want
    rows = [(w[0], w[2], w[3]) + tuple(w[4]) for w in want]
    reads = cases.make_reads(rows, rich=rich)
    reads["ref_end"] = np.array([w[1] for w in want], dtype=np.int32)
    m = orc.Msa(reads, (-1, -1, 42, -1))
    nuc_lines = [l.split(" ") for l in rest if l.startswith("NUC ")]
    begin = [l for l in rest if l.startswith("BEGIN ")][0].split(" ")
    assert (int(begin[1]), int(begin[3])) == (m.begin1, m.end1)
    assert (np.array([[int(x) for x in t[2:8]] for t in nuc_lines]) == m.nuc_counts()).all()
    cod = m.codon_counts()
    got_cod = {int(l.split(" ")[1]): {kv.split(":")[0]: int(kv.split(":")[1]) for kv in l.split(" ")[2:]} for l in rest if l.startswith("CODON ")}
    for i in range(m.W):
        assert got_cod.get(i, {}) == {pymodel.CODONS[k]: int(v) for k, v in enumerate(cod[i]) if v}, i
    assert sum(len(v) for v in got_cod.values()) > 500


@pytest.mark.gpu
@pytest.mark.parametrize("staging", ["compact", "records", "device"])
@pytest.mark.parametrize("rich,clean,region", [(False, True, (31, 391)), (True, True, (0, INT_MAX)), (True, False, (100, 300))])
def test_direct_decode_equals_array_read_path(tmp_path, monkeypatch, ref, rich, clean, region, staging):
    """BamToReadSource + AminoAcidCaller(ReadSource) -- records decoded straight into the device packer's staging
    buffer -- against BamToArrayReads + AminoAcidCaller(reads): the same names, intervals and base counts, and an
    identical dump (counts, codons, variants, p-values), also for records with soft clips, pads, N ops, absent
    QUAL and mixed rich tags (`clean=False`: those only through the read tables, the caller refuses 'N' cigars)."""
    # staging: the decoder translates base by base into one byte each (fillCompact, the default), or writes the 8-byte
    # records; with thresholds in the environment that mask bases (the front reads them per MSA, QvThresholds.cpp:45-56)
    # ... or the whole file is decoded on the GPU (BamToReadSourceDevice: inflate, record chain, filters, decode, pack)
    monkeypatch.setenv("MSQ_PACK_WIDE_STAGING", "1" if staging == "records" else "0")
    how = "--device" if staging == "device" else "--direct"
    if rich:
        monkeypatch.setenv("QUALQV", "30")
        monkeypatch.setenv("DELQV", "40")
    rng = np.random.default_rng(31 + rich + 2 * clean)
    recs = make_records(rng, ref, 1200, clean=clean, rich=rich)
    path = str(tmp_path / "d.bam")
    write_bam(path, recs, len(ref), block=30000)
    want = expected_reads(recs, *region)
    if clean:
        rc_a, reads_a, rest_a = run_bam_test(path, *region, "--call")
        rc_d, _, rest_d = run_bam_test(path, *region, "--call", how, "--threads", "7")
        assert rc_a == 0 and rc_d == 0
        if staging == "device":
            assert "FRONT device" in rest_d
        src = [l.split(" ") for l in rest_d if l.startswith("SRC ")]
        assert [(int(t[1]), int(t[2]), int(t[3]), t[5]) for t in src] == [(w[0], w[1], len(w[2]), w[5]) for w in want]
        body = lambda rest: [l for l in rest if l.split(" ")[0] in ("CASE", "BEGIN", "NUC", "CODON", "NTESTS", "VAR", "MSA", "PERF")]
        assert body(rest_d) == body(rest_a) and len(body(rest_a)) > 400
        # ... and after PhaseVariants() the whole report, msa table and error-mode lines, text for text
        report = lambda rest: rest[rest.index("JSON bam"):rest.index("ENDERRMODE") + 1]
        assert report(rest_d) == report(rest_a) and len(report(rest_a)) > 100
        import json
        rep = json.loads("\n".join(rest_d[rest_d.index("JSON bam") + 1:rest_d.index("ENDJSON")]))
        assert sorted(rep) == ["genes", "haplotype_read_counts", "haplotypes"]
        chems = [l.split(" ")[1] for l in rest_d if l.startswith("CHEMS ")]
        assert chems == [CHEM["abcd1234"]]
    else:
        rc_d, _, rest_d = run_bam_test(path, *region, "--call", how)
        src = [l.split(" ") for l in rest_d if l.startswith("SRC ")]
        if rc_d == 0:  # no 'N' / 'M' cigar survived the region
            if staging == "device":
                assert "FRONT device" in rest_d
            assert [(int(t[1]), int(t[2]), int(t[3]), t[5]) for t in src] == [(w[0], w[1], len(w[2]), w[5]) for w in want]
        else:  # the caller's own refusal, as through the ArrayRead path (src/MSA.cpp:182)
            assert any(l.startswith("ERR Unexpected cigar") for l in rest_d)
            rc_a, _, rest_a = run_bam_test(path, *region, "--call")
            assert rc_a == 1 and [l for l in rest_a if l.startswith("ERR")] == [l for l in rest_d if l.startswith("ERR")]


@pytest.mark.gpu
def test_device_front_reports_errors_like_the_host_front(tmp_path, ref):
    """Files the device front declines go through the host front: the same exception text, in file order."""
    rng = np.random.default_rng(9)
    good = str(tmp_path / "g.bam")
    write_bam(good, make_records(rng, ref, 40, clean=True), len(ref), block=3000)
    data = bytearray(open(good, "rb").read())
    bad = str(tmp_path / "bad.bam")
    blobs = [bytes(data[:len(data) // 2]), b"this is not a BGZF file at all........"]
    d = bytearray(data)
    d[40] ^= 0x55          # inside the first block's DEFLATE payload
    blobs.append(bytes(d))
    d = bytearray(data)
    d[len(d) // 2] ^= 0xFF  # somewhere in a later block
    blobs.append(bytes(d))
    text = b"@HD\tVN:1.5\n"
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 0x7FFFFFF0) + b"ref\0"
    blobs.append(bgzf_block(raw) + bgzf_block(b""))
    blobs.append(bgzf_block(b"BAX\1" + raw[4:]) + bgzf_block(b""))
    recs = make_records(rng, ref, 25, clean=True)
    head = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 4) + b"ref\0" + struct.pack("<I", len(ref))
    body = b"".join(encode_record(r) for r in recs)
    blobs.append(bgzf_block(head + body[:-9]) + bgzf_block(b""))                      # a record cut short
    blobs.append(bgzf_block(head + body[:300]) + bgzf_block(body[300:] + b"\0\0\0\0" + b"x" * 40) + bgzf_block(b""))  # block_size 0
    for k, blob in enumerate(blobs):
        open(bad, "wb").write(blob)
        rc_h, _, rest_h = run_bam_test(bad, 0, INT_MAX, "--call", "--direct")
        rc_d, _, rest_d = run_bam_test(bad, 0, INT_MAX, "--call", "--device")
        assert rc_h == 1 and rc_d == 1, k
        assert [l for l in rest_h if l.startswith("ERR")] == [l for l in rest_d if l.startswith("ERR")] != [], k
    rc_d, _, rest_d = run_bam_test(good, 0, INT_MAX, "--call", "--device")
    assert rc_d == 0 and "FRONT device" in rest_d
    # an empty file of records: juliet's "Empty input." case, no reads from either front
    write_bam(bad, [], len(ref))
    rc_d, _, rest_d = run_bam_test(bad, 0, INT_MAX, "--device")
    assert rc_d == 0 and "FRONT device" in rest_d and not [l for l in rest_d if l.startswith("SRC ")]
