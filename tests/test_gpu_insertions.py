"""The insertion histogram: msq_insertion_counts / msq_insertion_records on plain-layout reads from
the device packer against the oracle's MSAColumn::Insertions() (oracle/oracle.cpp: row.Insertions[pos] = insertion
at every flushing op, std::map overwrite, trailing run dropped; src/MSA.cpp:151-184, 88-90).  Parity unpinned, as
for the pileup: the reference holds no fixture for it."""
import collections

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return device.Ctx(0)


def device_histogram(ctx, reads, cols=None, chunk=0):
    packed = device.pack_device(ctx, reads, plain=True, chunk_bases=chunk)
    assert packed["flags"] & abi.MSQ_READS_PAD_MARKED and not packed["flags"] & abi.MSQ_READS_NORMALIZED
    dr = device.upload(packed)
    b0, e0 = device.window_bounds(ctx, dr)
    W = e0 - b0
    counts = device.insertion_counts(ctx, dr, b0, W)
    hist = [collections.Counter() for _ in range(W)]
    for col, s in device.insertion_records(ctx, dr, b0, W, cols=cols, reads=reads):
        hist[col][s] += 1
    return W, counts, hist, dr, b0


def check_against_oracle(ctx, reads, chunk=0):
    m = orc.Msa(reads)
    W, counts, hist, _, _ = device_histogram(ctx, reads, chunk=chunk)
    assert W == m.W
    total = 0
    for c in range(W):
        want = m.col_insertions(c)
        assert dict(hist[c]) == want, (c, dict(hist[c]), want)
        assert counts[c] == sum(want.values())
        total += counts[c]
    return total


@pytest.mark.parametrize("name", sorted(cases.edge_sets()))
def test_edge_sets(ctx, name):
    reads, _ = cases.edge_sets()[name]
    if reads["n_reads"] == 0:
        pytest.skip("no reads")
    check_against_oracle(ctx, reads)


@pytest.mark.parametrize("seed,W,p_ins", [(1, 150, 0.03), (2, 333, 0.2), (3, 2500, 0.05), (4, 40, 0.5)])
def test_random_reads(ctx, seed, W, p_ins):
    reads = cases.random_reads(seed, 300, W, p_ins=p_ins)
    assert check_against_oracle(ctx, reads, chunk=50000 if seed == 3 else 0) > 100


def ins_reads(rows):
    return cases.make_reads([(s, c, n) for s, c, n in rows] + [(0, "=" * 12, "ACGTACGTACGT")])


def test_flush_rules(ctx):
    """Every branch of CheckInsertion (src/MSA.cpp:152-156): flushed by a match / deletion / P / S, overwritten by a
    later run at the same column, kept when the later run is never flushed, trailing runs dropped, a run flushed
    at the column after the read's last base, and at column W (out of bounds in the reference: dropped)."""
    rows = [
        (2, "==II==", "ACGTAC"),          # plain: "GT" at column 4
        (2, "=IID=", "AGG-C"),            # flushed by a deletion: "GG" at column 3
        (2, "=IP==", "AC*GT"),            # flushed by the pad: "C" at column 3
        (2, "=IIPI==", "ACC*TGT"),        # "CC" then "T" at the same column 3: the later one overwrites
        (2, "=IIPIPI=", "ACC*T*GA"),      # three runs at column 3: the last one counts
        (2, "=IIPII", "ACC*TT"),          # the later run is trailing (never flushed): "CC" stays
        (2, "==III", "ACGGG"),            # trailing run: dropped
        (2, "==IS", "ACGT"),              # flushed by S at the column after the read's last base (4)
        (10, "==IS", "ACGT"),             # ... which is column W = 12 here: dropped
        (0, "II==", "TTAC"),              # leading run at column 0
        (3, "I", "A"),                    # a read that is only an insertion
        (3, "IPS", "A*C"),                # ... flushed by P at its start column
    ]
    reads = ins_reads(rows)
    m = orc.Msa(reads)
    assert m.col_insertions(3) == {"GG": 1, "C": 1, "T": 1, "G": 1, "CC": 1, "A": 1}
    assert m.col_insertions(4) == {"GT": 1, "G": 1}
    assert m.col_insertions(0) == {"TT": 1}
    check_against_oracle(ctx, reads)


def test_long_runs_and_block_borders(ctx):
    """Runs that span group borders, a 64-base run (the record's limit), a run that crosses the 64-group block of
    the scan (base 2048), runs ending on the last slot of a group."""
    rng = np.random.RandomState(9)
    seq = lambda n: "".join("ACGT"[x] for x in rng.randint(0, 4, n))
    rows = []
    for lead in (0, 1, 29, 30, 31, 32, 33, 63):
        for n in (1, 2, 3, 31, 32, 33, 64):
            tail = 5
            rows.append((0, "=" * lead + "I" * n + "=" * tail, seq(lead + n + tail)))
    big = 2040
    for n in (5, 16, 40):  # around base 2048 = the scan's block border
        rows.append((0, "=" * big + "I" * n + "=" * 30, seq(big + n + 30)))
    rows.append((0, "=" * 2100, seq(2100)))
    reads = cases.make_reads(rows)
    check_against_oracle(ctx, reads)
    check_against_oracle(ctx, reads, chunk=3000)


def test_long_and_non_acgt_insertions_keep_the_reference_strings(ctx):
    """The reference stores insertion strings of any length with their literal chars (src/MSA.cpp:151-156,172-174).
    The packed planes hold 2 bits per inserted base and a record 64 bases: a longer run, or one with a non-ACGT base,
    comes back with its place in the read (read, first_base) and is rebuilt from the read's own bases."""
    import random
    rnd = random.Random(9)
    long65 = "".join(rnd.choice("ACGT") for _ in range(65))
    long200 = "".join(rnd.choice("ACGT") for _ in range(200))
    rows = [
        (0, "==" + "I" * 65 + "==", "AC" + long65 + "TT"),
        (0, "====", "ACGT"),
        (0, "=" + "I" * 200 + "===", "A" + long200 + "CGT"),          # spans seven groups
        (0, "==" + "I" * 64 + "==", "AC" + long65[:64] + "TT"),        # exactly 64: still fits the record
        (0, "==III==", "ACGNTGT"),                                     # an 'N' inside an insertion
        (1, "=" + "I" * 70 + "==", "C" + "N" * 70 + "GT"),
    ]
    reads = cases.make_reads(rows)
    packed = device.pack_device(ctx, reads, plain=True)
    assert packed["flags"] & abi.MSQ_READS_INS_EXOTIC
    assert device.pack_host(reads)["flags"] & abi.MSQ_READS_INS_EXOTIC
    m = orc.Msa(reads)
    assert m.col_insertions(2) == {long65: 1, long65[:64]: 1, "GNT": 1, "N" * 70: 1} and m.col_insertions(1) == {long200: 1}
    check_against_oracle(ctx, reads)
    # without the reads only the packed prefix comes back, and nothing fails
    dr = device.upload(packed)
    got = sorted(device.insertion_records(ctx, dr, 0, 4))
    assert (2, long65[:64]) in got and (1, long200[:64]) in got and (2, "GAT") in got
    clean = cases.make_reads(rows[:4])
    assert not device.pack_device(ctx, clean, plain=True)["flags"] & abi.MSQ_READS_INS_EXOTIC
    check_against_oracle(ctx, clean)


def test_column_filter_and_capacity(ctx):
    reads = cases.random_reads(11, 200, 120, p_ins=0.1)
    m = orc.Msa(reads)
    keep = [5, 6, 77, 119]
    W, counts, hist, dr, b0 = device_histogram(ctx, reads, cols=keep)
    for c in range(W):
        assert dict(hist[c]) == (m.col_insertions(c) if c in keep else {})
    with pytest.raises(abi.MsqError) as e:
        device.insertion_records(ctx, dr, b0, W, capacity=3)
    assert e.value.code == abi.MSQ_ERR_RANGE and "capacity 3" in str(e.value)


def test_normalized_reads_are_refused(ctx):
    reads = cases.random_reads(12, 20, 60)
    dr = device.upload(device.pack_device(ctx, reads))
    with pytest.raises(abi.MsqError) as e:
        device.insertion_counts(ctx, dr, 100, 60)
    assert e.value.code == abi.MSQ_ERR_ARG and "plain" in str(e.value)


def test_pileup_accepts_marked_padding(ctx):
    """The plain layout with marked padding through the pileup: the same counts as from normalized reads."""
    reads = cases.random_reads(13, 500, 700, rich=True, exotic=True)
    thr = abi.QvThresholds.make(qual=25, sub=42)
    out = []
    for plain in (False, True):
        dr = device.upload(device.pack_device(ctx, reads, plain=plain))
        b0, e0 = device.window_bounds(ctx, dr)
        for mode in (abi.MSQ_PILEUP_STREAM, abi.MSQ_PILEUP_SIMPLE):
            nuc, codon = device.pileup(ctx, dr, thr, b0, e0 - b0, mode=mode)
            out.append((nuc.cpu().numpy().copy(), codon.cpu().numpy().copy()))
    for o in out[1:]:
        assert (o[0] == out[0][0]).all() and (o[1] == out[0][1]).all()
