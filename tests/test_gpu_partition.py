"""The column-slab partition (msq_reads_partition / msq_pileup_partition): per slab a read set of its own, holding of
every read only the groups that reach into the slab.  The counts must be those of msq_pileup over the whole set
(and so the oracle's), the sub-read tables must describe group-aligned pieces of the reads, and the partition must
hold about one copy of the data instead of one per slab."""
import numpy as np
import pytest

import cases
from minorseq_amd import abi, device
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return device.Ctx(0)


@pytest.mark.parametrize("seed,n,W,rich,thr,plain", [
    (61, 80, 5000, False, (-1, -1, -1, -1), False),
    (62, 60, 9000, True, (40, -1, 42, 10), False),
    (63, 40, 13000, False, (-1, -1, -1, -1), True),
    (64, 50, 4097, False, (20, -1, -1, -1), False),
    (65, 30, 700, False, (-1, -1, -1, -1), False),      # one slab: the partition is a copy
])
def test_partition_counts(ctx, seed, n, W, rich, thr, plain):
    reads = cases.random_reads(seed, n, W, rich=rich, p_ins=0.04, p_del=0.03)
    m = orc.Msa(reads, thr)
    q = abi.QvThresholds.make(*thr)
    packed = device.pack_device(ctx, reads, plain=plain)
    dr = device.upload(packed)
    b0, e0 = device.window_bounds(ctx, dr)
    assert e0 - b0 == W
    part = device.Partition(ctx, dr, b0, W)
    assert part.n_slabs == (((W + 31) // 32) + 127) // 128
    nuc, codon = part.pileup(q)
    assert (nuc.cpu().numpy()[:W * 6].reshape(-1, 6) == m.nuc_counts()).all()
    assert (codon.cpu().numpy()[:W * 64].reshape(-1, 64) == m.codon_counts()).all()
    # the tables: every sub-read is a run of whole groups of its read; ReferenceStart = the column of its first group
    goff = packed["group_off"].astype(np.int64)
    n_groups_read = np.diff(goff)
    gslab = -(-((W + 31) // 32) // part.n_slabs) * 32
    total = 0
    for s in range(part.n_slabs):
        rs, re, off = part.tables(s)
        cnt = np.diff(off.astype(np.int64))
        assert (cnt <= n_groups_read).all() and (cnt >= 0).all()
        assert (rs >= b0).all() and (re <= e0).all() and (rs <= re).all()
        lo, hi = b0 + s * gslab, b0 + min(W, (s + 1) * gslab) + 2
        covered = (packed["ref_start"] < hi) & (packed["ref_end"] > lo)
        assert (cnt[~covered] == 0).all()  # a read that does not reach the slab contributes nothing
        assert (cnt[covered & (packed["ref_end"] - packed["ref_start"] > 0)] > 0).all()
        # the sub-read starts at or before the slab (or at the read's own start) and spans it: at most one group of slack
        has = cnt > 0
        assert (rs[has] <= np.maximum(lo, packed["ref_start"][has])).all()
        assert (re[has] >= np.minimum(hi, packed["ref_end"][has])).all()
        total += int(cnt.sum())
    assert total == sum(part.n_groups())
    if part.n_slabs > 1:  # about one copy of the data: each read gains at most a few groups per slab border
        assert total <= int(n_groups_read.sum()) + 4 * part.n_slabs * len(n_groups_read)
    part.close()


def test_partition_of_nothing(ctx):
    empty = device.upload(device.pack_device(ctx, cases.make_reads([])))
    part = device.Partition(ctx, empty, 0, 9000)
    assert part.n_slabs == 3 and part.n_groups() == [0, 0, 0]
    nuc, codon = part.pileup(abi.QvThresholds.make())
    assert not nuc.any() and not codon.any()
    part.close()
    zero = device.upload(device.pack_device(ctx, cases.make_reads([(5, "", ""), (5, "=" * 40, "ACGT" * 10), (9, "", "")])))
    part = device.Partition(ctx, zero, 0, 5000)
    nuc, codon = part.pileup(abi.QvThresholds.make())
    assert int(nuc.sum()) == 40
    part.close()


def test_partition_shards_add_up(ctx):
    """Two shards of one sample, each partitioned inside the global window: the counts add up to the whole."""
    reads = cases.random_reads(71, 90, 8000)
    m = orc.Msa(reads)
    W = m.W
    q = abi.QvThresholds.make()
    nuc, codon = device.alloc_counts(ctx, W)
    off = reads["base_off"]
    for lo, hi in ((0, 40), (40, reads["n_reads"])):
        shard = {"n_reads": hi - lo, "ref_start": reads["ref_start"][lo:hi], "ref_end": reads["ref_end"][lo:hi],
                 "base_off": (off[lo:hi + 1] - off[lo]).astype(np.int64)}
        for f in ("cigar", "nuc", "qual"):
            shard[f] = reads[f][off[lo]:off[hi]]
        dr = device.upload(device.pack_device(ctx, shard))
        part = device.Partition(ctx, dr, m.begin1 - 1, W)
        part.pileup(q, nuc=nuc, codon=codon)
        part.close()
    assert (nuc.cpu().numpy()[:W * 6].reshape(-1, 6) == m.nuc_counts()).all()
    assert (codon.cpu().numpy()[:W * 64].reshape(-1, 64) == m.codon_counts()).all()
