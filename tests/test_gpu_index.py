"""The group column index (msq_reads_index, msq_packed_reads.gcol): derived data that lets the stream kernel skip phase
1's prefix sum and border walk.  The index itself against a host computation from the packed arrays; pileup counts and
phasing keys of a set WITH the index against the same set WITHOUT it (the sum-and-walk path) and the simple kernel, bit
for bit, on every edge set, both group layouts, random reads with heavy indels, wide windows, shards that start at a
read index that is no multiple of 8; library-made sets carry an index; QV-plane kernels ignore it."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth

pytestmark = pytest.mark.gpu
NONE = abi.QvThresholds.make()


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def host_index(p):
    bop = p["bop"].reshape(-1, 4)
    ncons = np.array([bin((~int(w)) & 0xffffffff).count("1") for w in bop[:, 3]], dtype=np.int64)
    out = np.zeros(p["n_groups"], dtype=np.uint32)
    for r in range(p["n_reads"]):
        g0, g1 = int(p["group_off"][r]), int(p["group_off"][r + 1])
        col = int(p["ref_start"][r])
        for g in range(g0, g1):
            out[g] = (0x1fffffff if (col < 0 or col + 1 > 0x1fffffff) else col + 1) | ((r & 7) << 29)  # saturates, never 0
            col += int(ncons[g])
    return out


def both(ctx, packed, thr, window=None):
    plain = device.upload(packed, ctx.device)
    idx = device.upload(packed, ctx.device, ctx=ctx)
    assert not plain.view.gcol and (idx.view.gcol or packed["n_groups"] == 0)
    b, e = window or device.window_bounds(ctx, plain)
    W = e - b
    if W <= 0:
        return
    want = device.pileup(ctx, plain, thr, b, W)
    got = device.pileup(ctx, idx, thr, b, W)
    simple = device.pileup(ctx, plain, thr, b, W, mode=abi.MSQ_PILEUP_SIMPLE)
    for x, y, z in zip(want, got, simple):
        assert torch.equal(x, y) and torch.equal(x, z)
    return plain, idx, b, W


@pytest.mark.parametrize("layout", ["normalized", "plain"])
def test_index_values_and_edge_sets(ctx, layout):
    for name, (reads, thr) in cases.edge_sets().items():
        packed = device.pack_host(reads) if layout == "normalized" else cases.pack_plain(reads)
        if packed["n_groups"]:
            idx = device.upload(packed, ctx.device, ctx=ctx)
            ctx.sync()
            got = idx.t["gcol"].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, host_index(packed)), name
        both(ctx, packed, abi.QvThresholds.make(*thr))
        both(ctx, packed, NONE)


@pytest.mark.parametrize("seed,n,W,p_ins", [(1, 300, 700, 0.03), (2, 2000, 2800, 0.10), (3, 64, 5000, 0.02), (4, 9, 100, 0.5)])
def test_random_reads_and_phase_keys(ctx, seed, n, W, p_ins):
    reads = cases.random_reads(seed, n, W, p_ins=p_ins)
    for packed in (device.pack_host(reads), cases.pack_plain(reads)):
        plain, idx, b, Wd = both(ctx, packed, NONE)
        rng = np.random.RandomState(seed)
        pos = np.array(sorted(set([-1, 0, 1, Wd - 3, Wd - 1] + list(rng.randint(1, Wd - 3, size=40)))), dtype=np.int32)
        keys = [torch.zeros(n * len(pos), dtype=torch.int16, device="cuda") for _ in range(2)]
        for dr, k in ((plain, keys[0]), (idx, keys[1])):
            abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, NONE, b, Wd, len(pos), abi.ptr(pos), k.data_ptr()))
        ctx.sync()
        assert torch.equal(keys[0], keys[1])
        # a window with empty columns on both sides
        both(ctx, packed, NONE, window=(b - 37, b + Wd + 70))


def test_shard_starting_anywhere(ctx):
    """The row of a group in its batch comes from (read index & 7): views that start at any read must still agree."""
    wl = synth.workload("pol_100k", n_reads=3000)
    dr = device.synth_device(ctx, wl, 0, 3000)
    assert dr.view.gcol
    for first, cnt in ((0, 3000), (3, 1001), (5, 16), (1021, 1979)):
        sub = device.synth_device(ctx, wl, first, cnt)  # the same reads generated as a set of their own: its index starts at read 0
        a = device.pileup(ctx, sub, NONE, wl.win_begin0, wl.W)
        b = device.pileup(ctx, sub, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_SIMPLE)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # additivity over shards cut at odd places = the whole set
    whole = device.pileup(ctx, dr, NONE, wl.win_begin0, wl.W)
    nuc, codon = device.alloc_counts(ctx, wl.W)
    for first, cnt in ((0, 1021), (1021, 1979)):
        device.pileup(ctx, device.synth_device(ctx, wl, first, cnt), NONE, wl.win_begin0, wl.W, nuc, codon)
    assert torch.equal(nuc, whole[0]) and torch.equal(codon, whole[1])


def test_library_sets_carry_an_index_and_qv_kernels_ignore_it(ctx):
    reads = cases.random_reads(11, 500, 900, rich=True, p_ins=0.04)
    lib_set = device.pack_device(ctx, reads, keep=True)
    assert lib_set.view.gcol
    packed = device.pack_device(ctx, reads)
    assert np.array_equal(packed["gcol"], host_index(packed))
    baked = device.bake_qv(ctx, lib_set, abi.QvThresholds.make(qual=40, sub=42))
    assert baked.view.gcol and baked.view.gcol != lib_set.view.gcol
    plain = device.upload(device.pack_host(reads), ctx.device)
    b, e = device.window_bounds(ctx, plain)
    for thr in (NONE, abi.QvThresholds.make(qual=40), abi.QvThresholds.make(qual=40, sub=42, ins=20)):
        want = device.pileup(ctx, plain, thr, b, e - b)
        got = device.pileup(ctx, lib_set, thr, b, e - b)
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    want = device.pileup(ctx, plain, abi.QvThresholds.make(qual=40, sub=42), b, e - b)
    got = device.pileup(ctx, baked, NONE, b, e - b)
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    part = device.Partition(ctx, lib_set, b, e - b, 256)
    nuc, codon = part.pileup(NONE)
    want = device.pileup(ctx, plain, NONE, b, e - b)
    assert torch.equal(nuc, want[0]) and torch.equal(codon, want[1])
    part.close()
    baked.close()
    lib_set.close()


def test_columns_beyond_the_index_range(ctx):
    """A read the 29 column bits cannot address is outside every window an indexed kernel takes: it raises the same
    deferred window error as without the index, and a window out there is piled up by the sum-and-walk path."""
    reads = cases.make_reads([(10, "=" * 40, "ACGT" * 10), ((1 << 29) + 10, "=" * 40, "ACGT" * 10)])
    packed = device.pack_host(reads)
    for with_ctx in (None, ctx):
        dr = device.upload(packed, ctx.device, ctx=with_ctx)
        nuc, codon = device.alloc_counts(ctx, 64)
        abi.check(abi.lib().msq_pileup(ctx.h, dr.view, NONE, 0, 64, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_STREAM))
        assert abi.lib().msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    far = cases.make_reads([((1 << 29) + 10, "=" * 40, "ACGT" * 10), ((1 << 29) + 12, "=" * 30, "GTAC" * 7 + "GT")])
    both(ctx, device.pack_host(far), NONE)


def test_negative_columns_never_yield_the_empty_word(ctx):
    """a group whose first column is -1 (ReferenceStart -1, read index & 7 == 0) used to index as the word 0,
    which the indexed kernels read as "lane past the batch" -- its bases vanished without the window error the
    sum-and-walk path raises (the reference's assert(pos >= 0), src/MSA.cpp:149).  Negative columns saturate now."""
    reads = cases.make_reads([(-1, "=" * 40, "ACGT" * 10), (3, "=" * 36, "ACGT" * 9)] + [(0, "=" * 40, "TGCA" * 10)] * 14)
    packed = device.pack_host(reads)
    idx = device.upload(packed, ctx.device, ctx=ctx)
    ctx.sync()
    got = idx.t["gcol"].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, host_index(packed)) and (got != 0).all() and (got[0] & 0x1fffffff) == 0x1fffffff
    for with_ctx in (None, ctx):
        dr = device.upload(packed, ctx.device, ctx=with_ctx)
        nuc, codon = device.alloc_counts(ctx, 64)
        abi.check(abi.lib().msq_pileup(ctx.h, dr.view, NONE, 0, 64, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_STREAM))
        assert abi.lib().msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    # the window the reads themselves span (begin -1): no index is taken for a negative begin, counts as the simple kernel's
    both(ctx, packed, NONE)
