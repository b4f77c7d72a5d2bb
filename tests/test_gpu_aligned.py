"""GPU parity, the column-aligned layout (msq_reads_align, MSQ_READS_ALIGNED) and its pileup kernel
(csrc/msq_pileup_rows.hip): AddRead's placement (src/MSA.cpp:148,158-171) done when the set is made, MSAByColumn +
CodonsAt (src/MSA.cpp:55-92,126-136) over the words as they are loaded.

The aligned set against the oracle's rows (every group IS a 32-column word of its read's row); its guide against the
column majorities; the counts of the aligned-set kernel against the oracle, bit for bit, on every edge set and both
source layouts, with thresholds, windows wider than the reads, random reads with heavy indels, samples that stray far
from their majority (the alt-haplotype and global-atomic paths, queue drains in mid-batch), several launches, the fused
step (msq_pileup_step: bounds + stored matrices) and the no-allocation door; and what the kernel does NOT take falls
back to the read-space kernels with the same counts.  Parity unpinned like the pileup (no MSA fixture in the reference)."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
EDGE = cases.edge_sets()
NONE = abi.QvThresholds.make()
SYM = {ord(" "): None, ord("A"): (0, 0), ord("C"): (0, 1), ord("G"): (0, 2), ord("T"): (0, 3), ord("-"): (1, 0), ord("N"): (1, 1)}


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def expected_words(rows_chars, starts, begin0, origin):
    """The aligned groups of every read from the oracle's rows: word k of a read = columns [origin + 32 (a + k), + 32)."""
    offs, words = [0], []
    for r, row in enumerate(rows_chars):
        cols = np.nonzero(row != 32)[0]
        if len(cols) == 0:
            offs.append(offs[-1])
            continue
        first = begin0 + int(cols[0]) - origin
        last = begin0 + int(cols[-1]) - origin
        assert first == starts[r] - origin
        a = first >> 5
        nw = (last >> 5) - a + 1
        w = np.zeros((nw, 4), dtype=np.uint32)
        w[:, 2] = w[:, 3] = 0xffffffff  # padding: op 3
        for c in cols:
            op, code = SYM[int(row[c])]
            g = begin0 + int(c) - origin
            k, j = (g >> 5) - a, g & 31
            bit = np.uint32(1 << j)
            w[k, 2] &= ~bit
            w[k, 3] &= ~bit
            if op:
                w[k, 2] |= bit
            if code & 1:
                w[k, 0] |= bit
            if code & 2:
                w[k, 1] |= bit
        words.append(w)
        offs.append(offs[-1] + nw)
    return np.array(offs, dtype=np.uint64), (np.concatenate(words) if words else np.zeros((0, 4), dtype=np.uint32))


def counts(ctx, dr, thr, b, W, mode):
    nuc, codon = device.pileup(ctx, dr, thr, b, W, mode=mode)
    return nuc.cpu().numpy()[:W * 6].reshape(W, 6), codon.cpu().numpy()[:W * 64].reshape(W, 64)


@pytest.mark.parametrize("layout", ["normalized", "plain"])
@pytest.mark.parametrize("name", sorted(EDGE) + ["random_indels", "random_rich"])
def test_aligned_set_and_counts_vs_oracle(ctx, name, layout):
    if name == "random_indels":
        reads, thr = cases.random_reads(11, 300, 700, p_ins=0.05), (-1, -1, -1, -1)
    elif name == "random_rich":
        reads, thr = cases.random_reads(12, 200, 2300, rich=True, p_ins=0.05), (30, 20, 42, 25)
    else:
        reads, thr = EDGE[name]
    exotic = name == "random_literalN"
    m = orc.Msa(reads, thr)
    packed = device.pack_host(reads) if layout == "normalized" else cases.pack_plain(reads)
    src = device.upload(packed, ctx.device, ctx=ctx)
    t = abi.QvThresholds.make(*thr)
    b0, W = m.begin1 - 1, m.W
    al = device.align(ctx, src, t, b0)
    assert al.flags & abi.MSQ_READS_ALIGNED and not al.flags & abi.MSQ_READS_NORMALIZED
    got = al.download()
    assert got["align_origin"] == b0 and got["qual"] is None
    if not (al.flags & abi.MSQ_READS_HAS_EXOTIC):
        offs, words = expected_words(m.rows(), reads["ref_start"], b0, b0)
        assert np.array_equal(got["group_off"], offs)
        assert np.array_equal(got["bop"].reshape(-1, 4), words)
    # the guide: with at most 2048 reads every read is sampled -- the majority of the '=' / 'X' ACGT bases per column
    nuc = m.nuc_counts()
    g = got["guide"].reshape(-1, 8)
    assert len(g) == (W + 31) // 32
    for c in range(W):
        bit = lambda plane: int(g[c >> 5, plane] >> (c & 31)) & 1
        acgt = nuc[c, :4]
        if not (al.flags & abi.MSQ_READS_HAS_EXOTIC):
            assert bit(0) == int(acgt.sum() > 0), (name, c)
        if acgt.sum() > 0 and not exotic:
            assert bit(1) | (bit(2) << 1) == int(np.argmax(acgt)), (name, c)
    # counts: the aligned-set kernel (or, for sets with exotic symbols, the fallback) against the oracle
    mode = abi.MSQ_PILEUP_AUTO if al.flags & abi.MSQ_READS_HAS_EXOTIC else abi.MSQ_PILEUP_ROWS
    n_, c_ = counts(ctx, al, NONE, b0, W, mode)
    assert np.array_equal(n_, nuc) and np.array_equal(c_, m.codon_counts()), name
    # ... and every other kernel takes the aligned set as a plain-layout set
    for md in (abi.MSQ_PILEUP_SIMPLE,):
        n2, c2 = counts(ctx, al, NONE, b0, W, md)
        assert np.array_equal(n2, nuc) and np.array_equal(c2, m.codon_counts())
    # a window 64 columns wider on the left and 40 on the right: still on the word grid
    if mode == abi.MSQ_PILEUP_ROWS:
        n3, c3 = counts(ctx, al, NONE, b0 - 64, W + 104, abi.MSQ_PILEUP_ROWS)
        assert np.array_equal(n3[64:64 + W], nuc) and not n3[:64].any() and not n3[64 + W:].any()
        want = m.codon_counts().copy()
        # the codon that starts at the reads' first column is counted once the window begins further left (CodonAt's i > 0)
        assert np.array_equal(c3[65:64 + W - 2], want[1:W - 2]) and not c3[:64].any()
    al.close()


def test_off_grid_windows_and_missing_guides_fall_back(ctx):
    reads = cases.random_reads(21, 200, 500, p_ins=0.04)
    m = orc.Msa(reads)
    src = device.upload(device.pack_host(reads), ctx.device, ctx=ctx)
    b0, W = m.begin1 - 1, m.W
    al = device.align(ctx, src, NONE, b0 - 5)  # the window begins 5 columns off the set's grid
    L = abi.lib()
    nuc, codon = device.alloc_counts(ctx, W)
    assert L.msq_pileup(ctx.h, al.view, NONE, b0, W, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_ROWS) == abi.MSQ_ERR_ARG
    n_, c_ = counts(ctx, al, NONE, b0, W, abi.MSQ_PILEUP_AUTO)
    assert np.array_equal(n_, m.nuc_counts()) and np.array_equal(c_, m.codon_counts())
    n_, c_ = counts(ctx, al, NONE, b0 - 5, W + 5, abi.MSQ_PILEUP_ROWS)
    assert np.array_equal(n_[5:], m.nuc_counts())
    # a view without the guide: the read-space kernel
    v = abi.PackedReads.from_buffer_copy(al.view)
    v.guide, v.guide_words = None, 0
    assert L.msq_pileup(ctx.h, v, NONE, b0 - 5, W + 5, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_ROWS) == abi.MSQ_ERR_ARG
    # an origin right of a read is refused
    h = C.c_void_p()
    assert L.msq_reads_align(ctx.h, src.view, NONE, b0 + 1, C.byref(h)) == abi.MSQ_ERR_ARG
    # a read outside the window: the deferred window error, as from the other kernels
    nuc, codon = device.alloc_counts(ctx, W)
    abi.check(L.msq_pileup(ctx.h, al.view, NONE, b0 - 5 + 32, W - 27, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_ROWS))
    assert L.msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    abi.check(L.msq_pileup(ctx.h, al.view, NONE, b0 - 5, W - 40, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_ROWS))
    assert L.msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    al.close()
    # no reads at all
    empty = device.upload(device.pack_host(cases.make_reads([])), ctx.device)
    e = device.align(ctx, empty, NONE, 0)
    assert e.n_reads == 0 and e.n_groups == 0
    e.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_sets(ctx, seed):
    rng = np.random.RandomState(100 + seed)
    n, W = int(rng.randint(1, 900)), int(rng.randint(40, 3900))
    reads = cases.random_reads(200 + seed, n, W, p_ins=float(rng.choice([0.0, 0.02, 0.3])), p_sub=float(rng.choice([0.0, 0.02, 0.2])),
                               p_del=float(rng.choice([0.0, 0.03, 0.2])))
    src = device.upload(device.pack_host(reads), ctx.device, ctx=ctx)
    b, e = device.window_bounds(ctx, src)
    al = device.align(ctx, src, NONE, b)
    want = counts(ctx, src, NONE, b, e - b, abi.MSQ_PILEUP_SIMPLE)
    got = counts(ctx, al, NONE, b, e - b, abi.MSQ_PILEUP_ROWS)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    al.close()


@pytest.mark.parametrize("kw,thr", [
    (dict(), None),
    (dict(n_minors=2, minor_frac=0.10, minor_codons=45), None),
    (dict(n_minors=1, minor_frac=0.30, minor_codons=300), None),
    (dict(sub=0.03), None),                       # queue drains in mid-batch, everything down the atomic path
    (dict(n_minors=5, minor_frac=0.15, minor_codons=60), None),
    (dict(rich_qv=True), (70, -1, 42, -1)),       # thresholds baked while aligning: the kernel with an N counter
])
def test_synthetic_samples_step_and_noalloc(ctx, kw, thr):
    n = 30000
    wl = synth.workload("pol_100k", n_reads=n, **kw)
    src = device.synth_device(ctx, wl, 0, n)
    t = abi.QvThresholds.make(*thr) if thr else NONE
    al = device.align(ctx, src, t, wl.win_begin0)
    assert bool(al.flags & abi.MSQ_READS_QV_BAKED) == (thr is not None)
    W, b0 = wl.W, wl.win_begin0
    want_n, want_c = device.pileup(ctx, src, t, b0, W, mode=abi.MSQ_PILEUP_STREAM)
    got_n, got_c = device.pileup(ctx, al, NONE, b0, W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(got_n, want_n) and torch.equal(got_c, want_c)
    L = abi.lib()
    # the fused step: bounds of the reads + matrices STORED (garbage before), twice in a row
    buf = torch.full((W * 70,), 12345, dtype=torch.int32, device="cuda")
    bounds = torch.zeros(2, dtype=torch.int32, device="cuda")
    for _ in range(2):
        abi.check(L.msq_pileup_step(ctx.h, al.view, NONE, b0, W, bounds.data_ptr(), buf.data_ptr()))
        abi.check(L.msq_pileup_check(ctx.h))
        assert bounds.tolist() == [-b0, b0 + W]
        assert torch.equal(buf[:W * 6], want_n) and torch.equal(buf[W * 6:], want_c)
    # the no-allocation door, accumulating on top of the first result
    nuc, codon = got_n.clone(), got_c.clone()
    abi.check(L.msq_pileup_noalloc(ctx.h, al.view, NONE, b0, W, nuc.data_ptr(), codon.data_ptr()))
    abi.check(L.msq_pileup_check(ctx.h))
    assert torch.equal(nuc, 2 * want_n) and torch.equal(codon, 2 * want_c)
    al.close()


def test_more_reads_than_one_launch_counts(ctx):
    """A wave's bit-sliced counters hold 1016 reads: 2.2 million reads take two launches."""
    wl = synth.workload("pol_100k", n_reads=2200000)
    src = device.synth_device(ctx, wl, 0, wl.n_reads)
    al = device.align(ctx, src, NONE, wl.win_begin0)
    want = device.pileup(ctx, src, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_STREAM)
    got = device.pileup(ctx, al, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert int(got[0].sum()) == wl.n_reads * wl.W
    al.close()


@pytest.mark.parametrize("name,n", [("hiv_mix_200k", 3000), ("amp20k_10m", 600)])
def test_wide_windows_go_slab_by_slab(ctx, name, n):
    """Windows of more than 126 words (9719 and 20000 columns): one launch per slab of <= 126 words, each loading only
    its words of every read; the oracle on the same reads, and the read-space kernels."""
    wl = synth.workload(name, n_reads=n)
    src = device.synth_device(ctx, wl, 0, n)
    al = device.align(ctx, src, NONE, wl.win_begin0)
    m = orc.Msa(wl.host_reads(0, n))
    W, b0 = wl.W, wl.win_begin0
    assert (m.begin1 - 1, m.W) == (b0, W)
    got = counts(ctx, al, NONE, b0, W, abi.MSQ_PILEUP_ROWS)
    assert np.array_equal(got[0], m.nuc_counts()) and np.array_equal(got[1], m.codon_counts())
    # staggered reads cut by a window that starts 320 columns further right: the deferred window error
    nuc, codon = device.alloc_counts(ctx, W)
    abi.check(abi.lib().msq_pileup(ctx.h, al.view, NONE, b0 + 320, W - 320, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_ROWS))
    assert abi.lib().msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    al.close()


def test_wide_window_large_sample(ctx):
    wl = synth.workload("hiv_mix_200k", n_reads=40000, minor_frac=0.05, sub=0.002)
    src = device.synth_device(ctx, wl, 0, wl.n_reads)
    al = device.align(ctx, src, NONE, wl.win_begin0)
    want = device.pileup(ctx, src, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_STREAM)
    got = device.pileup(ctx, al, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    al.close()


def test_noalloc_on_an_aligned_set_too_wide_for_its_built_in_partials(ctx):
    """4000 columns = 125 words: the aligned-set kernel's partials (9 rows per workgroup) exceed msq_pileup_noalloc's
    built-in 32 MiB, the read-space kernel's (7 rows) fit -- the call is served by that kernel, allocates nothing and
    gives the same counts."""
    reads = cases.random_reads(77, 700, 4000, p_ins=0.01, p_sub=0.01, p_del=0.01)
    src = device.upload(device.pack_host(reads), ctx.device, ctx=ctx)
    b, e = device.window_bounds(ctx, src)
    W = e - b
    assert 3900 < W <= 4096
    al = device.align(ctx, src, NONE, b)
    want = device.pileup(ctx, al, NONE, b, W, mode=abi.MSQ_PILEUP_ROWS)
    L = abi.lib()
    nuc, codon = device.alloc_counts(ctx, W)
    abi.check(L.msq_pileup_noalloc(ctx.h, al.view, NONE, b, W, nuc.data_ptr(), codon.data_ptr()))  # loads the code
    held = L.msq_debug_ctx_device_bytes(ctx.h)
    free = torch.cuda.mem_get_info()[0]
    nuc.zero_()
    codon.zero_()
    abi.check(L.msq_pileup_noalloc(ctx.h, al.view, NONE, b, W, nuc.data_ptr(), codon.data_ptr()))
    abi.check(L.msq_pileup_check(ctx.h))
    assert L.msq_debug_ctx_device_bytes(ctx.h) == held and torch.cuda.mem_get_info()[0] == free
    assert torch.equal(nuc, want[0]) and torch.equal(codon, want[1])
    al.close()


@pytest.mark.parametrize("kw,min_groups", [
    (dict(n_minors=2, minor_frac=0.03, minor_codons=300), 3),   # two clones whose variant codons overlap and touch
    (dict(n_minors=4, minor_frac=0.2, minor_codons=100), 5),    # five strains at 20 % each
    (dict(n_minors=4, minor_frac=0.01, minor_codons=100), 2),   # small clones: some may stay below the sample's support
    (dict(n_minors=7, minor_frac=0.1, minor_codons=30), 8),     # as many groups as the set holds
])
def test_haplotype_groups(ctx, kw, min_groups):
    """Samples of several clones: msq_reads_align groups the reads by haplotype (every read in exactly one group, the list
    ascending inside a group, a guide per group that is the group's own majority), the kernel counts every group against
    its own guide -- same matrices as the simple kernel, through msq_pileup, the fused step and with the grouping
    switched off."""
    n = 60000
    wl = synth.workload("pol_100k", n_reads=n, **kw)
    src = device.synth_device(ctx, wl, 0, n)
    W, b0 = wl.W, wl.win_begin0
    al = device.align(ctx, src, NONE, b0)
    got = al.download()
    K = got["hap_count"]
    assert min_groups <= K <= 8
    sizes = got["hap_reads"][:K]
    assert sum(sizes) == n and not any(got["hap_reads"][K:])
    perm = got["hap_perm"]
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
    lo = 0
    for h in range(K):
        part = perm[lo:lo + sizes[h]]
        assert (np.diff(part.astype(np.int64)) > 0).all()
        lo += sizes[h]
    assert sizes[0] >= 0.9 * max(sizes)  # ordered by their size in the SAMPLE
    want = device.pileup(ctx, src, NONE, b0, W, mode=abi.MSQ_PILEUP_SIMPLE)
    res = device.pileup(ctx, al, NONE, b0, W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(res[0], want[0]) and torch.equal(res[1], want[1])
    buf = torch.full((W * 70,), 777, dtype=torch.int32, device="cuda")
    bounds = torch.zeros(2, dtype=torch.int32, device="cuda")
    L = abi.lib()
    abi.check(L.msq_pileup_step(ctx.h, al.view, NONE, b0, W, bounds.data_ptr(), buf.data_ptr()))
    abi.check(L.msq_pileup_check(ctx.h))
    assert bounds.tolist() == [-b0, b0 + W] and torch.equal(buf[:W * 6], want[0]) and torch.equal(buf[W * 6:], want[1])
    # a group's guide is the majority of the group's own reads: group 0 of the five strains differs from the set's guide
    # nowhere but at variant columns, and every group's guide covers the window
    g = got["hap_guides"].reshape(K, -1, 8)
    assert (g[:, :W // 32, 0] == 0xffffffff).all()
    # grouping switched off: one population, same counts
    abi.check(L.msq_debug_tune(ctx.h, 3, 1))
    try:
        flat = device.align(ctx, src, NONE, b0)
    finally:
        abi.check(L.msq_debug_tune(ctx.h, 3, 0))
    assert flat.download()["hap_count"] == 0
    res = device.pileup(ctx, flat, NONE, b0, W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(res[0], want[0]) and torch.equal(res[1], want[1])
    flat.close()
    al.close()


def test_haplotype_groups_over_several_launches_and_slabs(ctx):
    """2.2 million reads of three clones take two launches (each group's reads in two ranges); a wide window goes slab by
    slab with the same groups."""
    wl = synth.workload("pol_100k", n_reads=2200000, n_minors=2, minor_frac=0.2, minor_codons=60)
    src = device.synth_device(ctx, wl, 0, wl.n_reads)
    al = device.align(ctx, src, NONE, wl.win_begin0)
    assert al.download()["hap_count"] == 3
    want = device.pileup(ctx, src, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_STREAM)
    got = device.pileup(ctx, al, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    al.close()
    del src
    wl = synth.workload("hiv_mix_200k", n_reads=30000, n_minors=3, minor_frac=0.15, minor_codons=200)
    src = device.synth_device(ctx, wl, 0, wl.n_reads)
    al = device.align(ctx, src, NONE, wl.win_begin0)
    assert al.download()["hap_count"] >= 2
    want = device.pileup(ctx, src, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_SIMPLE)
    got = device.pileup(ctx, al, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    al.close()
