"""GPU parity at the sizes the bench lines are quoted on (BASELINE.json configs #3, #4, #5): the streaming kernel
in the regime only those sizes reach -- hundreds of reads per wave, the upper levels of the bit-sliced counters,
the split last pass, several batches per wave and column slab -- against the independent simple kernel, against
the oracle through a tie-in that needs the oracle on a prefix only,

    counts(all N reads) - counts(reads k .. N) == oracle(first k reads),

and through the size-independent properties of a pileup; then the testing pass on the real count matrix at that
coverage, every cell against the oracle's Fisher."""
import numpy as np
import pytest
import torch

from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
P_RTOL, P_ATOL = 1e-9, 1e-300


def as_np(nuc, codon, W):
    return (nuc.cpu().numpy()[:W * 6].reshape(W, 6).astype(np.int64), codon.cpu().numpy()[:W * 64].reshape(W, 64).astype(np.int64))


def check_properties(N, Cc, n_reads, W, full_span=True):
    if full_span:
        assert (N.sum(1) == n_reads).all()  # every read spans the amplicon: coverage incl. '-' and N
    assert Cc[0].sum() == 0 and Cc[W - 2:].sum() == 0  # MSARow::CodonAt's i > 0 rule, CodingCodonAt's bound
    acgt = N[:, :4].sum(1)
    assert (Cc.sum(1) <= np.minimum(acgt, np.minimum(np.roll(acgt, -1), np.roll(acgt, -2)))).all()
    assert (Cc.reshape(W, 4, 16).sum(2) <= N[:, :4]).all()  # first codon base marginalises to at most the nucleotide count


def oracle_tie_in(ctx, wl, thr, k, total, whole, pileup_rest):
    """counts(total) - counts(reads k..total) == oracle(first k)."""
    W = wl.W
    rest = device.synth_device(ctx, wl, k, total - k)
    rn, rc = as_np(*pileup_rest(rest), W)
    del rest
    m = orc.Msa(wl.host_reads(0, k))
    assert (whole[0] - rn == m.nuc_counts()).all()
    assert (whole[1] - rc == m.codon_counts()).all()
    return m


def check_cells(ctx, wl, codon, Cc, n_reads, b0):
    """msq_call_cells on the device matrix; every evaluated cell against the oracle's fisher_exact_tiss."""
    gb, ge = wl.gene_arrays()
    plan = device.Plan(ctx, gb, ge, wl.reference if wl.use_reference else "", b0 + 1, wl.W)
    ctx.lfact_reserve(2 * n_reads + 2)
    device.call_cells(ctx, plan, codon)
    pos, p, expected, nt = device.results(plan)
    em = np.zeros(4)
    orc.lib().orc_error_estimates_default(em.ctypes.data)
    match, sub = em[0], em[1]
    want_nt, checked = 0, 0
    for k, row in enumerate(pos):
        c, cov, ref, skipped = int(row[3]), int(row[4]), int(row[5]), int(row[9])
        if not (1 <= c < wl.W - 2):
            assert cov == 0
            continue
        want_nt += int((Cc[c] > 0).sum())
        assert cov == Cc[c].sum()
        if skipped:
            continue
        for cod in np.nonzero(Cc[c])[0]:
            if cod == ref or cod == int(row[6]):
                assert p[k, cod] < 0
                continue
            pr = 1.0
            for sh in (4, 2, 0):
                pr *= sub if ((ref >> sh) & 3) != ((cod >> sh) & 3) else match
            e = cov * pr
            assert expected[k, cod] == e
            checked += 1
    assert nt == max(want_nt, 1)
    # the Fisher sums, all cells: oracle p * numberOfTests clipped (src/AminoAcidCaller.cpp:481-488)
    for k, row in enumerate(pos):
        c, cov = int(row[3]), int(row[4])
        for cod in np.nonzero(p[k] >= 0)[0]:
            cnt, e = int(Cc[c, cod]), expected[k, cod]
            want = min(1.0, orc.fisher(cnt, cov - cnt, int(np.ceil(e)), int(np.ceil(cov - e))) * nt)
            assert abs(p[k, cod] - want) <= P_RTOL * abs(want) + P_ATOL, (c, cod, cnt, p[k, cod], want)
    plan.close()
    return checked


def test_pol_1m_headline_size(ctx):
    """BASELINE config #3 on one GPU, the bench's default line: 1e6 reads x 2843 columns, ~488 reads per wave."""
    wl = synth.workload("pol_1m")
    n, W, b0 = wl.n_reads, wl.W, wl.win_begin0
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    assert device.window_bounds(ctx, dr) == (b0, b0 + W)
    nuc, codon = device.pileup(ctx, dr, thr, b0, W)  # MSQ_PILEUP_AUTO: the streaming kernel
    nuc_s, codon_s = device.pileup(ctx, dr, thr, b0, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(nuc, nuc_s) and torch.equal(codon, codon_s)
    del nuc_s, codon_s
    # the path bench.py times: window left in HBM
    bounds = device.window_bounds_device(ctx, dr)
    nuc_w, codon_w = device.pileup_device_window(ctx, dr, thr, bounds, b0, W)
    assert torch.equal(nuc, nuc_w) and torch.equal(codon, codon_w)
    del nuc_w, codon_w, dr
    N, Cc = as_np(nuc, codon, W)
    check_properties(N, Cc, n, W)
    assert (Cc[1:W - 2].sum(1) > 0.97 * n).all()
    oracle_tie_in(ctx, wl, thr, 2000, n, (N, Cc), lambda rest: device.pileup(ctx, rest, thr, b0, W))
    assert check_cells(ctx, wl, codon, Cc, n, b0) > 5000


def test_pol_1m_baked_thresholds(ctx):
    """The same size through the variant with an N counter (thresholds baked at pack time): source set + thresholds
    == baked set, stream == simple; oracle tie-in under the mask."""
    wl = synth.workload("pol_1m", n_reads=400000)
    n, W, b0 = wl.n_reads, wl.W, wl.win_begin0
    thr = abi.QvThresholds.make(qual=70)
    dr = device.synth_device(ctx, wl, 0, n)
    baked = device.bake_qv(ctx, dr, thr)
    nuc, codon = device.pileup(ctx, baked, thr, b0, W)
    nuc_u, codon_u = device.pileup(ctx, dr, thr, b0, W)  # unbaked: the QualQV plane read in the pass
    assert torch.equal(nuc, nuc_u) and torch.equal(codon, codon_u)
    nuc_s, codon_s = device.pileup(ctx, dr, thr, b0, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(nuc, nuc_s) and torch.equal(codon, codon_s)
    del dr, baked, nuc_u, codon_u, nuc_s, codon_s
    N, Cc = as_np(nuc, codon, W)
    check_properties(N, Cc, n, W)
    assert N[:, 5].sum() > 0.1 * n * W  # the mask bites: QualQV ~ N(80, 10) against 70
    k = 1500
    rest = device.synth_device(ctx, wl, k, n - k)
    rn, rc = as_np(*device.pileup(ctx, device.bake_qv(ctx, rest, thr), thr, b0, W), W)
    m = orc.Msa(wl.host_reads(0, k), (70, -1, -1, -1))
    assert (N - rn == m.nuc_counts()).all() and (Cc - rc == m.codon_counts()).all()


def test_hiv_mix_200k_full_size(ctx):
    """BASELINE config #4: 2e5 reads x 9719 columns (3 column slabs), 19 genes in 3 frames, 0.1 % minors.
    Partition == one read set == simple kernel; the 19-gene plan and numberOfTests against the oracle on a prefix."""
    wl = synth.workload("hiv_mix_200k")
    n, W, b0 = wl.n_reads, wl.W, wl.win_begin0
    assert (n, W, len(wl.genes)) == (200000, 9719, 19)
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    assert device.window_bounds(ctx, dr) == (b0, b0 + W)
    nuc, codon = device.pileup(ctx, dr, thr, b0, W)  # three slab launches over the whole set
    part = device.Partition(ctx, dr, b0, W)
    assert part.n_slabs == 3
    nuc_p, codon_p = part.pileup(thr)
    assert torch.equal(nuc, nuc_p) and torch.equal(codon, codon_p)
    part.close()
    nuc_s, codon_s = device.pileup(ctx, dr, thr, b0, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(nuc, nuc_s) and torch.equal(codon, codon_s)
    del dr, nuc_p, codon_p, nuc_s, codon_s
    N, Cc = as_np(nuc, codon, W)
    check_properties(N, Cc, n, W)

    def rest_through_partition(rest):
        p = device.Partition(ctx, rest, b0, W)
        out = p.pileup(thr)
        p.close()
        return out
    k = 2000
    m = oracle_tie_in(ctx, wl, thr, k, n, (N, Cc), rest_through_partition)
    # the testing pass over the 19-gene plan at full coverage, cells against the oracle's Fisher
    assert check_cells(ctx, wl, codon, Cc, n, b0) > 10000
    # ... and the whole CallVariants cell loop against the oracle on the prefix (positions, refs, numberOfTests)
    gb, ge = wl.gene_arrays()
    pre = device.synth_device(ctx, wl, 0, k)
    pn, pc = device.pileup(ctx, pre, thr, b0, W)
    plan = device.Plan(ctx, gb, ge, wl.reference, b0 + 1, W)
    device.call_cells(ctx, plan, pc)
    pos, p, expected, nt = device.results(plan)
    o_pos, o_cells, o_nt = m.call(gb, ge, wl.reference)
    assert nt == o_nt and (pos == o_pos).all()
    tested = np.argwhere(p >= 0)
    assert len(tested) == len(o_cells) > 1000
    for (kk, cod), oc in zip(tested, o_cells):
        assert (kk, cod) == (oc.position, oc.codon) and expected[kk, cod] == oc.expected
        assert abs(p[kk, cod] - oc.p) <= P_RTOL * abs(oc.p) + P_ATOL
    plan.close()


def test_amp20k_many_batches_per_slab(ctx):
    """BASELINE config #5's shape at 1e5 reads: 20000 columns = 5 slabs, ~6 wave batches per wave and slab through
    the partition; no configuration (one ORF over the window, majority codon as reference)."""
    wl = synth.workload("amp20k_10m", n_reads=100000)
    n, W, b0 = wl.n_reads, wl.W, wl.win_begin0
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    part = device.Partition(ctx, dr, b0, W)
    assert part.n_slabs == 5
    nuc, codon = part.pileup(thr)
    part.close()
    nuc_1, codon_1 = device.pileup(ctx, dr, thr, b0, W)
    assert torch.equal(nuc, nuc_1) and torch.equal(codon, codon_1)
    del dr, nuc_1, codon_1
    N, Cc = as_np(nuc, codon, W)
    check_properties(N, Cc, n, W)

    def rest_through_partition(rest):
        p = device.Partition(ctx, rest, b0, W)
        out = p.pileup(thr)
        p.close()
        return out
    oracle_tie_in(ctx, wl, thr, 400, n, (N, Cc), rest_through_partition)
    assert check_cells(ctx, wl, codon, Cc, n, b0) > 20000


def test_amp20k_one_gpu_shard_of_config_5(ctx):
    """BASELINE config #5 as one GPU of eight holds it: 1.25e6 reads x 20000 columns (2.5e10 bases).  The column-slab
    partition in merged launches (the reads of a slab exceed what its workgroups' counters hold: two rounds), the
    aligned set slab by slab, and the simple kernel give the same matrices; properties; the oracle on a 200-read prefix.
    Skipped when the card has not 100 GB free."""
    free, _ = torch.cuda.mem_get_info()
    if free < 100 * 2**30:
        pytest.skip("config #5's one-GPU shard needs ~100 GB of free HBM, this card has %.0f GB" % (free / 2**30))
    wl = synth.workload("amp20k_10m", n_reads=1250000)
    n, W, b0 = wl.n_reads, wl.W, wl.win_begin0
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    assert device.window_bounds(ctx, dr) == (b0, b0 + W)
    part = device.Partition(ctx, dr, b0, W)
    assert part.n_slabs == 5
    before = abi.lib().msq_debug_merged_launches()
    nuc, codon = part.pileup(thr)
    assert abi.lib().msq_debug_merged_launches() - before >= 1  # the merged path took it (two rounds: 1.25e6 reads per slab)
    part.close()
    al = device.align(ctx, dr, thr, b0)
    nuc_a, codon_a = device.pileup(ctx, al, thr, b0, W, mode=abi.MSQ_PILEUP_ROWS)
    assert torch.equal(nuc, nuc_a) and torch.equal(codon, codon_a)
    del nuc_a, codon_a
    nuc_s, codon_s = device.pileup(ctx, dr, thr, b0, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(nuc, nuc_s) and torch.equal(codon, codon_s)
    del dr, nuc_s, codon_s
    N, Cc = as_np(nuc, codon, W)
    check_properties(N, Cc, n, W)
    assert (Cc[1:W - 2].sum(1) > 0.97 * n).all()
    # counts(all) - counts(reads 200 ..) == oracle(first 200); the rest through the aligned set's kernel
    k = 200
    rest_al = device.align(ctx, device.synth_device(ctx, wl, k, n - k), thr, b0)
    rn, rc = as_np(*device.pileup(ctx, rest_al, thr, b0, W, mode=abi.MSQ_PILEUP_ROWS), W)
    m = orc.Msa(wl.host_reads(0, k))
    assert (N - rn == m.nuc_counts()).all() and (Cc - rc == m.codon_counts()).all()
    # additivity over the shard rule: the halves of this shard add up to it
    al.close()
    rest_al.close()
    half = n // 2
    h0 = device.align(ctx, device.synth_device(ctx, wl, 0, half), thr, b0)
    n0, c0 = device.pileup(ctx, h0, thr, b0, W)
    h1 = device.align(ctx, device.synth_device(ctx, wl, half, n - half), thr, b0)
    device.pileup(ctx, h1, thr, b0, W, nuc=n0, codon=c0)
    assert torch.equal(nuc, n0) and torch.equal(codon, c0)
