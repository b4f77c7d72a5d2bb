"""msq_reads_bake_qv: QV thresholds applied at pack time.  A baked set (failing match bases read 'N', no QV planes)
must give, with any thresholds, exactly what the source set gives under the baked thresholds -- counts, phasing
keys, partitioned pileup -- and what the oracle's MSAByRow::AddRead restatement gives (src/MSA.cpp:163-166).
Parity unpinned beyond the oracle: the reference holds no fixture for the pileup."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
EDGE = cases.edge_sets()
MODES = [abi.MSQ_PILEUP_SIMPLE, abi.MSQ_PILEUP_AUTO]
NONE = abi.QvThresholds.make(-1, -1, -1, -1)


def case(name):
    if name == "random_rich":
        return cases.random_reads(61, 200, 2300, rich=True, p_ins=0.05, p_del=0.03), (30, 20, 42, 25)
    if name == "random_qual":
        return cases.random_reads(62, 300, 700, rich=False, p_ins=0.05), (40, -1, -1, -1)
    if name == "wide_rich":
        return cases.random_reads(63, 80, 5000, rich=True, p_ins=0.02), (35, -1, 42, -1)
    if name == "high_thr":
        return cases.random_reads(64, 100, 400, rich=True), (200, 129, 128, 255)
    return EDGE[name]


@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("name", sorted(EDGE) + ["random_rich", "random_qual", "wide_rich", "high_thr"])
def test_baked_counts_equal_thresholded_counts(ctx, name, plain):
    reads, thr = case(name)
    m = orc.Msa(reads, thr)
    q = abi.QvThresholds.make(*thr[:4])
    dr = device.upload(cases.pack_plain(reads) if plain else device.pack_host(reads), ctx.device)
    baked = device.bake_qv(ctx, dr, q)
    v = baked.view
    assert not (v.qual or v.delqv or v.subqv or v.insqv)
    any_plane = any(t >= 0 and dr.t.get(f) is not None for t, f in zip(thr[:4], ("qual", "delqv", "subqv", "insqv")))
    assert bool(baked.flags & abi.MSQ_READS_QV_BAKED) == any_plane
    assert baked.flags & ~abi.MSQ_READS_QV_BAKED == dr.flags
    assert device.window_bounds(ctx, baked) == device.window_bounds(ctx, dr)
    b, e = device.window_bounds(ctx, dr)
    W = e - b if dr.n_reads else 0
    for mode in MODES:
        for t in (NONE, q):  # thresholds on a baked set have nothing left to mask
            nuc, codon = device.pileup(ctx, baked, t, b, W, mode=mode)
            assert (nuc.cpu().numpy()[:W * 6].reshape(W, 6) == m.nuc_counts()).all()
            assert (codon.cpu().numpy()[:W * 64].reshape(W, 64) == m.codon_counts()).all()
    # the baked planes themselves: only match bases that fail turned into literal N, everything else untouched
    src, got = device.download(dr), baked.download()
    for f in ("ref_start", "ref_end", "group_off"):
        assert (src[f][:len(got[f])] == got[f]).all()
    if dr.n_groups:
        s, g = src["bop"][:4 * dr.n_groups].reshape(-1, 4), got["bop"].reshape(-1, 4)
        turned = (s[:, 0] ^ g[:, 0]) | (s[:, 1] ^ g[:, 1]) | (s[:, 2] ^ g[:, 2])
        may = ~s[:, 3] & (~s[:, 2] | s[:, 1])                      # op 0, or op 1 with an exotic code
        assert ((turned & ~may) == 0).all()
        assert (g[:, 3] == s[:, 3]).all()                          # never the column-consuming bit
        assert ((g[:, 0] & turned) == turned).all() and ((g[:, 1] & turned) == 0).all()  # code 1 ..
        assert ((g[:, 2] & turned) == turned).all()                # .. under op 1
    baked.close()


@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("name", sorted(EDGE) + ["random_rich", "random_qual", "wide_rich", "high_thr", "some_absent"])
def test_packer_applies_thresholds_like_pack_then_bake(ctx, monkeypatch, name, plain):
    """msq_pack_reads_device_baked == msq_pack_reads_device + msq_reads_bake_qv, byte for byte: tables, bit planes,
    flags, no QV planes -- with one chunk and with many, with QVs present on some bases only."""
    has_qual = has_rich = None
    if name == "some_absent":
        reads, thr = cases.random_reads(65, 150, 600, rich=True, p_ins=0.04), (50, 30, 42, 35)
        rng = np.random.RandomState(65)
        total = int(reads["base_off"][-1])
        has_qual, has_rich = (rng.rand(total) < 0.7).astype(np.uint8), (rng.rand(total) < 0.5).astype(np.uint8)
    else:
        reads, thr = case(name)
    q = abi.QvThresholds.make(*thr[:4])
    src = device.upload(device.pack_device(ctx, reads, has_qual=has_qual, has_rich=has_rich, plain=plain), ctx.device)
    want_set = device.bake_qv(ctx, src, q)
    want = want_set.download()
    for chunk_bases, threads, wide in ((0, 0, False), (700, 3, False), (0, 0, True), (900, 2, True)):
        # wide: the 8-byte records staged as they are and translated on the device; else one translated byte per base
        if wide:
            monkeypatch.setenv("MSQ_PACK_WIDE_STAGING", "1")
        else:
            monkeypatch.delenv("MSQ_PACK_WIDE_STAGING", raising=False)
        got = device.pack_device(ctx, reads, has_qual=has_qual, has_rich=has_rich, plain=plain, thr=q, chunk_bases=chunk_bases, threads=threads)
        assert got["flags"] == want["flags"], (got["flags"], want["flags"])
        assert got["n_groups"] == want["n_groups"]
        for f in ("qual", "delqv", "subqv", "insqv"):
            assert got[f] is None
        for f in ("ref_start", "ref_end", "group_off", "bop"):
            assert (np.asarray(got[f]) == np.asarray(want[f])).all(), f
    want_set.close()
    monkeypatch.delenv("MSQ_PACK_WIDE_STAGING", raising=False)
    with pytest.raises(abi.MsqError) as err:
        device.pack_device(ctx, reads, thr=abi.QvThresholds.make(qual=256))
    assert err.value.code == abi.MSQ_ERR_ARG


@pytest.mark.parametrize("seed,W,thr", [(71, 2500, (40, -1, 42, -1)), (72, 5000, (35, 20, -1, 30))])
def test_baked_phase_keys_and_partition(ctx, seed, W, thr):
    reads = cases.random_reads(seed, 200, W, rich=True, p_ins=0.03, p_del=0.03)
    q = abi.QvThresholds.make(*thr)
    dr = device.upload(device.pack_host(reads), ctx.device)
    baked = device.bake_qv(ctx, dr, q)
    b, e = device.window_bounds(ctx, dr)
    n = reads["n_reads"]
    rng = np.random.RandomState(seed)
    pos = np.array(sorted(set([-1, 0, 1, W - 3, W - 1] + list(rng.randint(1, W - 3, size=60)))), dtype=np.int32)
    keys = [torch.zeros(n * len(pos), dtype=torch.int16, device="cuda") for _ in range(2)]
    abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, q, b, e - b, len(pos), abi.ptr(pos), keys[0].data_ptr()))
    abi.check(abi.lib().msq_phase_codons(ctx.h, baked.view, NONE, b, e - b, len(pos), abi.ptr(pos), keys[1].data_ptr()))
    assert torch.equal(keys[0], keys[1])
    want_nuc, want_codon = device.pileup(ctx, dr, q, b, e - b)
    part = device.Partition(ctx, baked, b, e - b, 1024)
    nuc, codon = part.pileup(NONE)
    assert torch.equal(nuc, want_nuc) and torch.equal(codon, want_codon)
    part.close()
    baked.close()


@pytest.mark.parametrize("n", [20000, 1000000])
def test_baked_synthetic_rich_set(ctx, n):
    """The bench's QV workload shape (up to its full size): device-generated rich reads, all four thresholds."""
    wl = synth.workload("pol_100k", n_reads=n, rich_qv=True)
    dr = device.synth_device(ctx, wl)
    q = abi.QvThresholds.make(60, 50, 42, 55)
    baked = device.bake_qv(ctx, dr, q)
    b, e = device.window_bounds(ctx, dr)
    want_nuc, want_codon = device.pileup(ctx, dr, q, b, e - b)
    nuc, codon = device.pileup(ctx, baked, NONE, b, e - b)
    assert torch.equal(nuc, want_nuc) and torch.equal(codon, want_codon)
    assert int(nuc.view(-1, 6)[:, 5].sum()) > 0  # the thresholds do mask bases here
    assert baked.n_bytes() < dr.n_bytes() // 4
    baked.close()


def test_bake_errors(ctx):
    reads, _ = EDGE[sorted(EDGE)[0]]
    dr = device.upload(device.pack_host(reads), ctx.device)
    with pytest.raises(abi.MsqError) as err:
        device.bake_qv(ctx, dr, abi.QvThresholds.make(256, -1, -1, -1))
    assert err.value.code == abi.MSQ_ERR_ARG
