"""GPU parity, pileup: msq_pileup (through the C ABI) against the oracle -- count matrices bit-exact -- on the
hand-made edge sets, on synthetic workloads, in every kernel mode; plus size-independent properties at the
bench size.  The device generator must reproduce the host generator bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
EDGE = cases.edge_sets()
MODES = [abi.MSQ_PILEUP_SIMPLE, abi.MSQ_PILEUP_AUTO]


def gpu_counts(ctx, reads, thr, mode, begin0=None, W=None):
    dr = device.upload(device.pack_host(reads), ctx.device)
    b, e = device.window_bounds(ctx, dr)
    if begin0 is None:
        begin0, W = b, (e - b if dr.n_reads else 0)
    nuc, codon = device.pileup(ctx, dr, abi.QvThresholds.make(*thr), begin0, W, mode=mode)
    return (b, e), nuc.cpu().numpy()[:W * 6].reshape(W, 6), codon.cpu().numpy()[:W * 64].reshape(W, 64)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", sorted(EDGE))
def test_edge_sets(ctx, name, mode):
    reads, thr = EDGE[name]
    m = orc.Msa(reads, thr)
    (b, e), nuc, codon = gpu_counts(ctx, reads, (thr[0], thr[1], thr[2], thr[3]), mode)
    assert (b + 1, e + 1) == (m.begin1, m.end1)  # MSAByRow::BeginPos/EndPos are 1-based (src/MSA.cpp:107-108)
    assert (nuc == m.nuc_counts()).all()
    assert (codon == m.codon_counts()).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", sorted(EDGE) + ["random_indels", "random_rich"])
def test_plain_layout_reads(ctx, name, mode):
    """Packed reads in the plain group layout (no MSQ_READS_NORMALIZED; written by tests/cases.pack_plain from the
    header's text): the kernels compact the groups themselves and must give the same counts."""
    if name == "random_indels":
        reads, thr = cases.random_reads(11, 300, 700, p_ins=0.05), (-1, -1, -1, -1)
    elif name == "random_rich":
        reads, thr = cases.random_reads(12, 200, 2300, rich=True, p_ins=0.05), (30, 20, 42, 25)
    else:
        reads, thr = EDGE[name]
    m = orc.Msa(reads, thr)
    packed = cases.pack_plain(reads)
    assert not packed["flags"] & abi.MSQ_READS_NORMALIZED
    dr = device.upload(packed, ctx.device)
    b, e = device.window_bounds(ctx, dr)
    W = e - b if dr.n_reads else 0
    nuc, codon = device.pileup(ctx, dr, abi.QvThresholds.make(*thr), b, W, mode=mode)
    assert (nuc.cpu().numpy()[:W * 6].reshape(W, 6) == m.nuc_counts()).all()
    assert (codon.cpu().numpy()[:W * 64].reshape(W, 64) == m.codon_counts()).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,n,rich,thr", [
    ("pol_100k", 3000, False, (-1, -1, -1, -1)),
    ("pol_100k", 1500, False, (75, -1, -1, -1)),
    ("pol_100k", 1500, True, (-1, -1, 42, -1)),
    ("pol_100k", 1000, True, (70, 50, 42, 55)),
    ("hiv_1k", 400, False, (-1, -1, -1, -1)),
    ("amp20k_10m", 60, False, (-1, -1, -1, -1)),
])
def test_synthetic_vs_oracle(ctx, name, n, rich, thr, mode):
    wl = synth.workload(name, n_reads=n, rich_qv=rich)
    reads = wl.host_reads(0, n)
    m = orc.Msa(reads, thr)
    _, nuc, codon = gpu_counts(ctx, reads, thr, mode)
    assert (nuc == m.nuc_counts()).all()
    assert (codon == m.codon_counts()).all()


@pytest.mark.parametrize("name,rich", [("pol_100k", False), ("hiv_1k", True)])
def test_device_generator_matches_host_generator(ctx, name, rich):
    wl = synth.workload(name, n_reads=500, rich_qv=rich)
    host = device.pack_host(wl.host_reads(100, 300))
    dev = device.download(device.synth_device(ctx, wl, 100, 300))
    assert dev["n_groups"] == host["n_groups"]
    for f in device.DeviceReads.FIELDS:
        if host[f] is None:
            assert dev[f] is None or not rich
            continue
        assert (np.asarray(dev[f]) == np.asarray(host[f])).all(), f


def test_errors_surface(ctx):
    L = abi.lib()
    thr = abi.QvThresholds.make()
    # "Unexpected base" (src/MSA.cpp:85) is raised on the device path as a deferred error
    dr = device.upload(device.pack_host(cases.make_reads([(0, "===", "ARG")])), ctx.device)
    nuc, codon = device.alloc_counts(ctx, 3)
    abi.check(L.msq_pileup(ctx.h, dr.view, thr, 0, 3, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_AUTO))
    assert L.msq_pileup_check(ctx.h) == abi.MSQ_ERR_BASE
    assert L.msq_pileup_check(ctx.h) == abi.MSQ_OK  # cleared
    # ... but not when the QV mask turns the base into 'N' first (MSA.cpp:163-166)
    nuc.zero_(); codon.zero_()
    device.pileup(ctx, dr, abi.QvThresholds.make(qual=61), 0, 3, nuc, codon)
    assert nuc.cpu().numpy()[:18].reshape(3, 6)[1].tolist() == [0, 0, 0, 0, 0, 1]
    # a read outside the stated window
    dr2 = device.upload(device.pack_host(cases.make_reads([(5, "====", "ACGT")])), ctx.device)
    abi.check(L.msq_pileup(ctx.h, dr2.view, thr, 6, 3, nuc.data_ptr(), codon.data_ptr(), abi.MSQ_PILEUP_AUTO))
    assert L.msq_pileup_check(ctx.h) == abi.MSQ_ERR_WINDOW
    # empty input: bounds are the MSAByRow defaults (include/pacbio/data/MSA.h:83-84)
    e = device.upload(device.pack_host(cases.make_reads([])), ctx.device)
    assert device.window_bounds(ctx, e) == (2 ** 31 - 1, 0)


def test_bench_size_properties(ctx):
    """At BASELINE config #2's full size (1e5 x 2843): shard additivity, coverage sums, codon <= nucleotide
    coverage, quirk columns, and agreement of the kernel modes."""
    wl = synth.workload("pol_100k")
    n = wl.n_reads
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    assert device.window_bounds(ctx, dr) == (2252, 5095)
    W = wl.W
    nuc, codon = device.pileup(ctx, dr, thr, 2252, W)
    nuc_s, codon_s = device.pileup(ctx, dr, thr, 2252, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(nuc, nuc_s) and torch.equal(codon, codon_s)
    N = nuc.cpu().numpy()[:W * 6].reshape(W, 6).astype(np.int64)
    Cc = codon.cpu().numpy()[:W * 64].reshape(W, 64).astype(np.int64)
    assert (N.sum(1) == n).all()  # every read spans the amplicon: coverage incl. '-' and N
    assert Cc[0].sum() == 0 and Cc[W - 2:].sum() == 0 and (Cc[1:W - 2].sum(1) > 0.97 * n).all()
    acgt = N[:, :4].sum(1)
    assert (Cc.sum(1) <= np.minimum(acgt, np.minimum(np.roll(acgt, -1), np.roll(acgt, -2)))).all()
    # first base of the codon histogram marginalises to at most the nucleotide count
    assert (Cc.reshape(W, 4, 16).sum(2) <= N[:, :4]).all()
    # shards add up (the multi-GPU path is exactly this + an all-reduce)
    acc_n, acc_c = device.alloc_counts(ctx, W)
    for first, cnt in ((0, 30000), (30000, 50000), (80000, 20000)):
        part = device.synth_device(ctx, wl, first, cnt)
        device.pileup(ctx, part, thr, 2252, W, acc_n, acc_c)
    assert torch.equal(acc_n, nuc) and torch.equal(acc_c, codon)
    # the oracle on a prefix the CPU finishes in seconds
    k = 2000
    m = orc.Msa(wl.host_reads(0, k))
    pn, pc = device.pileup(ctx, device.synth_device(ctx, wl, 0, k), thr, 2252, W)
    assert (pn.cpu().numpy()[:W * 6].reshape(W, 6) == m.nuc_counts()).all()
    assert (pc.cpu().numpy()[:W * 64].reshape(W, 64) == m.codon_counts()).all()


@pytest.mark.parametrize("W,rich,thr", [(5000, False, (-1, -1, -1, -1)), (9000, True, (40, -1, 42, -1))])
def test_wide_window_stream_mode(ctx, W, rich, thr):
    """Windows wider than 4096 columns go through the stream kernel's column slabs (forced with MSQ_PILEUP_STREAM):
    heavy indels, staggered and short reads, codons across slab borders."""
    reads = cases.random_reads(21 + W, 150, W, rich=rich, p_ins=0.04, p_del=0.04)
    m = orc.Msa(reads, thr)
    _, nuc, codon = gpu_counts(ctx, reads, thr, abi.MSQ_PILEUP_STREAM)
    assert (nuc == m.nuc_counts()).all()
    assert (codon == m.codon_counts()).all()


def test_launch_splitting(ctx):
    """More reads than one launch's bit-sliced counters hold (1023 per wave): msq_pileup splits the reads over
    several launches.  3.4e6 short reads; checked through shard additivity and the oracle on a prefix."""
    from minorseq_amd.synth import Workload, hxb2
    n, W = 3400000, 100
    wl = Workload("short", n, 2252, W, [], hxb2())
    thr = abi.QvThresholds.make()
    dr = device.synth_device(ctx, wl, 0, n)
    nuc, codon = device.pileup(ctx, dr, thr, 2252, W, mode=abi.MSQ_PILEUP_STREAM)
    N = nuc.cpu().numpy()[:W * 6].reshape(W, 6).astype(np.int64)
    assert (N.sum(1) == n).all()
    del dr
    acc_n, acc_c = device.alloc_counts(ctx, W)
    for first, cnt in ((0, 1000000), (1000000, 1300000), (2300000, 1100000)):
        part = device.synth_device(ctx, wl, first, cnt)
        device.pileup(ctx, part, thr, 2252, W, acc_n, acc_c)
        del part
    assert torch.equal(acc_n, nuc) and torch.equal(acc_c, codon)
    k = 3000
    m = orc.Msa(wl.host_reads(0, k))
    pn, pc = device.pileup(ctx, device.synth_device(ctx, wl, 0, k), thr, 2252, W)
    assert (pn.cpu().numpy()[:W * 6].reshape(W, 6) == m.nuc_counts()).all()
    assert (pc.cpu().numpy()[:W * 64].reshape(W, 64) == m.codon_counts()).all()


@pytest.mark.parametrize("seed,W,rich,thr", [(31, 300, False, (-1, -1, -1, -1)), (32, 2500, True, (40, -1, 42, -1)),
                                             (33, 5000, False, (-1, -1, -1, -1))])
def test_phase_codon_keys(ctx, seed, W, rich, thr):
    """msq_phase_codons: per read and position the three row symbols MSARow::CodonAt would string
    together, against the oracle's rows; indices CodonAt leaves out come back as 7."""
    reads = cases.random_reads(seed, 200, W, rich=rich, p_ins=0.03, p_del=0.03)
    m = orc.Msa(reads, thr)
    rows = m.rows()
    n = reads["n_reads"]
    rng = np.random.RandomState(seed)
    pos = np.array(sorted(set([-3, -2, -1, 0, 1, W - 3, W - 2, W - 1, W] + list(rng.randint(1, W - 3, size=40)))), dtype=np.int32)
    dr = device.upload(device.pack_host(reads), ctx.device)
    b, e = device.window_bounds(ctx, dr)
    keys = torch.zeros(n * len(pos), dtype=torch.int16, device="cuda")
    abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, abi.QvThresholds.make(*thr), b, e - b, len(pos), abi.ptr(pos),
                                         keys.data_ptr()))
    got = keys.cpu().numpy().view(np.uint16).reshape(n, len(pos))
    code = {ord(" "): 0, ord("A"): 1, ord("C"): 2, ord("G"): 3, ord("T"): 4, ord("-"): 5, ord("N"): 6}
    for k, p in enumerate(pos):
        want = np.zeros(n, dtype=np.uint16)
        for j in range(3):
            i = int(p) + j
            if 0 < i < W:
                sym = np.array([code[c] for c in rows[:, i]], dtype=np.uint16)
            else:
                sym = np.full(n, 7, dtype=np.uint16)
            want |= sym << (3 * j)
        assert (got[:, k] == want).all(), (k, p)


@pytest.mark.parametrize("name", sorted(EDGE) + ["random_indels", "random_rich", "wide"])
def test_device_resident_window(ctx, name):
    """msq_window_bounds_device + msq_pileup_device_window (no host round trip) against msq_window_bounds +
    msq_pileup: the same bounds, the same counts -- with matrices of exactly the window's width and wider ones
    (the extra columns stay zero); a window wider than the matrices raises the deferred MSQ_ERR_WINDOW."""
    if name == "random_indels":
        reads, thr = cases.random_reads(51, 300, 700, p_ins=0.05), (-1, -1, -1, -1)
    elif name == "random_rich":
        reads, thr = cases.random_reads(52, 200, 2300, rich=True, p_ins=0.05), (30, 20, 42, 25)
    elif name == "wide":
        reads, thr = cases.random_reads(53, 60, 5000), (-1, -1, -1, -1)
    else:
        reads, thr = EDGE[name]
    q = abi.QvThresholds.make(*thr)
    dr = device.upload(device.pack_host(reads), ctx.device)
    b, e = device.window_bounds(ctx, dr)
    bounds = device.window_bounds_device(ctx, dr)
    assert bounds.tolist() == [-b, e]
    if dr.n_reads == 0:
        return
    W = e - b
    if dr.flags & abi.MSQ_READS_HAS_EXOTIC:  # the streaming kernel does not take these: a clear refusal
        with pytest.raises(abi.MsqError) as err:
            device.pileup_device_window(ctx, dr, q, bounds, b, W)
        assert err.value.code == abi.MSQ_ERR_ARG
        return
    want_nuc, want_codon = device.pileup(ctx, dr, q, b, W)
    for W_max in (W, W + 1, W + 77):
        nuc, codon = device.pileup_device_window(ctx, dr, q, bounds, b, W_max)
        nuc, codon = nuc.cpu().numpy().reshape(-1, 6), codon.cpu().numpy().reshape(-1, 64)
        assert (nuc[:W] == want_nuc.cpu().numpy().reshape(-1, 6)[:W]).all() and not nuc[W:].any()
        assert (codon[:W] == want_codon.cpu().numpy().reshape(-1, 64)[:W]).all() and not codon[W:].any()
    if W > 3:
        with pytest.raises(abi.MsqError) as err:
            device.pileup_device_window(ctx, dr, q, bounds, b, W - 1)
        assert err.value.code == abi.MSQ_ERR_WINDOW
    for wrong in (b - 1, b + 1):  # the plan and the matrices were made for another origin
        with pytest.raises(abi.MsqError) as err:
            device.pileup_device_window(ctx, dr, q, bounds, wrong, W + 4)
        assert err.value.code == abi.MSQ_ERR_WINDOW
