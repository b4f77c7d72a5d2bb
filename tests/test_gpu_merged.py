"""Wide windows in one launch (stream_launch_merged, pileup_kernel<.., MERGED>): the column slabs of a window share one
launch, each with a share of the workgroups, instead of taking a launch each.  The counts must be those of one launch
per slab (MSQ_NO_MERGE=1) and of the simple kernel, bit for bit: one read set streamed by every slab, a partition's
set per slab, windows with empty columns around the reads, QV thresholds baked in; and the inputs the merged launch
does not take (few reads, a QV plane in the loop) must still come out right through the launches per slab."""
import pytest
import torch

from minorseq_amd import abi, device, synth

pytestmark = pytest.mark.gpu
NONE = abi.QvThresholds.make()


LIVE = []


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    LIVE.append(c)
    yield c
    LIVE.remove(c)
    c.close()


def merged_launches():
    return abi.lib().msq_debug_merged_launches()


TUNE_NO_MERGE = 1  # MSQ_TUNE_NO_MERGE


def both_ways(monkeypatch, run, expect_merged=True, ctx=None):
    """run() with the merged launch and with one launch per slab."""
    before = merged_launches()
    got = run()
    took = merged_launches() - before
    assert (took > 0) == expect_merged, took
    # the switch the library reads from MSQ_NO_MERGE when a context is made, set on the contexts alive in this module
    for c in LIVE:
        abi.check(abi.lib().msq_debug_tune(c.h, TUNE_NO_MERGE, 1))
    before = merged_launches()
    try:
        want = run()
    finally:
        for c in LIVE:
            abi.check(abi.lib().msq_debug_tune(c.h, TUNE_NO_MERGE, 0))
    assert merged_launches() == before
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    return got


@pytest.mark.parametrize("name,n,kw", [
    ("hiv_mix_200k", 40000, {}),
    ("hiv_mix_200k", 30000, dict(ins=0.02, dele=0.03, sub=0.01, n_minors=1, minor_frac=0.2, minor_codons=200)),
    ("amp20k_10m", 24000, {}),
    ("hiv_mix_200k", 36000, dict(jitter_frac=0.6, max_jitter=3000, ins=0.01)),  # ragged reads: the slabs hold unequal shares
    ("amp20k_10m", 30000, dict(jitter_frac=0.9, max_jitter=9000, dele=0.02, seed=7)),
    ("amp20k_10m", 450000, {}),  # more reads than a slab's share of the waves can count in one launch: two rounds
])
def test_one_set_and_partition(ctx, monkeypatch, name, n, kw):
    wl = synth.workload(name, n_reads=n, **kw)
    dr = device.synth_device(ctx, wl, 0, n)
    simple = device.pileup(ctx, dr, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_SIMPLE)
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, dr, NONE, wl.win_begin0, wl.W))
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
    # without the column index (the sum-and-walk kernels)
    plain = device.DeviceReads(dr.n_reads, dr.n_groups, {k: v for k, v in dr.t.items() if k != "gcol"}, dr.flags)
    assert not plain.view.gcol
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, plain, NONE, wl.win_begin0, wl.W))
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
    # a window with empty columns on both sides: other slab borders, slabs that hold few of the reads' groups
    b, W = wl.win_begin0 - 700 if wl.win_begin0 >= 700 else 0, wl.W + 1500
    wide = device.pileup(ctx, dr, NONE, b, W, mode=abi.MSQ_PILEUP_SIMPLE)
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, dr, NONE, b, W))
    assert torch.equal(got[0], wide[0]) and torch.equal(got[1], wide[1])
    # the window left in HBM by the bounds kernel, matrices wider than the window (the slabs are cut from the capacity)
    bounds = device.window_bounds_device(ctx, dr)
    cap = wl.W + 777
    got = both_ways(monkeypatch, lambda: device.pileup_device_window(ctx, dr, NONE, bounds, wl.win_begin0, cap))
    assert bounds.tolist() == [-wl.win_begin0, wl.win_begin0 + wl.W]
    assert torch.equal(got[0][:wl.W * 6], simple[0][:wl.W * 6]) and torch.equal(got[1][:wl.W * 64], simple[1][:wl.W * 64])
    assert not got[0][wl.W * 6:].any() and not got[1][wl.W * 64:].any()
    part = device.Partition(ctx, dr, wl.win_begin0, wl.W)
    assert part.n_slabs > 1
    got = both_ways(monkeypatch, lambda: part.pileup(NONE))
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
    # calls in a row: the partials of the last call must not reach the next
    again = part.pileup(NONE)
    assert torch.equal(again[0], simple[0]) and torch.equal(again[1], simple[1])
    part.close()


def test_baked_thresholds_and_what_is_left_to_the_launches_per_slab(ctx, monkeypatch):
    wl = synth.workload("hiv_mix_200k", n_reads=30000, rich_qv=True)
    dr = device.synth_device(ctx, wl, 0, 30000)
    thr = abi.QvThresholds.make(qual=60, sub=42)
    simple = device.pileup(ctx, dr, thr, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert int(simple[0].view(-1, 6)[:, 5].sum()) > 0  # masked bases: 'N'
    # a QV plane in the loop: one launch per slab
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, dr, thr, wl.win_begin0, wl.W), expect_merged=False)
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
    baked = device.bake_qv(ctx, dr, thr)  # the thresholds in the planes: the kernels with an N counter, merged
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, baked, NONE, wl.win_begin0, wl.W))
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
    baked.close()
    # few reads: the slabs cannot fill their share of the chip
    few = device.synth_device(ctx, wl, 0, 900)
    simple = device.pileup(ctx, few, NONE, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_SIMPLE)
    got = both_ways(monkeypatch, lambda: device.pileup(ctx, few, NONE, wl.win_begin0, wl.W), expect_merged=False)
    assert torch.equal(got[0], simple[0]) and torch.equal(got[1], simple[1])
