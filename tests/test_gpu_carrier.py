"""The alt haplotype's carrier rule (learn_alt_words) is a speed hint like the guide: counts are bit-exact under either
rule.  By default only launches of >= 16 batches per wave take it; MSQ_CARRIER_MIN_BATCHES=0 sends every launch through
it, so the edge sets, random reads, mixed samples and the fused step are checked under the rule as well."""
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


@pytest.fixture(params=["0", "1000000"], ids=["carrier", "any-runner-up"])
def rule(request, ctx):
    # MSQ_TUNE_CARRIER_MIN_BATCHES (what MSQ_CARRIER_MIN_BATCHES sets when a context is made)
    abi.check(abi.lib().msq_debug_tune(ctx.h, 2, int(request.param)))
    yield request.param
    abi.check(abi.lib().msq_debug_tune(ctx.h, 2, -1))


def check(ctx, dr, thr, b, W):
    got = device.pileup(ctx, dr, thr, b, W, mode=abi.MSQ_PILEUP_STREAM)
    want = device.pileup(ctx, dr, thr, b, W, mode=abi.MSQ_PILEUP_SIMPLE)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    counts = torch.full((W * 70,), 0x33, dtype=torch.int32, device="cuda")
    bounds = torch.zeros(2, dtype=torch.int32, device="cuda")
    abi.check(abi.lib().msq_pileup_step(ctx.h, dr.view, thr, b, W, bounds.data_ptr(), counts.data_ptr()))
    abi.check(abi.lib().msq_pileup_check(ctx.h))
    assert torch.equal(counts[:W * 6], want[0]) and torch.equal(counts[W * 6:], want[1])


def test_edge_sets(ctx, rule):
    for name, (reads, thr) in cases.edge_sets().items():
        for packed in (device.pack_host(reads), cases.pack_plain(reads)):
            if not packed["n_groups"]:
                continue
            dr = device.upload(packed, ctx.device, ctx=ctx)
            b, e = device.window_bounds(ctx, dr)
            if dr.flags & abi.MSQ_READS_HAS_EXOTIC or e <= b:
                continue
            check(ctx, dr, NONE, b, e - b)
            baked = device.bake_qv(ctx, dr, abi.QvThresholds.make(*thr))
            check(ctx, baked, NONE, b, e - b)
            baked.close()


@pytest.mark.parametrize("seed,n,W,p_sub", [(1, 700, 900, 0.02), (2, 5000, 2800, 0.002), (3, 100, 4000, 0.2), (4, 40, 64, 0.05)])
def test_random_reads(ctx, rule, seed, n, W, p_sub):
    reads = cases.random_reads(seed, n, W, p_sub=p_sub, p_ins=0.03)
    dr = device.upload(device.pack_host(reads), ctx.device, ctx=ctx)
    b, e = device.window_bounds(ctx, dr)
    check(ctx, dr, NONE, b, e - b)


@pytest.mark.parametrize("kw", [dict(n_minors=1, minor_frac=0.3, minor_codons=45), dict(n_minors=4, minor_frac=0.2, minor_codons=100),
                                dict(n_minors=2, minor_frac=0.03, minor_codons=300), dict(sub=0.02)])
def test_mixed_samples(ctx, rule, kw):
    wl = synth.workload("pol_100k", n_reads=30000, **kw)
    dr = device.synth_device(ctx, wl, 0, 30000)
    check(ctx, dr, NONE, wl.win_begin0, wl.W)
