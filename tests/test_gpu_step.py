"""msq_pileup_step: bounds + cleared matrices + pileup as one call (one kernel + its fold when the streaming kernel can
take the whole call).  It must leave exactly what msq_pileup_begin followed by msq_pileup leave -- bounds and both
count matrices, bit for bit -- whatever the caller's buffer held before, on calls in a row with different windows
(the context's deviation matrix must come back cleared), and on the inputs that take the fallback."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def two_calls(ctx, dr, thr, b0, W):
    L = abi.lib()
    want = torch.full((W * 70,), 7, dtype=torch.int32, device="cuda")
    wb = torch.zeros(2, dtype=torch.int32, device="cuda")
    abi.check(L.msq_pileup_begin(ctx.h, dr.view, wb.data_ptr(), want.data_ptr(), want.numel()))
    abi.check(L.msq_pileup(ctx.h, dr.view, thr, b0, W, want[:W * 6].data_ptr(), want[W * 6:].data_ptr(), abi.MSQ_PILEUP_AUTO))
    abi.check(L.msq_pileup_check(ctx.h))
    return wb, want


def step(ctx, dr, thr, b0, W):
    L = abi.lib()
    got = torch.full((W * 70,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")  # stale contents must not survive
    gb = torch.full((2,), -3, dtype=torch.int32, device="cuda")
    abi.check(L.msq_pileup_step(ctx.h, dr.view, thr, b0, W, gb.data_ptr(), got.data_ptr()))
    abi.check(L.msq_pileup_check(ctx.h))
    return gb, got


def same(ctx, dr, thr, b0, W):
    wb, want = two_calls(ctx, dr, thr, b0, W)
    gb, got = step(ctx, dr, thr, b0, W)
    assert gb.tolist() == wb.tolist()
    assert torch.equal(got, want)
    return got


@pytest.mark.parametrize("name,n", [("pol_100k", 100000), ("hiv_mix_200k", 50000), ("amp20k_10m", 20000)])
def test_workloads(ctx, name, n):
    wl = synth.workload(name, n_reads=n)
    dr = device.synth_device(ctx, wl, 0, n)
    got = same(ctx, dr, abi.QvThresholds.make(), wl.win_begin0, wl.W)
    assert int(got[:wl.W * 6].sum()) > 0


def test_calls_in_a_row_with_deviations_and_other_windows(ctx):
    """Deviation-heavy random reads (the kernel's global atomics go to the context's matrix), then other windows: what one call leaves in that matrix must not reach the next."""
    thr = abi.QvThresholds.make()
    for seed, n, length in ((5, 400, 900), (6, 3000, 2500), (7, 50, 120), (8, 1500, 900)):
        reads = cases.random_reads(seed, n, length, p_ins=0.05)
        dr = device.upload(device.pack_host(reads), ctx.device)
        b0, e0 = device.window_bounds(ctx, dr)
        same(ctx, dr, thr, b0, e0 - b0)
        same(ctx, dr, thr, b0 - 40, e0 - b0 + 100)  # a window with empty columns on both sides


def test_qv_variants(ctx):
    reads = cases.random_reads(21, 800, 700, rich=True, p_ins=0.04)
    dr = device.upload(device.pack_host(reads), ctx.device)
    b0, e0 = device.window_bounds(ctx, dr)
    for thr in (abi.QvThresholds.make(qual=40), abi.QvThresholds.make(qual=30, sub=42), abi.QvThresholds.make(qual=20, sub=42, ins=30),
                abi.QvThresholds.make(qual=20, del_=30, sub=42, ins=30)):
        same(ctx, dr, thr, b0, e0 - b0)
    baked = device.bake_qv(ctx, dr, abi.QvThresholds.make(qual=40))
    same(ctx, baked, abi.QvThresholds.make(qual=40), b0, e0 - b0)


def test_fallbacks(ctx):
    L = abi.lib()
    thr = abi.QvThresholds.make()
    # wider than one column slab: msq_pileup_begin + msq_pileup inside
    wide = device.upload(device.pack_host(cases.random_reads(92, 40, 4300)), ctx.device)
    b0, e0 = device.window_bounds(ctx, wide)
    assert e0 - b0 > 4096
    same(ctx, wide, thr, b0, e0 - b0)
    # exotic op codes: the simple kernel
    exotic = device.upload(device.pack_host(cases.random_reads(93, 30, 200, exotic=True)), ctx.device)
    if exotic.flags & abi.MSQ_READS_HAS_EXOTIC:
        eb0, ee0 = device.window_bounds(ctx, exotic)
        same(ctx, exotic, thr, eb0, ee0 - eb0)
    # no reads: the defaults of MSAByRow and cleared matrices
    empty = device.upload(device.pack_host(cases.make_reads([])), ctx.device)
    gb, got = step(ctx, empty, thr, 0, 10)
    assert gb.tolist() == [-(2 ** 31 - 1), 0] and not got.any()
    assert L.msq_pileup_step(ctx.h, empty.view, thr, 0, -1, gb.data_ptr(), got.data_ptr()) == abi.MSQ_ERR_ARG
