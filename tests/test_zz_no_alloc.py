"""msq_pileup_noalloc: the pileup entry point that allocates nothing (its workgroup partials live in a device array
built into the library).  Its counts must equal msq_pileup's, bit for bit, on two calls in a row -- the second on
a smaller input, so that whatever the first call left in the shared array must not leak into the second -- and it
must refuse what is beyond its stated limits."""
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


def both(ctx, dr, thr, b0, W):
    want_n, want_c = device.pileup(ctx, dr, thr, b0, W, mode=abi.MSQ_PILEUP_STREAM)
    nuc, codon = device.alloc_counts(ctx, W)
    abi.check(abi.lib().msq_pileup_noalloc(ctx.h, dr.view, thr, b0, W, nuc.data_ptr(), codon.data_ptr()))
    abi.check(abi.lib().msq_pileup_check(ctx.h))
    return (want_n, want_c), (nuc, codon)


def test_two_calls_in_a_row_second_smaller(ctx):
    thr = abi.QvThresholds.make()
    wl = synth.workload("pol_100k", n_reads=60000)
    big = device.synth_device(ctx, wl, 0, 60000)
    (wn, wc), (n, c) = both(ctx, big, thr, wl.win_begin0, wl.W)
    assert torch.equal(n, wn) and torch.equal(c, wc)
    assert int(n.sum()) == 60000 * wl.W
    # the second call: fewer reads, a narrower window, a QV mask (another kernel variant, another partial layout)
    reads = cases.random_reads(91, 300, 700, p_ins=0.05)
    small = device.upload(device.pack_host(reads), ctx.device)
    b0, e0 = device.window_bounds(ctx, small)
    thr2 = abi.QvThresholds.make(qual=40)
    (wn, wc), (n, c) = both(ctx, small, thr2, b0, e0 - b0)
    assert torch.equal(n, wn) and torch.equal(c, wc)
    # and the big one again after the small one
    (wn, wc), (n, c) = both(ctx, big, thr, wl.win_begin0, wl.W)
    assert torch.equal(n, wn) and torch.equal(c, wc)


@pytest.mark.parametrize("qual", [-1, 40])
def test_it_allocates_nothing(qual):
    """The entry point's defining property, asserted: on a FRESH context (no msq_pileup before it, so nothing has grown
    the context's scratch yet) a warm-up call loads the code object, then a second call leaves both the device's free
    memory and the context's own allocations exactly as they were -- for the unmasked kernel and the QV-mask variant.
    The counts are compared with msq_pileup's only afterwards (that call may allocate)."""
    L = abi.lib()
    c = device.Ctx(0)
    try:
        thr = abi.QvThresholds.make(qual=qual)
        wl = synth.workload("pol_100k", n_reads=20000)
        dr = device.synth_device(c, wl, 0, 20000)
        nuc, codon = device.alloc_counts(c, wl.W)
        abi.check(L.msq_pileup_noalloc(c.h, dr.view, thr, wl.win_begin0, wl.W, nuc.data_ptr(), codon.data_ptr()))  # warm-up
        c.sync()
        nuc.zero_()
        codon.zero_()
        torch.cuda.synchronize()
        own0, free0 = L.msq_debug_ctx_device_bytes(c.h), torch.cuda.mem_get_info()[0]
        abi.check(L.msq_pileup_noalloc(c.h, dr.view, thr, wl.win_begin0, wl.W, nuc.data_ptr(), codon.data_ptr()))
        c.sync()
        assert L.msq_debug_ctx_device_bytes(c.h) == own0
        assert torch.cuda.mem_get_info()[0] == free0
        abi.check(L.msq_pileup_check(c.h))
        got_n, got_c = nuc.clone(), codon.clone()
        wn, wc = device.pileup(c, dr, thr, wl.win_begin0, wl.W, mode=abi.MSQ_PILEUP_STREAM)
        assert torch.equal(got_n, wn) and torch.equal(got_c, wc)
    finally:
        c.close()


def test_accumulates_like_msq_pileup(ctx):
    thr = abi.QvThresholds.make()
    wl = synth.workload("pol_100k", n_reads=5000)
    nuc, codon = device.alloc_counts(ctx, wl.W)
    for first, cnt in ((0, 3000), (3000, 2000)):
        part = device.synth_device(ctx, wl, first, cnt)
        abi.check(abi.lib().msq_pileup_noalloc(ctx.h, part.view, thr, wl.win_begin0, wl.W, nuc.data_ptr(), codon.data_ptr()))
    abi.check(abi.lib().msq_pileup_check(ctx.h))
    wn, wc = device.pileup(ctx, device.synth_device(ctx, wl, 0, 5000), thr, wl.win_begin0, wl.W)
    assert torch.equal(nuc, wn) and torch.equal(codon, wc)


def test_limits_are_refused(ctx):
    L = abi.lib()
    thr = abi.QvThresholds.make()
    wide = device.upload(device.pack_host(cases.random_reads(92, 20, 4200)), ctx.device)
    b0, e0 = device.window_bounds(ctx, wide)
    assert e0 - b0 > 4096
    nuc, codon = device.alloc_counts(ctx, e0 - b0)
    assert L.msq_pileup_noalloc(ctx.h, wide.view, thr, b0, e0 - b0, nuc.data_ptr(), codon.data_ptr()) == abi.MSQ_ERR_ARG
    assert b"4096" in L.msq_last_error()
    exotic = device.upload(device.pack_host(cases.make_reads([(0, "===", "ARG")])), ctx.device)
    assert exotic.flags & abi.MSQ_READS_HAS_EXOTIC
    assert L.msq_pileup_noalloc(ctx.h, exotic.view, thr, 0, 3, nuc.data_ptr(), codon.data_ptr()) == abi.MSQ_ERR_ARG
    assert L.msq_pileup_noalloc(ctx.h, exotic.view, abi.QvThresholds.make(qual=300), 0, 3, nuc.data_ptr(), codon.data_ptr()) == abi.MSQ_ERR_ARG
    assert not nuc.any() and not codon.any()
    # an empty set is no work, as for msq_pileup
    empty = device.upload(device.pack_host(cases.make_reads([])), ctx.device)
    assert L.msq_pileup_noalloc(ctx.h, empty.view, thr, 0, 10, nuc.data_ptr(), codon.data_ptr()) == abi.MSQ_OK
