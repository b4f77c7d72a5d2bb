"""GPU parity, per-read rows: msq_rows_decode (MSARow::Bases, include/pacbio/data/MSA.h:96-101; MSAByRow::AddRead,
src/MSA.cpp:144-186) through the C ABI against the oracle's rows, character for character -- every edge set, both
group layouts, thresholds applied in the pass and baked at pack time, sub-ranges of the reads, a window wider than
the reads, a 9000-column window, and the deferred window error.  Parity unpinned like the pileup (the reference
holds no MSA fixture); the rows also reproduce the count matrices (MSAByColumn's loop over Rows(), MSA.cpp:68-87)."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
EDGE = cases.edge_sets()


def decode(ctx, dr, thr, b0, W, first=0, n=None):
    n = dr.n_reads - first if n is None else n
    rows = torch.full((max(n * W, 1),), 0x7f, dtype=torch.uint8, device="cuda")
    abi.check(abi.lib().msq_rows_decode(ctx.h, dr.view, thr, b0, W, first, n, rows.data_ptr()))
    rc = abi.lib().msq_pileup_check(ctx.h)
    return rc, rows.cpu().numpy()[:n * W].reshape(n, W)


@pytest.mark.parametrize("layout", ["normalized", "plain"])
@pytest.mark.parametrize("name", sorted(EDGE) + ["random_rich"])
def test_rows_equal_the_oracles(ctx, name, layout):
    reads, thr = EDGE[name] if name in EDGE else (cases.random_reads(12, 200, 2300, rich=True, p_ins=0.05), (30, 20, 42, 25))
    m = orc.Msa(reads, thr)
    want = m.rows()
    packed = device.pack_host(reads) if layout == "normalized" else cases.pack_plain(reads)
    dr = device.upload(packed, ctx.device, ctx=ctx)
    b0, W = m.begin1 - 1, m.W
    t = abi.QvThresholds.make(*thr)
    rc, got = decode(ctx, dr, t, b0, W)
    assert rc == abi.MSQ_OK and got.shape == want.shape
    assert np.array_equal(got, want), name
    # a range of the reads in the middle; nothing before or behind the rows asked for is touched
    n = dr.n_reads
    if n >= 3:
        rc, part = decode(ctx, dr, t, b0, W, 1, n - 2)
        assert rc == abi.MSQ_OK and np.array_equal(part, want[1:n - 1])
    # MSAByColumn's loop over the rows (src/MSA.cpp:68-87) gives the nucleotide matrix
    nuc = np.stack([(got == ord(c)).sum(0) for c in "ACGT-N"], 1)
    assert np.array_equal(nuc, m.nuc_counts())
    # thresholds baked at pack time: the same rows with no threshold in the pass
    if any(x >= 0 for x in thr):
        baked = device.bake_qv(ctx, dr, t)
        rc, got_b = decode(ctx, baked, abi.QvThresholds.make(), b0, W)
        assert rc == abi.MSQ_OK and np.array_equal(got_b, want)
        baked.close()


def test_wider_window_wide_rows_and_errors(ctx):
    reads = cases.random_reads(31, 40, 9000, p_ins=0.02)
    m = orc.Msa(reads)
    dr = device.upload(device.pack_host(reads), ctx.device)
    none = abi.QvThresholds.make()
    rc, got = decode(ctx, dr, none, m.begin1 - 1, m.W)
    assert rc == abi.MSQ_OK and np.array_equal(got, m.rows())
    # a window with 5 empty columns in front and 9 behind: ' ' there
    rc, wide = decode(ctx, dr, none, m.begin1 - 6, m.W + 14)
    assert rc == abi.MSQ_OK and (wide[:, :5] == 32).all() and (wide[:, -9:] == 32).all() and np.array_equal(wide[:, 5:-9], got)
    # a window that cuts the reads: the deferred error of the pileup (assert(pos >= 0), src/MSA.cpp:149)
    rc, _ = decode(ctx, dr, none, m.begin1 + 10, m.W)
    assert rc == abi.MSQ_ERR_WINDOW
    # argument checks: rows beyond the set, a NULL buffer
    L = abi.lib()
    assert L.msq_rows_decode(ctx.h, dr.view, none, 0, 10, dr.n_reads, 1, 1) == abi.MSQ_ERR_ARG
    assert L.msq_rows_decode(ctx.h, dr.view, none, 0, 10, 0, 1, None) == abi.MSQ_ERR_ARG
    assert L.msq_rows_decode(ctx.h, dr.view, none, 0, 10, 0, 0, None) == abi.MSQ_OK
    empty = device.upload(device.pack_host(cases.make_reads([])), ctx.device)
    assert L.msq_rows_decode(ctx.h, empty.view, none, 0, 10, 0, 0, None) == abi.MSQ_OK


def test_synthetic_rows(ctx):
    wl = synth.workload("pol_100k", n_reads=500)
    dr = device.synth_device(ctx, wl, 0, 500)
    m = orc.Msa(wl.host_reads(0, 500))
    rc, got = decode(ctx, dr, abi.QvThresholds.make(), wl.win_begin0, wl.W)
    assert rc == abi.MSQ_OK and np.array_equal(got, m.rows())
