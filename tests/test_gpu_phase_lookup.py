"""The phasing gather by lookup (phase_lookup_kernel, msq_phase_codons on sets that carry the group column index): the
keys must be MSARow::CodonAt's symbols -- against the oracle's rows, and against the streaming gather on the same set,
bit for bit -- on every edge set, both group layouts, reads with heavy indels (the group that holds a column is then
far behind the guess), wide windows, positions on and beyond the window's borders; a read that leaves the window
raises the same deferred error; the default choice between the two gathers follows the positions per read length."""
import numpy as np
import pytest
import torch

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
NONE = abi.QvThresholds.make()
CODE = {ord(" "): 0, ord("A"): 1, ord("C"): 2, ord("G"): 3, ord("T"): 4, ord("-"): 5, ord("N"): 6}


@pytest.fixture(scope="module")
def ctx():
    c = device.Ctx(0)
    yield c
    c.close()


def lookups():
    return abi.lib().msq_debug_phase_lookups()


def keys_of(ctx, dr, b, W, pos, thr=NONE):
    k = torch.zeros(dr.n_reads * len(pos), dtype=torch.int16, device="cuda")
    abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, thr, b, W, len(pos), abi.ptr(pos), k.data_ptr()))
    abi.check(abi.lib().msq_pileup_check(ctx.h))
    return k


def both(ctx, monkeypatch, dr, b, W, pos, thr=NONE, expect_lookup=True):
    monkeypatch.setenv("MSQ_PHASE_LOOKUP", "1")
    before = lookups()
    got = keys_of(ctx, dr, b, W, pos, thr)
    assert (lookups() - before == 1) == expect_lookup
    monkeypatch.setenv("MSQ_PHASE_LOOKUP", "0")
    before = lookups()
    want = keys_of(ctx, dr, b, W, pos, thr)
    assert lookups() == before
    monkeypatch.delenv("MSQ_PHASE_LOOKUP")
    assert torch.equal(got, want)
    return got


def positions(seed, W, n=40):
    rng = np.random.RandomState(seed)
    inner = list(rng.randint(1, max(2, W - 3), size=n)) if W > 4 else []
    return np.array(sorted(set([-3, -2, -1, 0, 1, 2, W - 3, W - 2, W - 1, W, W + 5] + inner)), dtype=np.int32)


def against_rows(got, reads, thr, pos):
    m = orc.Msa(reads, thr)
    rows, n, W = m.rows(), reads["n_reads"], m.W
    got = got.cpu().numpy().view(np.uint16).reshape(n, len(pos))
    for k, p in enumerate(pos):
        want = np.zeros(n, dtype=np.uint16)
        for j in range(3):
            i = int(p) + j
            sym = np.array([CODE[c] for c in rows[:, i]], dtype=np.uint16) if 0 < i < W else np.full(n, 7, dtype=np.uint16)
            want |= sym << (3 * j)
        assert (got[:, k] == want).all(), (k, p)


@pytest.mark.parametrize("layout", ["normalized", "plain"])
def test_edge_sets(ctx, monkeypatch, layout):
    for name, (reads, thr) in cases.edge_sets().items():
        packed = device.pack_host(reads) if layout == "normalized" else cases.pack_plain(reads)
        if not packed["n_groups"]:
            continue
        dr = device.upload(packed, ctx.device, ctx=ctx)
        b, e = device.window_bounds(ctx, dr)
        if e - b <= 0:
            continue
        pos = positions(7, e - b, 12)
        exotic = bool(dr.flags & abi.MSQ_READS_HAS_EXOTIC)
        got = both(ctx, monkeypatch, dr, b, e - b, pos, expect_lookup=not exotic)
        if not exotic:
            against_rows(got, reads, (-1, -1, -1, -1), pos)
        # thresholds that meet a QV plane: the lookup leaves the call to the streaming gather
        q = abi.QvThresholds.make(*thr)
        has_plane = any(t >= 0 and packed.get(f) is not None for t, f in zip(thr, ("qual", "delqv", "subqv", "insqv")))
        both(ctx, monkeypatch, dr, b, e - b, pos, thr=q, expect_lookup=not exotic and not has_plane)


@pytest.mark.parametrize("seed,n,W,p_ins,p_del", [(1, 300, 700, 0.03, 0.03), (2, 500, 2800, 0.10, 0.05), (3, 64, 5000, 0.02, 0.02),
                                                  (4, 40, 300, 0.5, 0.2), (5, 30, 13000, 0.01, 0.3), (6, 9, 100, 0.8, 0.0)])
def test_random_reads(ctx, monkeypatch, seed, n, W, p_ins, p_del):
    reads = cases.random_reads(seed, n, W, p_ins=p_ins, p_del=p_del, stagger=True)
    for packed in (device.pack_host(reads), cases.pack_plain(reads)):
        dr = device.upload(packed, ctx.device, ctx=ctx)
        b, e = device.window_bounds(ctx, dr)
        pos = positions(seed, e - b)
        got = both(ctx, monkeypatch, dr, b, e - b, pos)
        against_rows(got, reads, (-1, -1, -1, -1), pos)
        # a window with empty columns on both sides
        both(ctx, monkeypatch, dr, max(0, b - 37), e - b + 37 + 70 if b >= 37 else e - b + 70, pos)


def test_baked_sets_and_library_sets(ctx, monkeypatch):
    reads = cases.random_reads(11, 400, 900, rich=True, p_ins=0.04)
    thr = (40, -1, 42, -1)
    lib_set = device.pack_device(ctx, reads, keep=True)
    baked = device.bake_qv(ctx, lib_set, abi.QvThresholds.make(*thr))
    b, e = device.window_bounds(ctx, lib_set)
    pos = positions(11, e - b)
    got = both(ctx, monkeypatch, baked, b, e - b, pos)  # the thresholds are in the planes: 'N' where a QV failed
    against_rows(got, reads, thr, pos)
    got = both(ctx, monkeypatch, lib_set, b, e - b, pos)
    against_rows(got, reads, (-1, -1, -1, -1), pos)
    baked.close()
    lib_set.close()


def test_a_read_that_leaves_the_window(ctx, monkeypatch):
    reads = cases.make_reads([(100, "=" * 64, "ACGT" * 16), (110, "=" * 40, "ACGT" * 10), (90, "=" * 40, "GTAC" * 10)])
    dr = device.upload(device.pack_host(reads), ctx.device, ctx=ctx)
    pos = np.array([1, 5, 20], dtype=np.int32)
    for lookup in ("1", "0"):
        monkeypatch.setenv("MSQ_PHASE_LOOKUP", lookup)
        for b, W, ok in ((90, 74, True), (95, 69, False), (90, 60, False), (80, 100, True)):
            k = torch.zeros(3 * len(pos), dtype=torch.int16, device="cuda")
            abi.check(abi.lib().msq_phase_codons(ctx.h, dr.view, NONE, b, W, len(pos), abi.ptr(pos), k.data_ptr()))
            assert abi.lib().msq_pileup_check(ctx.h) == (abi.MSQ_OK if ok else abi.MSQ_ERR_WINDOW), (lookup, b, W)
    monkeypatch.delenv("MSQ_PHASE_LOOKUP")


@pytest.mark.parametrize("name,n,K,lookup", [("hiv_mix_200k", 30000, 20, True), ("amp20k_10m", 8000, 50, True), ("pol_100k", 30000, 20, False), ("pol_100k", 30000, 5, True),
                                             ("hiv_mix_200k", 4000, 400, False)])
def test_workloads_and_the_default_choice(ctx, monkeypatch, name, n, K, lookup):
    wl = synth.workload(name, n_reads=n)
    dr = device.synth_device(ctx, wl, 0, n)
    pos = np.linspace(1, wl.W - 4, K).astype(np.int32)
    before = lookups()
    default = keys_of(ctx, dr, wl.win_begin0, wl.W, pos)
    assert (lookups() - before == 1) == lookup
    got = both(ctx, monkeypatch, dr, wl.win_begin0, wl.W, pos)
    assert torch.equal(default, got)
