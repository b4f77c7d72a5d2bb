"""Seeded fuzzing of the whole path at small sizes: random windows (down to one column), read counts, error and
insertion rates, QV thresholds, layouts -- pileup counts, window bounds, cells, insertion histogram and phasing
keys, every one against the oracle.  Small cases find the border conditions the big synthetic workloads never
reach (windows narrower than a codon, reads that are one base long, columns nobody covers)."""
import os

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return device.Ctx(0)


def case_of(seed):
    rng = np.random.RandomState(1000 + seed)
    W = int(rng.choice([1, 2, 3, 4, 7, 31, 32, 33, 64, 65, 100, 257, 600, 2049]))
    n = int(rng.choice([1, 2, 9, 40, 200]))
    rich = bool(rng.randint(2))
    reads = cases.random_reads(seed, n, W, start0=int(rng.randint(0, 5000)), rich=rich, p_sub=float(rng.choice([0.0, 0.02, 0.3])),
                               p_del=float(rng.choice([0.0, 0.03, 0.3])), p_ins=float(rng.choice([0.0, 0.03, 0.4])),
                               stagger=bool(rng.randint(2)), exotic=bool(rng.randint(4) == 0))
    thr = tuple(int(x) for x in rng.choice([-1, -1, 0, 20, 42, 93, 255], size=4))
    return reads, thr, W


# MSQ_FUZZ_FIRST / MSQ_FUZZ_N: another seed range for a one-off longer run (the default 150 are what the suite runs)
_FIRST, _N = int(os.environ.get("MSQ_FUZZ_FIRST", "0")), int(os.environ.get("MSQ_FUZZ_N", "150"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _N))
def test_fuzz(ctx, seed):
    reads, thr, W = case_of(seed)
    m = orc.Msa(reads, thr)
    q = abi.QvThresholds.make(*thr)
    for plain in (False, True):
        packed = device.pack_device(ctx, reads, plain=plain, chunk_bases=int(1 + seed * 37))
        dr = device.upload(packed)
        b0, e0 = device.window_bounds(ctx, dr)
        assert (b0 + 1, e0 + 1) == (m.begin1, m.end1) and e0 - b0 == m.W
        for mode in (abi.MSQ_PILEUP_AUTO, abi.MSQ_PILEUP_SIMPLE):
            nuc, codon = device.pileup(ctx, dr, q, b0, m.W, mode=mode)
            assert (nuc.cpu().numpy()[:m.W * 6].reshape(-1, 6) == m.nuc_counts()).all(), (plain, mode)
            assert (codon.cpu().numpy()[:m.W * 64].reshape(-1, 64) == m.codon_counts()).all(), (plain, mode)
        if packed["n_groups"]:  # the same set with a group column index: the kernels that skip the prefix sum
            idx = device.upload(packed, ctx.device, ctx=ctx)
            nuc, codon = device.pileup(ctx, idx, q, b0, m.W)
            assert (nuc.cpu().numpy()[:m.W * 6].reshape(-1, 6) == m.nuc_counts()).all(), ("index", plain)
            assert (codon.cpu().numpy()[:m.W * 64].reshape(-1, 64) == m.codon_counts()).all(), ("index", plain)
        if not (dr.flags & abi.MSQ_READS_HAS_EXOTIC):  # the window left in HBM, matrices wider than the window
            bounds = device.window_bounds_device(ctx, dr)
            nuc, codon = device.pileup_device_window(ctx, dr, q, bounds, b0, m.W + seed % 5)
            assert bounds.tolist() == [-b0, e0]
            assert (nuc.cpu().numpy()[:m.W * 6].reshape(-1, 6) == m.nuc_counts()).all(), plain
            assert (codon.cpu().numpy()[:m.W * 64].reshape(-1, 64) == m.codon_counts()).all(), plain
            codon = codon[:m.W * 64].contiguous()
        if reads["n_reads"]:  # the set aligned under its thresholds (msq_reads_align): the aligned-set kernel, or -- exotic symbols -- the fallback
            al = device.align(ctx, dr, q, b0 - 32 * (seed % 3))  # the window 0, 1 or 2 words right of the set's origin
            nuc, codon = device.pileup(ctx, al, abi.QvThresholds.make(), b0, m.W)
            assert (nuc.cpu().numpy()[:m.W * 6].reshape(-1, 6) == m.nuc_counts()).all(), ("aligned", plain)
            assert (codon.cpu().numpy()[:m.W * 64].reshape(-1, 64) == m.codon_counts()).all(), ("aligned", plain)
            al.close()
        baked = device.bake_qv(ctx, dr, q)  # thresholds applied at pack time: same counts with no thresholds left
        for mode in (abi.MSQ_PILEUP_AUTO, abi.MSQ_PILEUP_SIMPLE):
            nuc, codon = device.pileup(ctx, baked, abi.QvThresholds.make(-1, -1, -1, -1), b0, m.W, mode=mode)
            assert (nuc.cpu().numpy()[:m.W * 6].reshape(-1, 6) == m.nuc_counts()).all(), ("baked", plain, mode)
            assert (codon.cpu().numpy()[:m.W * 64].reshape(-1, 64) == m.codon_counts()).all(), ("baked", plain, mode)
        codon = codon[:m.W * 64].contiguous()
        if plain:
            assert device.insertion_counts(ctx, baked, b0, m.W).tolist() == device.insertion_counts(ctx, dr, b0, m.W).tolist()
        baked.close()
        if plain:
            counts = device.insertion_counts(ctx, dr, b0, m.W)
            hist = [dict() for _ in range(m.W)]
            for col, s in device.insertion_records(ctx, dr, b0, m.W):
                hist[col][s] = hist[col].get(s, 0) + 1
            for c in range(m.W):
                want = m.col_insertions(c)
                assert hist[c] == want and counts[c] == sum(want.values()), c
    # cells: one ORF over the window, as juliet without a target config (src/AminoAcidCaller.cpp:399-403)
    gb, ge = np.array([m.begin1], dtype=np.int32), np.array([m.end1], dtype=np.int32)
    plan = device.Plan(ctx, gb, ge, "", b0 + 1, m.W)
    ctx.lfact_reserve(2 * reads["n_reads"] + 2)
    device.call_cells(ctx, plan, codon)
    pos, p, expected, n_tests = device.results(plan)
    o_pos, o_cells, o_nt = m.call(gb, ge, "")
    assert n_tests == o_nt and (pos == o_pos).all()
    for c in o_cells:
        g = p[c.position, c.codon]
        assert abs(g - c.p) <= 1e-9 * abs(c.p) + 1e-300, (c.position, c.codon, g, c.p)
