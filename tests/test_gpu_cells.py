"""GPU parity, testing pass: msq_fisher_batch against the bit-exact vectors of the reference's compiled
Fisher.cpp and its R known-answer table; msq_call_cells against the oracle's CallVariants cell loop --
positions and counts identical, expected identical, p within 1e-9 relative (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest

import cases
from minorseq_amd import abi, device, synth
from oracle import orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P_RTOL = 1e-9      # north_star: p-values within 1e-9 relative
P_ATOL = 1e-300    # denormal results only: device exp vs glibc exp may round the last denormal bit differently


def close(p_gpu, p_ref):
    return abs(p_gpu - p_ref) <= P_RTOL * abs(p_ref) + P_ATOL


def test_fisher_equivalent_with_R(ctx):
    # mirrors the reference's FisherTest.EquivalentWithR (tests/unit/FisherTest.cpp:49-74)
    p_r = json.load(open(os.path.join(GOLD, "fisher_R_greater.json")))["p"]
    i = np.arange(11, 100)
    p = device.fisher_batch(ctx, i, np.full_like(i, 1000), np.full_like(i, 10), np.full_like(i, 1000))
    for k, x in enumerate(i):
        assert abs(p_r[x] - p[k]) <= p_r[x] / 1e5


def test_fisher_golden_vectors(ctx):
    cs = json.load(open(os.path.join(GOLD, "fisher_ref.json")))["cases"]
    a, b, c, d = (np.array([x[k] for x in cs]) for k in range(4))
    want = np.array([float.fromhex(x[4]) for x in cs])
    p = device.fisher_batch(ctx, a, b, c, d)
    bad = [(cs[k][:4], p[k], want[k]) for k in range(len(cs)) if not close(p[k], want[k])]
    assert not bad, bad[:5]
    # quirks survive: saddle -> exactly 1, strong signal -> exactly 0, table edge -> point probability
    assert p[1] == 1.0 and want[1] == 1.0
    assert (p[want == 0.0] == 0.0).all()


def test_fisher_scalar_and_errors(ctx):
    import ctypes as C
    out = C.c_double()
    abi.check(abi.lib().msq_fisher(ctx.h, 20, 1000, 10, 1000, C.byref(out)))
    assert close(out.value, orc.fisher(20, 1000, 10, 1000))
    assert abi.lib().msq_fisher(ctx.h, -1, 1, 1, 1, C.byref(out)) == abi.MSQ_ERR_ARG


def run_call(ctx, reads, thr, genes, reference, maximal_perc=100.0):
    m = orc.Msa(reads, thr)
    dr = device.upload(device.pack_host(reads), ctx.device)
    b0, e0 = device.window_bounds(ctx, dr)
    W = e0 - b0
    nuc, codon = device.pileup(ctx, dr, abi.QvThresholds.make(*thr), b0, W)
    gb = [g[0] for g in genes]
    ge = [g[1] for g in genes]
    plan = device.Plan(ctx, gb, ge, reference, b0 + 1, W)
    ctx.lfact_reserve(2 * reads["n_reads"] + 2)
    device.call_cells(ctx, plan, codon, maximal_perc=maximal_perc)
    pos, p, expected, nt = device.results(plan)
    o_pos, o_cells, o_nt = m.call(gb, ge, reference, maximal_perc=maximal_perc)
    assert nt == o_nt
    assert (pos == o_pos).all()
    tested = np.argwhere(p >= 0)
    assert len(tested) == len(o_cells)
    for (k, cod), oc in zip(tested, o_cells):  # both in (position, codon) order
        assert (k, cod) == (oc.position, oc.codon)
        assert expected[k, cod] == oc.expected
        assert close(p[k, cod], oc.p), (k, cod, p[k, cod], oc.p)
        assert (p[k, cod] < np.float64(np.float32(0.01))) == bool(oc.significant)
    plan.close()
    return len(o_cells)


@pytest.mark.parametrize("name", sorted(cases.edge_sets()))
def test_edge_sets(ctx, name):
    reads, thr = cases.edge_sets()[name]
    if reads["n_reads"] == 0:
        return
    b = int(reads["ref_start"].min()) + 1
    W = int(reads["ref_end"].max()) + 1 - b
    run_call(ctx, reads, thr, [], "")
    genes = [(b, b + W), (b + 1, b + 40), (b + 2, b + W + 9), (max(1, b - 6), b + 12)]
    ref = "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(b + W + 60))
    run_call(ctx, reads, thr, genes, "")
    run_call(ctx, reads, thr, genes, ref)
    run_call(ctx, reads, thr, genes, ref, maximal_perc=50.0)


@pytest.mark.parametrize("name,n,with_ref", [("pol_100k", 6000, True), ("pol_100k", 3000, False), ("hiv_1k", 1000, True)])
def test_synthetic_calls(ctx, name, n, with_ref):
    wl = synth.workload(name, n_reads=n)
    reads = wl.host_reads(0, n)
    genes = [(g["begin"], g["end"]) for g in wl.genes]
    ncells = run_call(ctx, reads, (-1, -1, -1, -1), genes, wl.reference if with_ref else "")
    assert ncells > 100


def test_reference_out_of_range_and_small_table(ctx):
    import ctypes as C
    plan = device.Plan(ctx, [1], [40], "ACGT" * 5, 1, 50)  # gene runs past the 20-nt reference
    _, codon = device.alloc_counts(ctx, 50)
    em = abi.ErrorModel()
    abi.lib().msq_error_model_default(C.byref(em))
    rc = abi.lib().msq_call_cells(ctx.h, plan.h, codon.data_ptr(), C.byref(em), 100.0, plan.positions.data_ptr(),
                                  plan.p.data_ptr(), plan.expected.data_ptr(), plan.n_tests.data_ptr())
    assert rc == abi.MSQ_ERR_RANGE  # std::out_of_range of referenceSequence.substr (src/AminoAcidCaller.cpp:446)
    plan.close()


@pytest.mark.parametrize("coverage", [1000000, 8000000])
def test_high_coverage_cells(ctx, coverage):
    """The testing pass at the coverage of one GPU's shard and of an 8-GPU node (count matrices written directly:
    no reads needed).  Here the summation loop of fisher_exact_tiss runs 175 / 1400 steps per cell and is spread
    over the wave (d_fisher_wave): every cell against the oracle's sequential loop, 1e-9 relative."""
    import torch
    rng = np.random.RandomState(coverage % 97)
    W, b1 = 64, 101
    codon = np.zeros((W, 64), dtype=np.int64)
    major = rng.randint(0, 64, size=W)
    em = np.zeros(4)
    orc.lib().orc_error_estimates_default(em.ctypes.data)
    match, sub = em[0], em[1]
    for c in range(1, W - 2):
        others = [int(major[c]) ^ (d << sh) for sh in (0, 2, 4) for d in (1, 2, 3)]  # one substitution away: long loops
        picks = rng.choice(others, size=8, replace=False)
        exp1 = coverage * match * match * sub  # a codon one substitution away
        for j, k in enumerate(picks):
            codon[c, k] = [1, 3, int(exp1 * 0.5), int(exp1), int(exp1 * 1.05) + 1, int(exp1 * 1.3), int(exp1 * 3), 40000][j]
        codon[c, major[c]] = coverage - codon[c].sum()
    d_codon = torch.from_numpy(codon.astype(np.int32).reshape(-1)).cuda()
    plan = device.Plan(ctx, [b1], [b1 + W], "", b1, W)
    ctx.lfact_reserve(2 * coverage + 2)
    device.call_cells(ctx, plan, d_codon)
    pos, p, expected, nt = device.results(plan)
    assert nt == int((codon[[int(x[3]) for x in pos]] > 0).sum())
    checked = 0
    for k, row in enumerate(pos):
        c, cov, ref = int(row[3]), int(row[4]), int(row[5])
        if not (1 <= c < W - 2):
            continue
        assert cov == coverage and ref == major[c]
        for cod in np.nonzero(codon[c])[0]:
            if cod == ref:
                assert p[k, cod] < 0
                continue
            pr = 1.0
            for sh in (4, 2, 0):
                pr *= sub if ((ref >> sh) & 3) != ((cod >> sh) & 3) else match
            e = cov * pr
            assert expected[k, cod] == e
            cnt = int(codon[c, cod])
            want = min(1.0, orc.fisher(cnt, cov - cnt, int(np.ceil(e)), int(np.ceil(cov - e))) * nt)
            assert close(p[k, cod], want), (c, cod, cnt, p[k, cod], want)
            checked += 1
    assert checked >= 150
    plan.close()
