"""The oracle's Fisher against (1) the reference's own known-answer table (tests/unit/FisherTest.cpp:53-73),
(2) bit-exact vectors captured from the reference's compiled Fisher.cpp, (3) that compiled code itself when
oracle/_ref is present.  CPU only."""
import json
import os

import pytest

from oracle import orc, pymodel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_equivalent_with_R():
    # the reference's FisherTest.EquivalentWithR: i = 11..99, EXPECT_NEAR(p_R, p, p_R / 1e5)
    p_r = json.load(open(os.path.join(GOLD, "fisher_R_greater.json")))["p"]
    for i in range(11, 100):
        assert abs(p_r[i] - orc.fisher(i, 1000, 10, 1000)) <= p_r[i] / 1e5


def test_golden_vectors_bit_exact():
    cases = json.load(open(os.path.join(GOLD, "fisher_ref.json")))["cases"]
    assert len(cases) > 300
    for a, b, c, d, p_hex in cases:
        assert orc.fisher(a, b, c, d) == float.fromhex(p_hex), (a, b, c, d)


def test_quirks_are_kept():
    # table edge -> point probability only (src/Fisher.cpp:42); saddle -> exactly 1 (:71-74)
    assert orc.fisher(0, 1000, 10, 1000) == pytest.approx(0.0010036466075338858, rel=1e-15)
    for i in range(1, 11):
        assert orc.fisher(i, 1000, 10, 1000) == 1.0
    assert orc.fisher(10000, 990000, 174, 999827) == 0.0


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (reference absent)")
def test_against_compiled_reference():
    r = orc.ref()
    import random
    rng = random.Random(5)
    for _ in range(3000):
        cov = rng.choice([10, 100, 1000, 6000, 100000, 2000000])
        a = rng.randint(0, cov)
        c = rng.randint(0, max(1, cov // rng.choice([1, 10, 1000])))
        t = (a, cov - a, c, cov - c + rng.randint(0, 1))
        assert orc.fisher(*t) == r.ref_fisher_exact_tiss(*t), t


def test_pymodel_agrees():
    cases = json.load(open(os.path.join(GOLD, "fisher_ref.json")))["cases"]
    for a, b, c, d, p_hex in cases:
        if min(b, c) > 20000:
            continue  # pure-Python loop: keep it short
        assert pymodel.fisher_exact_tiss(a, b, c, d) == float.fromhex(p_hex), (a, b, c, d)
