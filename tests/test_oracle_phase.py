"""Phasing: the C++ restatement of AminoAcidCaller::PhaseVariants (oracle/oracle.cpp, orc_phase)
against the independent Python model (oracle/pymodel.phase_variants), incl. the CodonAt edge rule (indices <= 0
and >= width are left out of the string) and the Haplotype flags.  Parity unpinned: the reference holds no
fixture for phasing."""
import numpy as np
import pytest

import cases
from oracle import orc, pymodel


def phase_case(seed, n_reads, W, p_sub=0.004, p_del=0.004, thr=(-1, -1, -1, -1), rich=False, stagger=True):
    reads = cases.random_reads(seed, n_reads, W, rich=rich, p_sub=p_sub, p_del=p_del, p_ins=0.01, stagger=stagger)
    rng = np.random.RandomState(seed)
    pos = sorted(set([-2, 0, 1, W - 3, W - 2, W - 1] + list(rng.randint(2, W - 4, size=6))))
    pm = pymodel.MSAByRow(reads, thr)
    hits = []
    for p in pos:  # the codons a caller would have stored: the two most frequent strings at the position
        seen = {}
        for row in pm.rows:
            c = pm.codon_at(row, p)
            seen[c] = seen.get(c, 0) + 1
        hits.append([c for c, _ in sorted(seen.items(), key=lambda kv: -kv[1])[:2]])
    return reads, thr, pos, hits, pm


@pytest.mark.parametrize("seed,n,W,kw", [
    (1, 400, 60, {}), (2, 900, 90, {"p_sub": 0.001, "p_del": 0.001}), (3, 300, 40, {"stagger": False}),
    (4, 500, 70, {"thr": (-1, -1, 42, -1), "rich": True}), (5, 50, 30, {}),
])
def test_oracle_phase_matches_python_model(seed, n, W, kw):
    reads, thr, pos, hits, pm = phase_case(seed, n, W, **kw)
    m = orc.Msa(reads, thr)
    rep = m.phase(pos, hits)
    gen, fil, counts = pymodel.phase_variants(pm, pos, hits)
    assert rep["cnt"] == counts
    key = lambda h: tuple(h["codons"])
    got = {key(h): h for h in rep["gen"] + rep["fil"]}
    want = {key(h): h for h in gen + fil}
    assert set(got) == set(want)
    for k, w in want.items():
        assert [int(x) for x in got[k]["reads"]] == w["reads"] and got[k]["flags"] == w["flags"]
    assert {key(h) for h in rep["gen"]} == {key(h) for h in gen}
    # generators: non-increasing sizes, names A, B, ... (Aa, Ab, ... beyond 26), frequencies over the generators
    sizes = [h["size"] for h in rep["gen"]]
    assert sizes == sorted(sizes, reverse=True)
    tot = sum(sizes)
    for g, h in enumerate(rep["gen"]):
        name = chr(65 + g // 26) + chr(97 + g % 26) if len(rep["gen"]) > 26 else chr(65 + g)
        assert h["name"] == name and h["frequency"] == h["size"] / tot
    fs = [h["size"] for h in rep["fil"]]
    assert fs == sorted(fs)  # filtered haplotypes: ascending (src/AminoAcidCaller.cpp:216)
    assert sum(len(h["reads"]) for h in rep["gen"] + rep["fil"]) == n + 1  # random_reads appends one full-span read
