"""The C++ restatement (oracle/oracle.cpp) against the independent literal Python model (oracle/pymodel.py) on
every hand-made edge case and on small synthetic workloads.  Pileup parity is otherwise UNPINNED: the reference
holds no MSA / AminoAcidCaller test or fixture.  CPU only."""
import numpy as np
import pytest

import cases
from minorseq_amd import synth
from oracle import orc, pymodel

EDGE = cases.edge_sets()
CODONS = pymodel.CODONS


def check_pileup(reads, thr):
    m = orc.Msa(reads, thr)
    pm = pymodel.MSAByRow(reads, thr)
    assert (m.begin1, m.end1, m.W) == (pm.begin_pos, pm.end_pos, pm.width)
    rows = m.rows()
    for r, row in enumerate(pm.rows):
        assert bytes(rows[r]).decode() == "".join(row), r
    assert (m.nuc_counts() == np.array(pymodel.column_counts(pm)).reshape(-1, 6)).all()
    cod = m.codon_counts()
    for i in range(m.W):
        d = pm.codons_at(i)
        want = np.zeros(64, dtype=np.int32)
        for k, v in d.items():
            want[CODONS.index(k)] = v
        assert (cod[i] == want).all(), i
    return m, pm


def check_call(m, pm, genes, reference, maximal_perc=100.0):
    gb = [g[0] for g in genes]
    ge = [g[1] for g in genes]
    pos, cells, nt = m.call(gb, ge, reference, maximal_perc=maximal_perc)
    tests, out = pymodel.call_variants(pm, genes, reference, 0.9956844883, 0.0005244257 / 3.0, 0.003791086, maximal_perc)
    assert nt == tests
    assert len(pos) == len(out)
    flat = []
    for k, o in enumerate(out):
        p = pos[k]
        assert (p[0], p[1], p[2], p[4]) == (o["gene"], o["i"], o["aa_pos"], o["coverage"])
        assert p[5] == (CODONS.index(o["ref"]) if o["ref"] else -1)
        assert p[6] == (CODONS.index(o["alt"]) if o["alt"] else -1)
        assert p[7] == (CODONS.index(o["maj"]) if o["maj"] else -1)
        assert bool(p[9]) == o["skipped"]
        flat += [(k, c) for c in o["cells"]]
    assert len(flat) == len(cells)
    for (k, c), oc in zip(flat, cells):
        assert (oc.position, oc.codon, oc.count) == (k, CODONS.index(c["codon"]), c["count"])
        assert oc.expected == c["expected"] and oc.p == c["p"] and oc.frequency == c["frequency"]
        assert bool(oc.variable) == c["variable"] and bool(oc.significant) == c["significant"]


@pytest.mark.parametrize("name", sorted(EDGE))
def test_edge_sets(name):
    reads, thr = EDGE[name]
    m, pm = check_pileup(reads, thr)
    check_call(m, pm, [], "")  # no config: one ORF over the window
    # a gene list in all three frames that also pokes outside the window, with and without a reference
    b = m.begin1
    genes = [(b, b + m.W), (b + 1, b + 40), (b + 2, b + m.W + 9), (max(1, b - 6), b + 12)]
    check_call(m, pm, genes, "")
    ref = "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(m.end1 + 60))
    check_call(m, pm, genes, ref)
    check_call(m, pm, genes, ref, maximal_perc=50.0)


def test_first_window_column_codon_is_never_counted():
    reads, thr = EDGE["single_read"]
    m = orc.Msa(reads, thr)
    cod = m.codon_counts()
    assert cod[0].sum() == 0 and cod[1].sum() == 1  # MSARow::CodonAt's i > 0 (src/MSA.cpp:193)
    assert cod[m.W - 2].sum() == 0 and cod[m.W - 1].sum() == 0


def test_trailing_insertion_dropped_and_errors():
    reads = cases.make_reads([(0, "===I", "ACGT")])
    m = orc.Msa(reads)
    buf = (b" " * 64)
    import ctypes
    b = ctypes.create_string_buffer(256)
    for c in range(m.W):
        assert orc.lib().orc_col_insertions(m.h, c, b, 256) == 0
    bad = cases.make_reads([(0, "==N=", "ACGT")])
    with pytest.raises(RuntimeError):
        orc.Msa(bad)  # "Unexpected cigar"
    badbase = cases.make_reads([(0, "===", "ARG")])
    with pytest.raises(RuntimeError):
        orc.Msa(badbase)  # "Unexpected base"


def test_empty_input():
    reads = cases.make_reads([])
    m = orc.Msa(reads)
    assert m.W == 0


@pytest.mark.parametrize("name,n", [("pol_100k", 120), ("hiv_1k", 25)])
def test_synthetic_workloads(name, n):
    wl = synth.workload(name, n_reads=n)
    reads = wl.host_reads(0, n)
    m, pm = check_pileup(reads, (-1, -1, -1, -1))
    genes = [(g["begin"], g["end"]) for g in wl.genes]
    check_call(m, pm, genes, wl.reference)
