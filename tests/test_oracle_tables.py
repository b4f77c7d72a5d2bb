"""Codon table, error priors and tag order of the oracle against values captured from the reference's
compiled AminoAcidTable.cpp / ErrorEstimates.cpp / NucleotideConversion.cpp (tests/golden/tables_ref.json)."""
import json
import os

import numpy as np
import pytest

from oracle import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TABS = json.load(open(os.path.join(GOLD, "tables_ref.json")))
CODONS = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT"]


def test_amino_acid_table():
    assert TABS["amino_acid_table_size"] == 64
    for idx, codon in enumerate(CODONS):
        assert orc.lib().orc_amino_acid(idx).decode() == TABS["amino_acid"][codon], codon
    assert all(TABS["not_codons"].values())


def test_error_estimates():
    em = np.zeros(4)
    orc.lib().orc_error_estimates_default(em.ctypes.data)
    assert [float(x).hex() for x in em] == TABS["error_estimates_chemistry"]
    orc.lib().orc_error_estimates_rates(0.001, 0.005, em.ctypes.data)
    assert [float(x).hex() for x in em] == TABS["error_estimates_rates_0.001_0.005"]


def test_tag_order():
    assert TABS["nucleotide_to_tag"] == {"A": 0, "C": 1, "G": 2, "T": 3, "-": 4, "N": 5}


@pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (reference absent)")
def test_live_reference_tables():
    r = orc.ref()
    for idx, codon in enumerate(CODONS):
        assert r.ref_amino_acid(codon.encode()) == orc.lib().orc_amino_acid(idx)
    for t, ch in enumerate("ACGT-N"):
        assert r.ref_tag_to_nucleotide(t).decode() == ch
