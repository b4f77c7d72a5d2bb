#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own code compiled into oracle/_ref (make -C oracle ref).

Run in the build container only (needs /root/reference); the JSON outputs are committed so the GPU box and
the CPU test tier can check against them without the reference.  Doubles are stored as C99 hex floats so the
vectors are bit-exact.

  fisher_ref.json        fisher_exact_tiss outputs of the reference's src/Fisher.cpp (unit-test family,
                         table-edge and "saddle point" quirks, juliet-shaped tables up to 1e7 coverage)
  fisher_R_greater.json  the R fisher.test(alternative="greater") table the reference's own unit test holds
                         (tests/unit/FisherTest.cpp:53-70), data only
  tables_ref.json        AminoAcidTable::FromCodon, ErrorEstimates and NucleotideToTag as the compiled
                         reference reports them
"""
import json
import math
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("MINORSEQ_REF", "/root/reference")


def fisher_cases():
    cases = [(i, 1000, 10, 1000) for i in range(0, 101)]  # the reference unit-test family incl. i = 0..10
    # table edges / degenerate tables
    cases += [(0, 0, 0, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1), (1, 1, 1, 1), (5, 0, 0, 5),
              (0, 5, 5, 0), (3, 0, 2, 7), (2, 7, 3, 0), (10, 10, 10, 10), (1, 100, 1, 100), (100, 1, 1, 100)]
    # juliet-shaped: (cnt, cov-cnt, ceil(expected), ceil(cov-expected)) with expected = cov * M^2 S etc.
    M, S = 0.9956844883, 0.0005244257 / 3.0
    for cov in (50, 298, 1000, 6000, 25000, 100000, 1000000, 10000000):
        for pr in (M * M * S, M * S * S, S * S * S):
            exp = cov * pr
            c, d = int(math.ceil(exp)), int(math.ceil(cov - exp))
            for cnt in sorted({1, 2, 3, 5, 10, 30, 60, max(1, cov // 1000), max(1, cov // 100), max(1, cov // 10),
                               max(1, cov // 2), cov - 1, cov}):
                if 0 < cnt <= cov:
                    cases.append((cnt, cov - cnt, c, d))
    return cases


def main():
    r = orc.ref()
    out = []
    for a, b, c, d in fisher_cases():
        out.append([a, b, c, d, float(r.ref_fisher_exact_tiss(a, b, c, d)).hex()])
    with open(os.path.join(GOLD, "fisher_ref.json"), "w") as f:
        json.dump({"source": "reference src/Fisher.cpp compiled unmodified (oracle/_ref), glibc lgamma/exp of this image",
                   "columns": ["chi11", "chi12", "chi21", "chi22", "p_hex"], "cases": out}, f, indent=0)

    src = open(os.path.join(REF, "tests", "unit", "FisherTest.cpp")).read()
    body = src[src.index("pValuesFromR{") + len("pValuesFromR{"):]
    body = body[:body.index("}")]
    vals = [float(x) for x in re.findall(r"[-+0-9.eE]+", body)]
    assert len(vals) == 101, len(vals)
    with open(os.path.join(GOLD, "fisher_R_greater.json"), "w") as f:
        json.dump({"source": "R fisher.test(matrix(c(x,1000,10,1000)), alternative='greater'), x = 0..100; "
                             "values held by the reference's tests/unit/FisherTest.cpp:53-70; the reference "
                             "checks x = 11..99 with tolerance p/1e5",
                   "p": vals}, f, indent=0)

    codons = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT"]
    import ctypes as C
    import numpy as np
    em = np.zeros(4)
    tabs = {"amino_acid": {c: r.ref_amino_acid(c.encode()).decode() for c in codons},
            "amino_acid_table_size": r.ref_amino_acid_table_size(),
            "not_codons": {s: (r.ref_amino_acid(s.encode()) == b"\x00") for s in ("AAN", "A-A", "NNN", "AA ", "aaa")},
            "nucleotide_to_tag": {ch: r.ref_nucleotide_to_tag(ch.encode()) for ch in "ACGT-N"}}
    r.ref_error_estimates_chemistry(b"S/P2-C2", em.ctypes.data)
    tabs["error_estimates_chemistry"] = [float(x).hex() for x in em]
    r.ref_error_estimates_rates(0.001, 0.005, em.ctypes.data)
    tabs["error_estimates_rates_0.001_0.005"] = [float(x).hex() for x in em]
    with open(os.path.join(GOLD, "tables_ref.json"), "w") as f:
        json.dump({"source": "reference AminoAcidTable.cpp / ErrorEstimates.cpp / NucleotideConversion.cpp compiled "
                             "unmodified (oracle/_ref)", **tabs}, f, indent=0)
    print("fisher cases:", len(out))


if __name__ == "__main__":
    main()
