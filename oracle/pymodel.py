"""Literal pure-Python model of juliet's pileup + caller + Fisher for SMALL cases.  TEST INFRASTRUCTURE ONLY.

Written independently of oracle/oracle.cpp (strings, dicts and lists where the reference has std::string,
std::map and std::vector) so the two restatements can be checked against each other; the pileup parity of
both is otherwise unpinned (see the header of oracle/oracle.cpp).  Citations: file:line under the reference.
"""
import ctypes
import ctypes.util
import math

_libm = ctypes.CDLL(ctypes.util.find_library("m"))
_libm.lgamma.restype = ctypes.c_double
_libm.lgamma.argtypes = [ctypes.c_double]
_libm.exp.restype = ctypes.c_double
_libm.exp.argtypes = [ctypes.c_double]

BASES = "ACGT"
CODONS = [a + b + c for a in BASES for b in BASES for c in BASES]  # AminoAcidTable keys, src/AminoAcidTable.cpp:43-54


def factln(n):  # src/Fisher.cpp:115-125
    if n <= 1:
        return 0.0
    return _libm.lgamma(float(n) + 1.0)


def binomialln(n, k):  # src/Fisher.cpp:127
    return factln(n) - factln(k) - factln(n - k)


def fisher_exact_tiss(chi11, chi12, chi21, chi22):  # src/Fisher.cpp:12-77
    co_occ = chi11
    gene_a, gene_b = chi11 + chi12, chi11 + chi21
    total = chi11 + chi12 + chi21 + chi22
    min_co = gene_a + gene_b - total if gene_a + gene_b > total else 0
    max_co = gene_a if gene_a < gene_b else gene_b
    base_p = _libm.exp(binomialln(chi11 + chi12, chi11) + binomialln(chi21 + chi22, chi21) - binomialln(total, chi11 + chi21))
    if not (co_occ == max_co or co_occ == min_co):
        inc = (float(chi12) * chi21) / (float(chi11 + 1) * (chi22 + 1))
        dec = (float(chi11) * chi22) / (float(chi21 + 1) * (chi12 + 1))
        curr = base_p
        if inc < dec:
            while True:
                curr *= inc
                base_p += curr
                co_occ += 1
                chi11 += 1
                chi22 += 1
                chi12 -= 1
                chi21 -= 1
                inc = (float(chi12) * chi21) / (float(chi11 + 1) * (chi22 + 1))
                if not co_occ < max_co:
                    break
        else:
            base_p = 1.0
    return base_p


def _meets(thr, qv, i):  # ArrayRead.inl:72-87
    return thr < 0 or qv is None or qv[i] >= thr


class MSAByRow:  # src/MSA.cpp:94-124
    def __init__(self, reads, thresholds=(-1, -1, -1, -1)):
        """reads: dict of numpy arrays (synth.Workload.host_reads layout); thresholds: qual, del, sub, ins."""
        tq, td, ts, ti = thresholds
        n = reads["n_reads"]
        begin, end = 2 ** 31 - 1, 0
        for r in range(n):  # UpdateBoundaries, :138-142
            begin = min(begin, int(reads["ref_start"][r]))
            end = max(end, int(reads["ref_end"][r]))
        self.rows, self.insertions = [], []
        width = end - begin if n else 0
        for r in range(n):  # AddRead, :144-186
            row = [" "] * width
            ins_map = {}
            pos = int(reads["ref_start"][r]) - begin
            insertion = ""
            for i in range(int(reads["base_off"][r]), int(reads["base_off"][r + 1])):
                cg, nt = chr(reads["cigar"][i]), chr(reads["nuc"][i])
                if cg in "X=":
                    if insertion:
                        ins_map[pos] = insertion
                        insertion = ""
                    ok = (_meets(tq, reads.get("qual"), i) and _meets(td, reads.get("delqv"), i)
                          and _meets(ts, reads.get("subqv"), i) and _meets(ti, reads.get("insqv"), i))
                    row[pos] = nt if ok else "N"
                    pos += 1
                elif cg == "D":
                    if insertion:
                        ins_map[pos] = insertion
                        insertion = ""
                    row[pos] = "-"
                    pos += 1
                elif cg == "I":
                    insertion += nt
                elif cg in "PS":
                    if insertion:
                        ins_map[pos] = insertion
                        insertion = ""
                else:
                    raise RuntimeError("Unexpected cigar %d" % ord(cg))
            self.rows.append(row)
            self.insertions.append(ins_map)
        self.begin_pos = begin + 1  # :107-108
        self.end_pos = end + 1
        self.width = width

    def codon_at(self, row, pos):  # MSARow::CodonAt, :188-195
        return "".join(row[i] for i in range(pos, pos + 3) if 0 < i < len(row))

    def coding_codon_at(self, row, win_pos):  # :197-219
        if win_pos + 2 >= len(row) or win_pos < 0:
            return None
        trio = row[win_pos:win_pos + 3]
        if " " in trio or "-" in trio:
            return None
        proposed = self.codon_at(row, win_pos)
        return proposed if proposed in CODONS else None

    def codons_at(self, i):  # :126-136 ; returned sorted like std::map
        out = {}
        for row in self.rows:
            c = self.coding_codon_at(row, i)
            if c is not None:
                out[c] = out.get(c, 0) + 1
        return dict(sorted(out.items()))


def column_counts(msa):  # MSAByColumn ctor, src/MSA.cpp:55-92
    tag = {"A": 0, "C": 1, "G": 2, "T": 3, "-": 4, "N": 5}
    counts = [[0] * 6 for _ in range(msa.width)]
    for row in msa.rows:
        for c, ch in enumerate(row):
            if ch == " ":
                continue
            if ch not in tag:
                raise RuntimeError("Unexpected base " + ch)
            counts[c][tag[ch]] += 1
    return counts


def call_variants(msa, genes, reference, match, substitution, deletion, maximal_perc=100.0):
    """CountNumberOfTests + the cell loop of CallVariants (src/AminoAcidCaller.cpp:86-103, 397-497).
    genes: list of (begin, end).  Returns (number_of_tests, list of position dicts each with a 'cells' list)."""
    if not genes:
        genes = [(msa.begin_pos, msa.end_pos)]
    tests = 0
    for begin, end in genes:
        for i in range(begin, end - 2):
            if (i - begin) % 3 == 0:
                tests += len(msa.codons_at(i - msa.begin_pos))
    tests = tests or 1
    alpha = 0.009999999776482582  # (double)(float)0.01, AminoAcidCaller.h:108

    def probability(a, b):  # :325-339
        p = 1.0
        for x, y in zip(a, b):
            p *= deletion if "-" in (x, y) else (substitution if x != y else match)
        return p

    out = []
    for g, (begin, end) in enumerate(genes):
        for i in range(begin, end - 2):
            if (i - begin) % 3:
                continue
            codons = msa.codons_at(i - msa.begin_pos)
            coverage = sum(codons.values())
            maj, maj_cnt = "", -1
            for k, v in codons.items():
                if v > maj_cnt:
                    maj, maj_cnt = k, v
            if maj not in CODONS:
                maj, maj_cnt = "", 0
            rec = {"gene": g, "i": i, "aa_pos": 1 + (i - begin) // 3, "coverage": coverage, "maj": maj,
                   "maj_count": maj_cnt, "ref": "", "alt": "", "skipped": False, "cells": []}
            out.append(rec)
            if reference:
                ref = reference[i - 1:i + 2]
                if ref not in CODONS:
                    rec["skipped"] = True
                    continue
                rec["ref"] = ref
                if maj_cnt == 0:
                    rec["skipped"] = True
                    continue
                if maj_cnt * 100.0 / coverage > maximal_perc:
                    rec["alt"] = maj
            else:
                if maj_cnt == 0:
                    rec["skipped"] = True
                    continue
                rec["ref"] = maj
            for codon, cnt in codons.items():
                if codon == rec["ref"] or (rec["alt"] and codon == rec["alt"]):
                    continue
                expected = coverage * probability(rec["ref"], codon)
                p = fisher_exact_tiss(cnt, coverage - cnt, int(math.ceil(expected)), int(math.ceil(coverage - expected))) * tests
                if p > 1:
                    p = 1
                freq = 1.0 * cnt / coverage
                rec["cells"].append({"codon": codon, "count": cnt, "expected": expected, "p": p, "frequency": freq,
                                     "variable": freq < 0.8, "significant": p < alpha})
    return tests, out


def phase_variants(msa, pos, hits):
    """AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:126-323) up to the ordering of equal-sized
    haplotypes: exact-tuple clustering in read order, Haplotype flags (src/Haplotype.cpp:50-57), LOW_COV below 10
    reads, generators = flag-free haplotypes.  Returns (generators, filtered, counts) with haplotypes as dicts
    {codons, reads, flags}; counts = [genCounts, lowCov, margOfftarget, margWithGap, margWithHetero, margPartial]."""
    obs, index = [], {}
    for r, row in enumerate(msa.rows):
        codons = tuple(msa.codon_at(row, p) for p in pos)
        flag = 0
        for k, c in enumerate(codons):
            if c not in hits[k]:
                flag = 16  # OFFTARGET
        if codons in index:
            index[codons]["reads"].append(r)
            continue
        h = {"codons": list(codons), "reads": [r], "flags": flag}
        for c in codons:
            if "-" in c:
                h["flags"] |= 1
            if "N" in c:
                h["flags"] |= 2
            if " " in c:
                h["flags"] |= 4
        index[codons] = h
        obs.append(h)
    gen, fil = [], []
    for h in obs:
        if len(h["reads"]) < 10:
            h["flags"] |= 8
        (gen if h["flags"] == 0 else fil).append(h)
    by_flags = {}
    for h in fil:
        by_flags[h["flags"]] = by_flags.get(h["flags"], 0) + len(h["reads"])
    counts = [sum(len(h["reads"]) for h in gen), by_flags.get(8, 0),
              sum(v for f, v in by_flags.items() if f & 16), sum(v for f, v in by_flags.items() if f & 1),
              sum(v for f, v in by_flags.items() if f & 2), sum(v for f, v in by_flags.items() if f & 4)]
    return gen, fil, counts
