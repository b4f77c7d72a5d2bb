"""Hand-made read sets for the parity tests: every edge the reference's row builder has
(src/MSA.cpp:144-219) -- deletions, insertions (leading, inner, trailing, multi-base), staggered starts,
reads shorter than the window, the first-window-column codon, majority ties, QV masks, P/S ops, literal N."""
import numpy as np


def make_reads(reads, rich=False, qual=True):
    """reads: list of (ref_start, cigar_string, nuc_string[, qual list[, sub, del, ins lists]]).
    ref_end = ref_start + number of column-consuming ops unless given as a dict key 'ref_end'."""
    n = len(reads)
    base_off = np.zeros(n + 1, dtype=np.int64)
    ref_start = np.zeros(n, dtype=np.int32)
    ref_end = np.zeros(n, dtype=np.int32)
    cig, nuc, q, s, d, i = [], [], [], [], [], []
    for k, r in enumerate(reads):
        start, c, nt = r[0], r[1], r[2]
        assert len(c) == len(nt)
        ref_start[k] = start
        ref_end[k] = start + sum(1 for x in c if x in "=XD")
        base_off[k + 1] = base_off[k] + len(c)
        cig += list(c.encode())
        nuc += list(nt.encode())
        q += list(r[3]) if len(r) > 3 and r[3] is not None else [60] * len(c)
        s += list(r[4]) if len(r) > 4 else [60] * len(c)
        d += list(r[5]) if len(r) > 5 else [60] * len(c)
        i += list(r[6]) if len(r) > 6 else [60] * len(c)
    out = {"n_reads": n, "base_off": base_off, "ref_start": ref_start, "ref_end": ref_end,
           "cigar": np.array(cig, dtype=np.uint8), "nuc": np.array(nuc, dtype=np.uint8),
           "qual": np.array(q, dtype=np.uint8) if qual else None,
           "subqv": np.array(s, dtype=np.uint8) if rich else None,
           "delqv": np.array(d, dtype=np.uint8) if rich else None,
           "insqv": np.array(i, dtype=np.uint8) if rich else None}
    return out


def random_reads(seed, n_reads, W, start0=100, rich=False, p_sub=0.02, p_del=0.03, p_ins=0.03, stagger=True,
                 exotic=False):
    """Noisy reads over a random reference with heavy indels, staggered/short reads and runs of insertions."""
    rng = np.random.RandomState(seed)
    ref = rng.randint(0, 4, size=W)
    reads = []
    for _ in range(n_reads):
        a, b = 0, W
        if stagger and rng.rand() < 0.5:
            a = rng.randint(0, max(1, W // 2))
            b = rng.randint(a + 1, W + 1)
        c, nt = [], []
        if rng.rand() < 0.1:
            c.append("S"); nt.append("A")
        for col in range(a, b):
            if col > a and rng.rand() < p_ins:
                for _k in range(rng.randint(1, 5)):
                    c.append("I"); nt.append("ACGT"[rng.randint(4)])
            u = rng.rand()
            if u < p_del:
                c.append("D"); nt.append("-")
            elif u < p_del + p_sub:
                c.append("X"); nt.append("ACGT"[(ref[col] + 1 + rng.randint(3)) % 4])
            else:
                c.append("="); nt.append("ACGT"[ref[col]])
            if exotic and rng.rand() < 0.01:
                nt[-1] = "N" if c[-1] != "D" else nt[-1]
            if rng.rand() < 0.01:
                c.append("P"); nt.append("*")
        if rng.rand() < 0.2:  # trailing insertion: dropped by the reference
            c.append("I"); nt.append("G")
        L = len(c)
        reads.append((start0 + a, "".join(c), "".join(nt), rng.randint(0, 94, size=L), rng.randint(0, 94, size=L),
                      rng.randint(0, 94, size=L), rng.randint(0, 94, size=L)))
    # make sure the window is [start0, start0+W)
    reads.append((start0, "=" * W, "".join("ACGT"[x] for x in ref), rng.randint(0, 94, size=W), rng.randint(0, 94, size=W),
                  rng.randint(0, 94, size=W), rng.randint(0, 94, size=W)))
    return make_reads(reads, rich=rich)


def edge_sets():
    """name -> (reads dict, thresholds (qual, del, sub, ins))."""
    none = (-1, -1, -1, -1)
    sets = {}
    sets["single_read"] = (make_reads([(10, "=" * 9, "ACGTACGTA")]), none)
    sets["two_codons_tie"] = (make_reads([(0, "=" * 7, "AACGTTT"), (0, "=" * 7, "ACCGTTT"), (0, "=" * 7, "AACGTTT"),
                                          (0, "=" * 7, "ACCGTTT")]), none)
    sets["deletion_and_insertions"] = (make_reads([
        (5, "==D==I==", "AC-GTAGG"), (5, "I======", "TACGTGG"), (5, "===III===I", "ACGTTTTGGA"), (6, "====", "CGTG"),
        (5, "=X====", "ATGTGG")]), none)
    sets["staggered_short"] = (make_reads([
        (100, "=" * 30, "ACGT" * 7 + "AC"), (110, "=" * 10, "GTACGTACGT"), (125, "=" * 5, "CGTAC"), (100, "=" * 3, "ACG"),
        (127, "===", "TAC")]), none)
    sets["pad_softclip"] = (make_reads([(0, "S==P==IS", "AACTGTAC"), (0, "====", "ACGT")]), none)
    sets["literal_N"] = (make_reads([(0, "======", "ACNTGA"), (0, "======", "ACGTGA")]), none)
    q = [10, 50, 10, 50, 50, 50]
    sets["qualqv_mask"] = (make_reads([(0, "======", "ACGTGA", q), (0, "======", "ACGTGA")]), (20, -1, -1, -1))
    sets["rich_default_subqv42"] = (make_reads([(0, "=D====", "A-GTGA", [50] * 6, [41, 0, 42, 43, 10, 90], [5] * 6, [5] * 6),
                                                (0, "======", "ACGTGA", [50] * 6, [42] * 6, [0] * 6, [0] * 6)], rich=True),
                                    (-1, -1, 42, -1))
    sets["rich_all_thresholds"] = (random_reads(7, 40, 90, rich=True), (30, 20, 42, 10))
    sets["plain_subqv42_is_noop"] = (random_reads(8, 30, 60, rich=False), (-1, -1, 42, -1))
    sets["random_indels"] = (random_reads(1, 200, 150), none)
    sets["random_wide"] = (random_reads(2, 60, 700, stagger=True), none)
    sets["random_literalN"] = (random_reads(3, 80, 120, exotic=True), (40, -1, -1, -1))
    sets["group_boundaries"] = (make_reads([(0, "=" * 31, "A" * 31), (0, "=" * 32, "C" * 32), (0, "=" * 33, "G" * 33),
                                            (0, "=" * 64, "T" * 64), (0, "=" * 65, "ACGTA" * 13)]), none)
    sets["threshold_zero_and_255"] = (random_reads(9, 20, 50, rich=True), (0, 255, 0, 0))
    return sets
