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


def pack_plain(reads):
    """The packed read format of include/msq_abi.h in its PLAIN layout (base j of a group in slot j, no
    MSQ_READS_NORMALIZED flag), written from the header's text alone: what a third-party producer may hand to
    msq_pileup.  Returns the dict layout of minorseq_amd.device.pack_host()."""
    n = reads["n_reads"]
    off = reads["base_off"]
    group_off = np.zeros(n + 1, dtype=np.uint64)
    for r in range(n):
        group_off[r + 1] = group_off[r] + (int(off[r + 1] - off[r]) + 31) // 32
    g = int(group_off[n])
    bop = np.zeros((g, 4), dtype=np.uint32)
    planes = {f: (np.zeros((g, 32), dtype=np.uint8) if reads.get(f) is not None else None)
              for f in ("qual", "delqv", "subqv", "insqv")}
    exotic = False
    for r in range(n):
        for i in range(int(off[r]), int(off[r + 1])):
            j = i - int(off[r])
            grp, k = int(group_off[r]) + j // 32, j % 32
            c, nt = chr(reads["cigar"][i]), chr(reads["nuc"][i])
            if c in "=X":
                if nt in "ACGT":
                    op, code = 0, "ACGT".index(nt)
                elif nt == "N":
                    op, code = 1, 1
                elif nt == "-":
                    op, code, exotic = 1, 3, True
                else:
                    op, code, exotic = 1, 2, True
            elif c == "D":
                op, code = 1, 0
            elif c == "I":
                op, code = 2, "ACGT".index(nt) if nt in "CGT" else 0
            else:
                assert c in "PS"
                op, code = 3, 0
            for plane, bit in enumerate((code & 1, code >> 1, op & 1, op >> 1)):
                bop[grp, plane] |= np.uint32(bit << k)
            for f, a in planes.items():
                if a is not None:
                    a[grp, 4 * (k & 7) + (k >> 3)] = reads[f][i]  # MSQ_QV_BYTE
        last, used = int(group_off[r + 1]) - 1, int(off[r + 1] - off[r]) % 32
        if used and last >= int(group_off[r]):  # padding after the last base: op 3
            pad = np.uint32((0xFFFFFFFF << used) & 0xFFFFFFFF)
            bop[last, 2] |= pad
            bop[last, 3] |= pad
    out = {"n_reads": n, "n_groups": g, "ref_start": reads["ref_start"].copy(), "ref_end": reads["ref_end"].copy(),
           "group_off": group_off, "bop": bop.reshape(-1), "flags": 1 if exotic else 0}
    out.update({f: (a.reshape(-1) if a is not None else None) for f, a in planes.items()})
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
