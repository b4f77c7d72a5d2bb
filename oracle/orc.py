"""ctypes binding of oracle/liborc.so (the CPU restatement) and oracle/_ref/libmsq_ref.so (compiled reference
pieces).  TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from minorseq_amd/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liborc.so")
REF_LIB = os.path.join(_HERE, "_ref", "libmsq_ref.so")


class Position(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "gene", "i", "aa_pos", "win_pos", "coverage", "ref_codon", "alt_codon", "maj_codon", "maj_count", "skipped")]


class Cell(C.Structure):
    _fields_ = [("position", C.c_int32), ("codon", C.c_int32), ("count", C.c_int32), ("significant", C.c_int32),
                ("variable", C.c_int32), ("pad", C.c_int32), ("expected", C.c_double), ("p", C.c_double),
                ("frequency", C.c_double)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        P = C.c_void_p
        l.orc_fisher.restype = C.c_double
        l.orc_fisher.argtypes = [C.c_int] * 4
        l.orc_amino_acid.restype = C.c_char
        l.orc_amino_acid.argtypes = [C.c_int]
        l.orc_error_estimates_default.argtypes = [P]
        l.orc_error_estimates_rates.argtypes = [C.c_double, C.c_double, P]
        l.orc_msa_build.restype = P
        l.orc_msa_build.argtypes = [C.c_int64] + [P] * 10 + [C.POINTER(C.c_int)]
        l.orc_msa_free.argtypes = [P]
        for f in ("orc_msa_begin1", "orc_msa_end1", "orc_msa_width"):
            getattr(l, f).restype = C.c_int
            getattr(l, f).argtypes = [P]
        l.orc_msa_nuc_counts.argtypes = [P, P]
        l.orc_msa_rows.argtypes = [P, P]
        l.orc_msa_codon_counts.argtypes = [P, P]
        l.orc_col_coverage.restype = C.c_int
        l.orc_col_coverage.argtypes = [P, C.c_int]
        l.orc_col_maxbase.restype = C.c_char
        l.orc_col_maxbase.argtypes = [P, C.c_int]
        l.orc_col_insertions.restype = C.c_int
        l.orc_col_insertions.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        l.orc_row_insertions.restype = C.c_int
        l.orc_row_insertions.argtypes = [P, C.c_int64, C.c_char_p, C.c_int]
        l.orc_fuse.restype = C.c_int
        l.orc_fuse.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        l.orc_count_positions.restype = C.c_int64
        l.orc_count_positions.argtypes = [C.c_int, P, P]
        l.orc_call.restype = C.c_int64
        l.orc_call.argtypes = [P, C.c_int, P, P, C.c_char_p, P, C.c_double, P, P, P]
        l.orc_phase.restype = C.c_int64
        l.orc_phase.argtypes = [P, C.c_int, P, C.c_char_p, P, C.c_int64]
        _lib = l
    return _lib


def ref_available():
    return os.path.exists(REF_LIB)


def ref():
    """The reference's own Fisher / AminoAcidTable / ErrorEstimates / NucleotideConversion (oracle/_ref)."""
    global _ref
    if _ref is None:
        r = C.CDLL(REF_LIB)
        r.ref_fisher_exact_tiss.restype = C.c_double
        r.ref_fisher_exact_tiss.argtypes = [C.c_int] * 4
        r.ref_amino_acid.restype = C.c_char
        r.ref_amino_acid.argtypes = [C.c_char_p]
        r.ref_amino_acid_table_size.restype = C.c_int
        r.ref_error_estimates_chemistry.argtypes = [C.c_char_p, C.c_void_p]
        r.ref_error_estimates_rates.argtypes = [C.c_double, C.c_double, C.c_void_p]
        r.ref_nucleotide_to_tag.restype = C.c_int
        r.ref_nucleotide_to_tag.argtypes = [C.c_char]
        r.ref_tag_to_nucleotide.restype = C.c_char
        r.ref_tag_to_nucleotide.argtypes = [C.c_int]
        _ref = r
    return _ref


def fisher(a, b, c, d):
    return lib().orc_fisher(int(a), int(b), int(c), int(d))


def _p(a):
    return None if a is None else a.ctypes.data


def parse_phase_report(text):
    """GEN / FIL lines (name size flags frequency codons) each followed by a READS line, then CNT: -> dict."""
    out = {"gen": [], "fil": [], "cnt": None}
    cur = None
    for line in text.splitlines():
        t = line.split(" ")
        if t[0] in ("GEN", "FIL"):
            cur = {"name": t[1], "size": int(t[2]), "flags": int(t[3]), "frequency": float(t[4]),
                   "codons": [("" if c == "." else c.replace("_", " ")) for c in t[5].split(",")] if len(t) > 5 and t[5] else [],
                   "reads": []}
            out["gen" if t[0] == "GEN" else "fil"].append(cur)
        elif t[0] == "READS":
            cur["reads"] = [x for x in t[1:] if x]
        elif t[0] == "CNT":
            out["cnt"] = [int(x) for x in t[1:7]]
    return out


class Msa:
    """MSAByRow + MSAByColumn of the restatement, built from unpacked reads (dict of numpy arrays)."""

    def __init__(self, reads, thresholds=(-1, -1, -1, -1)):
        thr = np.array(thresholds, dtype=np.int32)  # qual, del, sub, ins
        err = C.c_int(0)
        self._keep = reads
        self.h = lib().orc_msa_build(reads["n_reads"], _p(reads["ref_start"]), _p(reads["ref_end"]),
                                     _p(reads["base_off"]), _p(reads["cigar"]), _p(reads["nuc"]), _p(reads.get("qual")),
                                     _p(reads.get("subqv")), _p(reads.get("delqv")), _p(reads.get("insqv")), _p(thr),
                                     C.byref(err))
        self.error = err.value
        if not self.h:
            raise RuntimeError("oracle msa_build failed with %d" % err.value)
        self.begin1 = lib().orc_msa_begin1(self.h)
        self.end1 = lib().orc_msa_end1(self.h)
        self.W = lib().orc_msa_width(self.h)

    def nuc_counts(self):
        out = np.zeros((self.W, 6), dtype=np.int32)
        if self.W:
            lib().orc_msa_nuc_counts(self.h, _p(out))
        return out

    def codon_counts(self):
        out = np.zeros((self.W, 64), dtype=np.int32)
        if self.W:
            lib().orc_msa_codon_counts(self.h, _p(out))
        return out

    def rows(self):
        n = self._keep["n_reads"]
        out = np.zeros((n, self.W), dtype=np.uint8)
        if n and self.W:
            lib().orc_msa_rows(self.h, _p(out))
        return out

    def call(self, gene_begin, gene_end, ref_seq, err_model=None, maximal_perc=100.0):
        gb = np.ascontiguousarray(gene_begin, dtype=np.int32)
        ge = np.ascontiguousarray(gene_end, dtype=np.int32)
        if len(gb) == 0:
            b1, e1 = np.array([self.begin1], np.int32), np.array([self.end1], np.int32)
            P = int(lib().orc_count_positions(1, _p(b1), _p(e1)))
        else:
            P = int(lib().orc_count_positions(len(gb), _p(gb), _p(ge)))
        if err_model is None:
            em = np.zeros(4)
            lib().orc_error_estimates_default(_p(em))
        else:
            em = np.ascontiguousarray(err_model, dtype=np.float64)
        pos = (Position * max(P, 1))()
        cells = (Cell * max(P * 64, 1))()
        nt = C.c_int32(0)
        ref = ref_seq.encode() if isinstance(ref_seq, str) and ref_seq else None
        n = lib().orc_call(self.h, len(gb), _p(gb) if len(gb) else None, _p(ge) if len(ge) else None, ref, _p(em),
                           float(maximal_perc), C.cast(pos, C.c_void_p), C.cast(cells, C.c_void_p), C.byref(nt))
        if n < 0:
            raise RuntimeError("oracle orc_call failed with %d" % n)
        pos_np = np.frombuffer(pos, dtype=np.int32).reshape(-1, 10)[:P].copy()
        return pos_np, [cells[k] for k in range(n)], nt.value

    def phase(self, pos, hits):
        """PhaseVariants on the rows: pos[k] = CodonAt argument of variant position k, hits[k] = codons IsHit accepts.
        Returns parse_phase_report() of the oracle's text report."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        text = ";".join("|".join(h) for h in hits).encode()
        cap = 1 << 20
        while True:
            buf = C.create_string_buffer(cap)
            need = lib().orc_phase(self.h, len(pos), _p(pos), text, C.cast(buf, C.c_void_p), cap)
            if need <= cap:
                return parse_phase_report(buf.value.decode())
            cap = int(need)

    def col_insertions(self, col):
        """MSAColumn::Insertions() of window column col: {string: count}."""
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_col_insertions(self.h, int(col), buf, cap)
            if n >= 0:
                break
            cap = -n + 1
        out = {}
        for line in buf.value.decode().splitlines():
            s, c = line.split("\t")
            out[s] = int(c)
        return out

    def row_insertions(self, row):
        """MSARow::Insertions of one row as {window column: string}."""
        cap = 1 << 12
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_row_insertions(self.h, int(row), buf, cap)
            if n >= 0:
                break
            cap = -n + 16
        return {int(t.split(":")[0]): t.split(":")[1] for t in buf.value.decode().split()}

    def fuse(self, min_coverage=50):
        """Fuse::CreateConsensus on this MSA (src/tools/Fuse.cpp:70-97)."""
        cap = self.W * 70 + 64
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_fuse(self.h, int(min_coverage), buf, cap)
            if n >= 0:
                return buf.value.decode()
            cap = -n + 1

    def close(self):
        if self.h:
            lib().orc_msa_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
