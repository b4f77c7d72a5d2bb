"""Synthetic CCS amplicon workloads (harness side of include/msq_synth.h).

The five workloads realise BASELINE.json's configs as SURVEY.md section 8d lays them out.  Reads come from
the library's deterministic generator: msq_synth_host_* gives the unpacked (ArrayBase-like) form the oracle
and the packer take, msq_synth_dev_* writes the packed form straight into HBM.
"""
import json
import os

import numpy as np

from . import abi

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_M64 = (1 << 64) - 1


def hxb2():
    with open(os.path.join(_DATA, "hxb2.fasta")) as f:
        return "".join(l.strip() for l in f if not l.startswith(">"))


def hiv_genes():
    with open(os.path.join(_DATA, "hiv_genes.json")) as f:
        return json.load(f)["genes"]


def syn_hash(seed, read, col, stream):
    """Python twin of syn_hash() in csrc/msq_synth_model.h."""
    x = (seed + 0x9E3779B97F4A7C15 * (read + 1)) & _M64
    x ^= ((col + 16) * 0xBF58476D1CE4E5B9) & _M64
    x = (x + (stream + 1) * 0x94D049BB133111EB) & _M64
    for _ in range(2):
        x ^= x >> 30
        x = (x * 0xBF58476D1CE4E5B9) & _M64
        x ^= x >> 27
        x = (x * 0x94D049BB133111EB) & _M64
        x ^= x >> 31
    return x


def _thr(p):
    return min(int(p * 4294967296.0), 0xFFFFFFFF)


class Workload:
    """One synthetic data set: window, genes, reference, clone sequences and generator parameters."""

    def __init__(self, name, n_reads, win_begin0, W, genes, reference, seed=42, sub=0.0005244257, dele=0.003791086,
                 ins=0.001, n_minors=2, minor_frac=0.01, minor_codons=10, jitter_frac=0.0, max_jitter=0, rich_qv=False,
                 use_reference=True):
        self.name = name
        self.n_reads = int(n_reads)
        self.win_begin0 = int(win_begin0)
        self.W = int(W)
        self.genes = genes  # list of {name, begin, end} (1-based, TargetConfig coordinates) or []
        self.reference = reference  # full reference string in TargetConfig coordinates
        self.use_reference = use_reference  # hand referenceSequence to the caller (a config is present)
        self.seed = seed
        self.rich_qv = bool(rich_qv)
        self.params = abi.SynthParams(seed, self.win_begin0, self.W, _thr(sub), _thr(dele), _thr(ins), _thr(minor_frac),
                                      n_minors, _thr(jitter_frac), max_jitter, 1 if rich_qv else 0)
        self.minor_sites = []  # (minor, 1-based codon start i, new codon string)
        self.clone_codes = self._clones(n_minors, minor_codons)

    def _clones(self, n_minors, minor_codons):
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        win = self.reference[self.win_begin0:self.win_begin0 + self.W]
        assert len(win) == self.W, "window outside the reference"
        major = np.array([code[c] for c in win], dtype=np.uint8)
        clones = np.tile(major, (1 + n_minors, 1))
        # codon starts (0-based window columns) inside the tested genes and fully inside the window
        genes = self.genes or [{"begin": self.win_begin0 + 1, "end": self.win_begin0 + 1 + self.W}]
        starts = []
        for g in genes:
            for i in range(g["begin"], g["end"] - 2, 3):
                c = i - 1 - self.win_begin0
                if 1 <= c and c + 2 < self.W:
                    starts.append((i, c))
        for m in range(1, n_minors + 1):
            used = set()
            j = 0
            while len([s for s in self.minor_sites if s[0] == m]) < min(minor_codons, len(starts)):
                h = syn_hash(self.seed, 1 << 40 | m, j, 9)
                j += 1
                i, c = starts[h % len(starts)]
                if any(abs(c - u) < 3 for u in used):
                    continue
                used.add(c)
                n_sub = 1 + (h >> 32) % 3  # 1..3 nucleotides differ
                new = clones[m, c:c + 3].copy()
                for k in range(n_sub):
                    pos = ((h >> 40) + k) % 3
                    new[pos] = (new[pos] + 1 + ((h >> (48 + 2 * k)) % 3)) & 3
                clones[m, c:c + 3] = new
                self.minor_sites.append((m, i, "".join("ACGT"[x] for x in new)))
        return np.ascontiguousarray(clones)

    # ---------------------------------------------------------------- host (unpacked) reads
    def host_reads(self, first=0, n=None):
        """Unpacked reads [first, first+n) as numpy arrays (the oracle's and msq_pack_reads' input)."""
        n = self.n_reads - first if n is None else n
        L = abi.lib()
        lens = np.zeros(n, dtype=np.int64)
        abi.check(L.msq_synth_host_lengths(self.params, first, n, abi.ptr(lens)))
        base_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=base_off[1:])
        tot = int(base_off[-1])
        r = {
            "n_reads": n,
            "base_off": base_off,
            "ref_start": np.zeros(n, dtype=np.int32),
            "ref_end": np.zeros(n, dtype=np.int32),
            "cigar": np.zeros(tot, dtype=np.uint8),
            "nuc": np.zeros(tot, dtype=np.uint8),
            "qual": np.zeros(tot, dtype=np.uint8),
            "subqv": np.zeros(tot, dtype=np.uint8) if self.rich_qv else None,
            "delqv": np.zeros(tot, dtype=np.uint8) if self.rich_qv else None,
            "insqv": np.zeros(tot, dtype=np.uint8) if self.rich_qv else None,
        }
        abi.check(L.msq_synth_host_fill(self.params, abi.ptr(self.clone_codes), first, n, abi.ptr(base_off),
                                        abi.ptr(r["ref_start"]), abi.ptr(r["ref_end"]), abi.ptr(r["cigar"]),
                                        abi.ptr(r["nuc"]), abi.ptr(r["qual"]), abi.ptr(r["subqv"]), abi.ptr(r["delqv"]),
                                        abi.ptr(r["insqv"])))
        return r

    def gene_arrays(self):
        gb = np.array([g["begin"] for g in self.genes], dtype=np.int32)
        ge = np.array([g["end"] for g in self.genes], dtype=np.int32)
        return gb, ge

    def algorithmic_bytes(self, n_reads=None, n_bases=None):
        """1.5 B per base (4.5 with rich QVs) + 12 B per read; n_bases ~ 1.001 * W per read."""
        n = self.n_reads if n_reads is None else n_reads
        per_base = 4.5 if self.rich_qv else 1.5
        nb = n * 1.001 * self.W if n_bases is None else n_bases
        return nb * per_base + 12.0 * n


_POL = ("Protease", "Reverse Transcriptase", "RNase", "Integrase")


def _random_reference(length, seed):
    return "".join("ACGT"[syn_hash(seed, 1 << 41, i, 8) & 3] for i in range(length))


def workload(name, n_reads=None, **kw):
    """BASELINE.json configs: hiv_1k (#1), pol_100k (#2), pol_1m (#3), hiv_mix_200k (#4), amp20k_10m (#5)."""
    ref = hxb2()
    genes = hiv_genes()
    pol = [g for g in genes if g["name"] in _POL]
    if name == "hiv_1k":
        return Workload(name, n_reads or 1000, 0, 9719, genes, ref, jitter_frac=0.10, max_jitter=150, **kw)
    if name == "pol_100k":  # juliet --region 2253-5096 -c HIV
        return Workload(name, n_reads or 100000, 2252, 2843, pol, ref, **kw)
    if name == "pol_1m":
        return Workload(name, n_reads or 1000000, 2252, 2843, pol, ref, **kw)
    if name == "hiv_mix_200k":
        kw.setdefault("minor_frac", 0.001)
        return Workload(name, n_reads or 200000, 0, 9719, genes, ref, **kw)
    if name == "amp20k_10m":
        return Workload(name, n_reads or 10000000, 0, 20000, [], _random_reference(20000, 42), use_reference=False, **kw)
    raise KeyError(name)
