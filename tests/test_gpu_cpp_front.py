"""GPU parity of the C++ front (minorseq_amd/cpp/msq_front.hpp): tests/cpp/front_test drives MSAByRow /
MSAByColumn / AminoAcidCaller / Fisher exactly as juliet's workflow drives the reference classes and dumps what it
computed; here the dump is checked against the oracle (counts bit-exact, numberOfTests, stored variants with the
store policy of src/AminoAcidCaller.cpp:499-539 restated below, p within 1e-9 relative) and against the
reference's own FisherTest.EquivalentWithR table.  The binary is prebuilt by __graft_entry__.build()."""
import json
import os
import subprocess

import numpy as np
import pytest

import cases
from oracle import orc, pymodel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "front_test")
CODONS = pymodel.CODONS
ALPHA = float(np.float32(0.01))  # AminoAcidCaller.h:108


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    assert os.path.exists(BIN), "tests/cpp/front_test not built: run __graft_entry__.build()"
    env = dict(os.environ)
    for k in ("DELQV", "SUBQV", "INSQV", "QUALQV"):
        env.pop(k, None)
    cwd = tmp_path_factory.mktemp("front")  # the expected-minors case writes validation.json into the working directory
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env, cwd=str(cwd))
    assert out.returncode == 0, out.stderr[-2000:]
    d = {"fisher": {}, "reads": {}, "cases": {}, "err": [], "ref": None, "phase": {}, "refx": None, "minors": []}
    vj = os.path.join(str(cwd), "validation.json")
    d["validation_json"] = open(vj).read() if os.path.exists(vj) else None
    d["stderr"] = out.stderr
    cur_reads = cur_case = cur_phase = None
    d["json"], d["msatable"], d["errmode"] = {}, {}, {}
    block = None  # (kind, name, lines) while inside a JSON / MSATABLE / ERRMODE block
    for line in out.stdout.splitlines():
        t = line.split(" ")
        if block is not None:
            if line == "END" + block[0]:
                d[block[0].lower()][block[1]] = "\n".join(block[2]) + "\n"
                block = None
            else:
                block[2].append(line)
            continue
        if t[0] in ("JSON", "MSATABLE", "ERRMODE"):
            block = (t[0], t[1], [])
            continue
        if t[0] == "JSONFORM":
            d["jsonform"] = line[len("JSONFORM "):]
            continue
        if t[0] == "FUSE":
            d.setdefault("fuse", {})[(t[1], int(t[2]))] = t[3] if len(t) > 3 else ""
            continue
        if t[0] == "INS":
            d.setdefault("ins", {}).setdefault(t[1], {})[int(t[2])] = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in t[3:]}
            continue
        if t[0] == "ROWS":
            cur_rows = d.setdefault("rows", {}).setdefault(t[1], {"n": int(t[2]), "W": int(t[3]), "rows": [], "check": None})
            continue
        if t[0] == "ROW":
            cur_rows["rows"].append((t[2] if len(t) > 2 else "", {int(x.split(":")[0]): x.split(":")[1] for x in t[3:]}))
            continue
        if t[0] == "ROWCHECK":
            cur_rows["check"] = dict(x.split("=") for x in t[2:])
            continue
        if t[0] == "ENDROWS":
            continue
        if t[0] == "COLUMN":
            d.setdefault("column", {})[t[1]] = line[len("COLUMN " + t[1] + " "):]
            continue
        if t[0] == "PHASE":
            cur_phase = d["phase"].setdefault(t[1], {"vpos": [], "hit": [], "report": []})
        elif t[0] == "ENDPHASE":
            cur_phase = None
        elif cur_phase is not None:
            if t[0] == "VPOS":
                cur_phase["vpos"].append((t[1], int(t[2]), int(t[3]), t[4], t[5]))
            elif t[0] == "HIT":
                cur_phase["hit"].append((t[1], int(t[2]), t[3], t[4] if len(t) > 4 else ""))
            else:
                cur_phase["report"].append(line)
        elif t[0] == "FISHER":
            d["fisher"][int(t[1])] = float(t[2])
        elif t[0] == "REF":
            d["ref"] = t[1]
        elif t[0] == "REFX":
            d["refx"] = t[1]
        elif t[0] == "MINOR":
            d["minors"].append((t[1], int(t[2]), t[3], t[4]))
        elif t[0] == "READS":
            cur_reads = d["reads"].setdefault(t[1], [])
            reads_name = t[1]
        elif t[0] == "READ":
            qs = [[int(x) for x in f.split(",")] for f in t[5:]]
            cur_reads.append((int(t[1]), int(t[2]), t[3], t[4], qs))
        elif t[0] == "CASE":
            cur_case = d["cases"].setdefault(t[1], {"reads": reads_name, "nuc": [], "codon": {}, "var": [], "msa": []})
        elif t[0] == "BEGIN":
            cur_case["begin"], cur_case["end"] = int(t[1]), int(t[3])
        elif t[0] == "NUC":
            cur_case["nuc"].append([int(x) for x in t[1:8]] + [int(t[9]), t[11]])
        elif t[0] == "CODON":
            cur_case["codon"][int(t[1])] = {kv.split(":")[0]: int(kv.split(":")[1]) for kv in t[2:]}
        elif t[0] == "NTESTS":
            cur_case["ntests"] = int(t[1])
        elif t[0] == "VAR":
            cur_case["var"].append((t[1], int(t[2]), t[3], t[4], t[5], float(t[6]), float(t[7]), " ".join(t[8:-2]), int(t[-1])))
        elif t[0] == "MSA":
            cur_case["msa"].append((t[1], int(t[2])) + tuple(int(x) for x in t[3:11]) + (t[11],))
        elif t[0] == "PERF":
            cur_case["perf"] = [float(x) for x in t[1:]]
        elif t[0] == "ERR":
            d["err"].append(" ".join(t[1:]))
    return d


def test_one_rank_communicator_changes_nothing():
    """The front as the one rank of a sharded sample (Msq::SetCommunicator: RCCL communicator, share rule, bounds and
    count all-reduces, all-gather of the phasing keys -- SURVEY 8e behind the class interface): byte-identical dump."""
    env = dict(os.environ)
    for k in ("DELQV", "SUBQV", "INSQV", "QUALQV"):
        env.pop(k, None)
    import tempfile
    with tempfile.TemporaryDirectory() as cwd:  # validation.json lands in the working directory
        plain = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env, cwd=cwd)
        env["MSQ_TEST_COMM"] = "1"
        comm = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env, cwd=cwd)
    assert plain.returncode == 0 and comm.returncode == 0, comm.stderr[-2000:]
    assert "communicator: rank 0 of 1" in comm.stderr
    keep = lambda text: [l for l in text.splitlines() if not l.startswith(("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl"))]
    assert keep(comm.stdout) == keep(plain.stdout) and len(plain.stdout) > 100000


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_front_with_several_ranks_prints_the_plain_dump_on_every_rank(world):
    """The C++ front with world > 1 on the one-GPU box: `world` processes share the sample (Msq::SetCommunicator), each
    piles up its share, bounds and counts are all-reduced, phasing keys all-gathered (padded shares laid out in read
    order), statuses agreed before any rank throws -- through the library's own msq_comm_* entry points, over the
    shared-memory rehearsal transport (MSQ_COMM_TRANSPORT=shm: RCCL refuses ranks that share a GPU).  Every rank must
    print, line for line, what the plain single-process run prints: matrices, calls, haplotypes, reports, exceptions."""
    env = dict(os.environ)
    for k in ("DELQV", "SUBQV", "INSQV", "QUALQV"):
        env.pop(k, None)
    import tempfile
    with tempfile.TemporaryDirectory() as cwd:
        plain = subprocess.run([BIN], capture_output=True, text=True, timeout=300, env=env, cwd=cwd)
        assert plain.returncode == 0
        env["MSQ_COMM_TRANSPORT"] = "shm"
        env["MSQ_TEST_COMM_ID"] = os.urandom(16).hex()
        procs = []
        for r in range(world):
            e = dict(env, MSQ_TEST_COMM="%d/%d" % (r, world))
            d = os.path.join(cwd, "rank%d" % r)
            os.mkdir(d)
            # into files: a rank blocked on a full pipe never reaches the next collective, and the others wait for it there
            procs.append(subprocess.Popen([BIN], stdout=open(os.path.join(d, "out"), "w"), stderr=open(os.path.join(d, "err"), "w"), env=e, cwd=d))
        for p in procs:
            p.wait(timeout=280)
        outs = [(open(os.path.join(cwd, "rank%d" % r, "out")).read(), open(os.path.join(cwd, "rank%d" % r, "err")).read()) for r in range(world)]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, err[-2000:])
        assert "communicator: rank %d of %d" % (r, world) in err
        assert out.splitlines() == plain.stdout.splitlines(), r


def to_reads(recs, rich):
    rows = []
    for start, end, cig, nuc, qs in recs:
        if rich:
            rows.append((start, cig, nuc, qs[0], qs[1], qs[2], qs[3]))
        else:
            rows.append((start, cig, nuc, qs[0]))
    r = cases.make_reads(rows, rich=rich)
    r["ref_end"] = np.array([x[1] for x in recs], dtype=np.int32)
    return r


def test_fisher_equivalent_with_R(dump):
    p_r = json.load(open(os.path.join(HERE, "golden", "fisher_R_greater.json")))["p"]
    for i in range(11, 100):  # the reference's own loop and tolerance (tests/unit/FisherTest.cpp:71-73)
        assert abs(p_r[i] - dump["fisher"][i]) <= p_r[i] / 1e5
    for i in range(0, 101):
        want = orc.fisher(i, 1000, 10, 1000)
        assert abs(dump["fisher"][i] - want) <= 1e-9 * want


GENES = {"noconfig": [], "rich_default_subqv42": [],
         "config": [("geneA", 41, 161), ("geneB", 102, 230)], "drmonly": [("geneA", 41, 161), ("geneB", 102, 230)],
         "debug_maxperc90": [("geneA", 41, 161), ("geneB", 102, 230)],
         "expectedminors": [("geneA", 41, 161), ("geneB", 102, 230)]}


def drm_string(gene, aa_pos):  # the DRM table of tests/cpp/front_test.cpp through FindDRMs (src/AminoAcidCaller.cpp:105-124)
    if gene != "geneA":
        return ""
    return {11: "drugA_+_drugB", 21: "drugB"}.get(aa_pos, "")


@pytest.mark.parametrize("case", sorted(GENES))
def test_caller_matches_oracle(dump, case):
    c = dump["cases"][case]
    rich = c["reads"] == "rich"
    reads = to_reads(dump["reads"][c["reads"]], rich)
    m = orc.Msa(reads, (-1, -1, 42, -1))  # QvThresholds defaults (src/QvThresholds.cpp:52-55)
    assert (c["begin"], c["end"]) == (m.begin1, m.end1)
    nuc = m.nuc_counts()
    got = np.array([x[1:7] for x in c["nuc"]])
    assert (got == nuc).all()
    for k, x in enumerate(c["nuc"]):
        assert x[0] == m.begin1 + k and x[7] == nuc[k].sum()  # RefPos is 1-based (src/MSA.cpp:60-66)
        best = int(np.argmax(nuc[k]))
        assert x[8] == ("_" if best == 5 else "ACGT-"[best])
    cod = m.codon_counts()
    for i in range(m.W):
        want = {CODONS[k]: int(v) for k, v in enumerate(cod[i]) if v}
        assert c["codon"].get(i, {}) == want, i

    genes = GENES[case]
    ref = dump["ref"] if genes else ""
    minors = []
    if case == "expectedminors":  # the configured reference differs from the sample in two codons; geneA lists minors
        ref, minors = dump["refx"], dump["minors"]
        assert len(minors) == 4 and ref != dump["ref"]
    perf = {"TP": 0, "FP": 0, "FN": 0, "TN": 0}
    kinds = set()
    debug = case == "debug_maxperc90"
    drm_only = case == "drmonly"
    max_perc = 90.0 if debug else 100.0
    pos, cells, nt = m.call([g[1] for g in genes], [g[2] for g in genes], ref, maximal_perc=max_perc)
    assert c["ntests"] == nt
    names = [g[0] for g in genes] or ["Unnamed_ORF"]
    want_vars = []
    for oc in cells:
        p = pos[oc.position]
        gene, aa_pos, coverage = names[p[0]], int(p[2]), int(p[4])
        aa = orc.lib().orc_amino_acid(oc.codon).decode()
        # MeasurePerformance (src/AminoAcidCaller.cpp:341-378): predictor = an expected minor of this gene with this
        # position, amino acid and codon; counted at variable sites (frequency < 0.8) and for predictors anywhere
        predictor = any(mg == gene and mp == aa_pos and ma[0] == aa and mc == CODONS[oc.codon] for mg, mp, ma, mc in minors)
        variable = oc.frequency < 0.8
        sig = oc.p < ALPHA
        if variable or predictor:
            perf[("TP" if sig else "FN") if predictor else ("FP" if sig else "TN")] += 1
        store = False
        if debug:
            store = True
        elif sig:
            if drm_only:
                store = bool(drm_string(gene, aa_pos))
            elif predictor:  # :529-537
                store = True
            elif minors:
                store = variable
            else:
                store = True
            kinds.add((predictor, variable, store))
        if store and (debug or oc.frequency * 100 >= 0.1):
            want_vars.append((gene, aa_pos, CODONS[p[5]], aa, CODONS[oc.codon], oc.frequency, oc.p, drm_string(gene, aa_pos) or "-", coverage))
    key = lambda v: (names.index(v[0]), v[1], v[3], CODONS.index(v[4]))
    got_vars = sorted(c["var"], key=key)
    want_vars = sorted(want_vars, key=key)
    assert len(got_vars) == len(want_vars), (len(got_vars), len(want_vars))
    assert len(want_vars) > 0  # every case is built to call something
    for g, w in zip(got_vars, want_vars):
        assert g[:5] == w[:5] and g[7] == w[7] and g[8] == w[8], (g, w)
        assert g[5] == w[5]
        assert abs(g[6] - w[6]) <= 1e-9 * abs(w[6]) + 1e-300
    # the 9-column context of every stored position comes from the nucleotide matrix (src/AminoAcidCaller.cpp:544-566)
    stored = {(v[0], v[1]) for v in got_vars}
    assert {(x[0], x[1]) for x in c["msa"]} == stored
    gene_begin = {g[0]: g[1] for g in genes} or {"Unnamed_ORF": m.begin1}
    for gname, aa_pos, rel, absp, A, C_, G, T, gap, N, wt in c["msa"]:
        i = gene_begin[gname] + 3 * (aa_pos - 1)
        assert absp == i - 1 + rel and m.begin1 <= i + rel < m.end1
        col = absp - (m.begin1 - 1)
        assert [A, C_, G, T, gap, N] == nuc[col].tolist()
        if ref:
            assert wt == ref[absp]
        else:
            best = int(np.argmax(nuc[col]))
            assert wt == ("_" if best == 5 else "ACGT-"[best])
    assert c["perf"] == [perf["TP"], perf["FP"], perf["FN"], perf["TN"]]
    if minors:
        # every branch of the policy is taken: a predictor at a variable site and one at a non-variable site (stored),
        # a variable non-predictor (stored), a significant non-variable non-predictor (the sample's majority codon
        # against the altered reference: NOT stored), and a predictor that stays insignificant
        assert {(True, True, True), (True, False, True), (False, True, True), (False, False, False)} <= kinds
        assert perf["TP"] >= 2 and perf["FN"] >= 1 and perf["FP"] >= 2 and perf["TN"] > 100
        stored = {(v[0], v[1], v[4]) for v in got_vars}
        assert ("geneA", 11, minors[0][3]) in stored and ("geneA", 7, minors[1][3]) in stored
        assert not any(v[0] == "geneA" and v[1] == 5 for v in got_vars)  # the unlisted majority codon at amino acid 5
    else:
        assert c["perf"][0] == 0  # no expected minors: nothing can be a true positive


@pytest.mark.parametrize("case", ["noconfig", "config", "rich_default_subqv42"])
def test_phase_variants_match_oracle(dump, case):
    """AminoAcidCaller::PhaseVariants through the front (GPU gather + host clustering) against the oracle's
    literal restatement on the same reads: haplotypes, their reads, flags, names, frequencies and order, the
    read-count summary and every haplotypeHit vector (parity unpinned, see oracle/oracle.cpp)."""
    c = dump["cases"][case]
    ph = dump["phase"][case]
    rich = c["reads"] == "rich"
    m = orc.Msa(to_reads(dump["reads"][c["reads"]], rich), (-1, -1, 42, -1))
    genes = GENES[case]
    names = [g[0] for g in genes] or ["Unnamed_ORF"]
    begin = {g[0]: g[1] for g in genes} or {"Unnamed_ORF": m.begin1}
    ref = dump["ref"] if genes else ""
    o_pos, _, _ = m.call([g[1] for g in genes], [g[2] for g in genes], ref)
    by_key = {(names[int(p[0])], int(p[2])): p for p in o_pos}
    show = lambda s: "" if s == "." else s.replace("_", " ")
    pos, hits = [], []
    assert len(ph["vpos"]) > 0
    for gene, offset, aa_pos, ref_codon, alt_codon in ph["vpos"]:
        assert offset == begin[gene]  # VariantGene(gene.name, gene.begin), src/AminoAcidCaller.cpp:419
        p = by_key[(gene, aa_pos)]
        if p[5] >= 0:
            assert show(ref_codon) == CODONS[p[5]]
        assert show(alt_codon) == (CODONS[p[6]] if p[6] >= 0 else "")
        pos.append(offset + aa_pos * 3 - m.begin1 - 3)  # :134 and :154
        assert pos[-1] == int(p[3])                     # ... which is the codon's window column
        stored = [v[4] for v in c["var"] if v[0] == gene and v[1] == aa_pos]
        assert stored
        hits.append([show(ref_codon), show(alt_codon)] + stored)
    want = m.phase(pos, hits)
    got = orc.parse_phase_report("\n".join(ph["report"]))
    assert got["cnt"] == want["cnt"]
    assert sum(want["cnt"][:1]) + sum(h["size"] for h in want["fil"]) == len(dump["reads"][c["reads"]])
    for kind in ("gen", "fil"):
        assert len(got[kind]) == len(want[kind])
        for g, w in zip(got[kind], want[kind]):
            assert g == w, (kind, g["codons"], w["codons"])
    # the major clone and the planted 8 % clone phase into separate haplotypes (under the SubQV mask of the rich
    # case too few clean minor reads are left)
    assert len(want["gen"]) >= (1 if rich else 2)
    # haplotypeHit: one bit per reported haplotype, set when it carries the codon (src/AminoAcidCaller.cpp:253-263)
    kpos = {(v[0], v[2]): k for k, v in enumerate(ph["vpos"])}
    assert len(ph["hit"]) == len(c["var"])
    for gene, aa_pos, codon, bits in ph["hit"]:
        if (gene, aa_pos) not in kpos:
            assert bits == ""
            continue
        k = kpos[(gene, aa_pos)]
        assert bits == "".join("1" if h["codons"][k] == codon else "0" for h in want["gen"])


def test_minor_clone_is_called(dump):
    # the planted 8 % clone (columns 31, 62/63, 121 of the window) must surface in the no-config run
    got = {(v[1], v[4]) for v in dump["cases"]["noconfig"]["var"] if v[5] > 0.03}
    assert len(got) >= 3


def test_exceptions_keep_reference_types(dump):
    e = dump["err"]
    assert e[0] == "runtime_error Unexpected cigar 78"   # std::to_string('N'), src/MSA.cpp:182
    assert e[1] == "runtime_error Unexpected base"       # src/MSA.cpp:85
    assert e[2].startswith("out_of_range")               # referenceSequence.substr, src/AminoAcidCaller.cpp:446
    assert e[3] == "runtime_error Empty input. Could not find records."  # src/tools/Fuse.cpp:72


@pytest.mark.parametrize("case", ["noconfig", "config", "rich_default_subqv42"])
def test_json_report(dump, case):
    """AminoAcidCaller::JSON().dump(2) (src/AminoAcidCaller.cpp:590-619, src/VariantGene.cpp:42-81,
    src/Haplotype.cpp:59-70).  Content: exactly the stored variants, 9-column contexts, haplotypes and
    read counts the tests above compared with the oracle.  Form: nlohmann's dump(2) -- sorted keys, two spaces,
    one member per line -- which is also Python's json.dumps(indent=2, sort_keys=True); numbers are compared as
    values ("%.15g" here, see msq_json.hpp; parity unpinned, the reference holds no JSON fixture)."""
    import re
    text = dump["json"][case]
    rep = json.loads(text)
    c, ph = dump["cases"][case], dump["phase"][case]
    assert sorted(rep) == ["genes", "haplotype_read_counts", "haplotypes"]
    num = re.compile(r"-?\d+\.\d+(?:e[+-]?\d+)?|-?\d+e[+-]?\d+")
    relaid = json.dumps(rep, indent=2, sort_keys=True, ensure_ascii=False) + "\n"
    assert num.sub("#", text) == num.sub("#", relaid)
    for a, b in zip(num.findall(text), num.findall(relaid)):
        assert abs(float(a) - float(b)) <= 1e-14 * abs(float(b))
    close = lambda x, y: abs(x - y) <= 1e-14 * abs(y) + 1e-300

    hits = {(g, p, cod): bits for g, p, cod, bits in ph["hit"]}
    seen_var, seen_msa = [], []
    for gene in rep["genes"]:
        gname = gene["name"].replace(" ", "_")
        assert sorted(gene) == ["name", "variant_positions"]
        for vp in gene["variant_positions"]:
            assert sorted(vp) == ["coverage", "msa", "ref_amino_acid", "ref_codon", "ref_position", "variant_amino_acids"]
            for aa in vp["variant_amino_acids"]:
                for vc in aa["variant_codons"]:
                    seen_var.append((gname, vp["ref_position"], vp["ref_codon"], aa["amino_acid"], vc["codon"], vc["frequency"],
                                     vc["pValue"], vc["known_drm"].replace(" ", "_") or "-", vp["coverage"]))
                    bits = hits[(gname, vp["ref_position"], vc["codon"])]
                    assert vc["haplotype_hit"] == [b == "1" for b in bits]
            assert vp["ref_amino_acid"] == orc.lib().orc_amino_acid(CODONS.index(vp["ref_codon"])).decode()
            for m in vp["msa"]:
                assert sorted(m) == ["-", "A", "C", "G", "N", "T", "abs_pos", "rel_pos", "wt"]
                seen_msa.append((gname, vp["ref_position"], m["rel_pos"], m["abs_pos"], m["A"], m["C"], m["G"], m["T"], m["-"], m["N"],
                                 "_" if m["wt"] == " " else m["wt"]))
    assert len(seen_var) == len(c["var"]) > 0
    for g, w in zip(seen_var, c["var"]):  # same order: genes, positions, amino acids, codons as stored
        assert g[:5] == w[:5] and g[7:] == w[7:], (g, w)
        assert close(g[5], w[5]) and close(g[6], w[6])
    assert seen_msa == c["msa"]

    rp = orc.parse_phase_report("\n".join(ph["report"]))
    assert len(rep["haplotypes"]) == len(rp["gen"])
    for h, w in zip(rep["haplotypes"], rp["gen"]):
        assert sorted(h) == ["codons", "frequency", "name", "read_names", "reads_hard", "reads_soft"]
        assert h["name"] == w["name"] and h["reads_hard"] == w["size"] and h["reads_soft"] == float(w["size"])
        assert close(h["frequency"], w["frequency"]) and h["codons"] == w["codons"]
        assert h["read_names"] == ["read/" + r for r in w["reads"]]
    cnt = rep["haplotype_read_counts"]
    assert [cnt[k] for k in ("healthy_reported", "healthy_low_coverage", "all_damaged", "marginal_with_gaps",
                             "marginal_with_heteroduplexes", "marginal_partial_reads")] == rp["cnt"]


@pytest.mark.parametrize("name", ["fuse400", "fuse30"])
def test_fuse_consensus_and_insertion_maps(dump, name):
    """PacBio::Fuse::Fuse (src/tools/Fuse.cpp:60-140) and MSAColumn::Insertions() through the front -- pileup and
    insertion histogram on the GPU -- against the oracle's restatement on the same reads (parity
    unpinned: the reference holds no fuse fixture)."""
    m = orc.Msa(to_reads(dump["reads"][name], False))
    for min_cov in (50, 10):
        want = m.fuse(min_cov)
        assert dump["fuse"][(name, min_cov)] == want
    got = dump["ins"][name]
    n_cols = 0
    for c in range(m.W):
        want = m.col_insertions(c)
        assert got.get(c, {}) == want, c
        n_cols += bool(want)
    assert n_cols > 5
    cons = dump["fuse"][(name, 50)]
    if name == "fuse400":
        # the planted structure: GGA (80 %) goes in, TTTCCC (60 %, 10 columns on) is suppressed by it, AC is out of
        # frame, CAT stays below half, neither AAA nor AAC reaches half alone; column 50 is a majority deletion
        nuc = m.nuc_counts()
        base = lambda c: "ACGT-"[int(np.argmax(nuc[c]))]
        plain = "".join(base(c) for c in range(m.W) if base(c) != "-")
        assert len(cons) == len(plain) + 3 and base(50) == "-"
        at = sum(1 for c in range(100) if base(c) != "-")
        assert cons == plain[:at] + "GGA" + plain[at:]
    else:
        assert len(m.fuse(50)) > 0 and len(dump["reads"][name]) < 50  # permissive mode: minCoverage 1


@pytest.mark.parametrize("name", ["plain", "rich", "fuse400", "fuse30"])
def test_rows_match_oracle(dump, name):
    """MSAByRow::Rows() / NameToRow() / MSARow (include/pacbio/data/MSA.h:72-75,96-110; AddRead, src/MSA.cpp:144-186)
    through the front -- the characters from msq_rows_decode over the packed reads in HBM, the insertion maps from the
    insertion scan (over a set packed again in the plain layout for "plain" / "rich", over the resident one for the
    fuse sets) -- against the oracle's rows and per-row insertion maps; parity unpinned like the pileup."""
    got = dump["rows"][name]
    rich = name == "rich"
    m = orc.Msa(to_reads(dump["reads"][name], rich), (-1, -1, 42, -1))
    want = m.rows()
    assert (got["n"], got["W"]) == (want.shape[0], m.W) and len(got["rows"]) == want.shape[0]
    n_ins = n_masked = 0
    for r, (bases, ins) in enumerate(got["rows"]):
        assert bases.replace("_", " ") == want[r].tobytes().decode(), r
        # a run flushed behind the last column is an out-of-bounds write in the reference and dropped here
        assert ins == {c: s for c, s in m.row_insertions(r).items() if c < m.W}, r
        n_ins += len(ins)
        n_masked += bases.count("N")
    assert n_ins > 20
    if rich:
        assert n_masked > 1000  # SUBQV = 42 masks a tenth of the match bases: 'N' in the rows (src/MSA.cpp:163-166)
    if name == "fuse400":  # the strings the packed form cannot hold come back whole: 69 bases, a non-ACGT character
        assert any(len(s) == 69 for _, ins in got["rows"] for s in ins.values())
        assert any("N" in s for _, ins in got["rows"] for s in ins.values())
    chk = got["check"]
    # CodingCodonAt over the rows == the MSA's own CodonsAt at every column (incl. -1, 0 and W); NameToRow finds every
    # row and throws std::out_of_range otherwise; rows share the reads only in the shared_ptr constructor (MSA.cpp:101)
    assert chk["codons_equal"] == "1" and chk["name_to_row"] == "1" and chk["reads"] == "1" and chk["unknown_throws"] == "1"
    assert chk["codonat0"] == "2" and chk["codonat_edge"] == "2"  # CodonAt's i > 0 (MSA.cpp:193) and the right edge


def test_validation_record_is_written(dump):
    """With expected minors configured the caller writes validation.json into the working directory and the same
    numbers to stderr (src/AminoAcidCaller.cpp:571-588, internal/AminoAcidCaller.inl:42-47): ostream formatting (6
    significant digits) of the rates worked out from the TP / FP / FN / TN counts that test_caller_matches_oracle pins."""
    c = dump["cases"]["expectedminors"]
    tp, fp, fn, tn = c["perf"]
    n_tests, n_minors = c["ntests"], len(dump["minors"])
    g = lambda x: "%g" % x
    want = ('{"true_positive_rate":%s,"false_positive_rate":%s,"num_tests":%s,"num_false_positives":%s,"accuracy":%s}' %
            (g(tp / n_minors), g(fp / (n_tests - n_minors)), g(n_tests), g(fp), g((tp + tn) / (tp + fp + fn + tn))))
    assert dump["validation_json"] == want
    line = "%s %s %s %s %s" % (g(tp / n_minors), g(fp / (n_tests - n_minors)), g(n_tests), g((tp + tn) / (tp + fp + fn + tn)), g(fp))
    assert line in dump["stderr"].splitlines()


def test_column_result_slots(dump):
    """MSAColumn::PValue / SignificantInsertions / AddFisherResult / InDels / operator<< (MSA.h:178-191, MSA.inl:82-112,
    src/MSA.cpp:223-237): 1 until a result is attached, then what was attached."""
    assert dump["column"]["before"] == "1 1 0 | 1\t0\t0\t0\t1\t|"
    assert dump["column"]["after"] == "0.5 0.001 1 T | 2\t(-,0,0.001)\t(T,1,0.001)\t"


def test_json_serializer_fixed_points(dump):
    """nlohmann's published text forms that the report may not hit: null, empty containers, escapes, integral,
    small and large doubles ("%.15g" + ".0"), 64-bit unsigned, nesting; compact and dump(2)."""
    want = ('{"a":null,"b":[],"d":[1.0,0.1,-2.5e-07,1e+21,0.0,123456789.125],"i":-7,"n":{"deep":[true,false]},'
            '"s":"q\\"\\\\\\n\\t\\u0001z","u":18446744073709551615}')
    assert dump["jsonform"] == want
    pretty = dump["json"]["form"]
    assert json.loads(pretty) == json.loads(want)
    assert pretty.startswith('{\n  "a": null,\n  "b": [],\n  "d": [\n    1.0,\n    0.1,\n') and pretty.endswith('\n}\n')
    assert '  "n": {\n    "deep": [\n      true,\n      false\n    ]\n  },\n' in pretty


@pytest.mark.parametrize("case", ["noconfig", "config", "rich_default_subqv42"])
def test_msa_table_and_error_mode(dump, case):
    """--msa output (src/tools/JulietWorkflow.cpp:178-191) and --mode-error (:193-213) from the count matrix."""
    c = dump["cases"][case]
    lines = dump["msatable"][case].splitlines()
    assert lines[0] == "pos A C G T - N"
    assert len(lines) == 1 + len(c["nuc"])
    for line, x in zip(lines[1:], c["nuc"]):
        assert [int(v) for v in line.split(" ")] == x[:7]  # RefPos (1-based) + the six counts, checked against the oracle above
    sub = dl = n = 0.0
    for x in c["nuc"]:
        cov = sum(x[1:7])
        if cov > 100:
            gap = x[5] / cov
            best = max(x[1:7])
            assert x[1:7].index(best) != 5  # an 'N' majority has no Frequency(' ') in the reference either
            dl += gap
            sub += 1.0 - gap - best / cov
            n += 1
    e = dump["errmode"][case].splitlines()
    assert e[0] == "input.bam" and e[1].startswith("sub: ") and e[2].startswith("del: ")
    assert abs(float(e[1][5:]) - sub / n) <= 1e-5 * (sub / n)  # ostream default precision: 6 digits
    assert abs(float(e[2][5:]) - dl / n) <= 1e-5 * (dl / n)
