// front_test.cpp -- drives the C++ front (minorseq_amd/cpp/msq_front.hpp) the way juliet's workflow drives the
// reference classes (src/tools/JulietWorkflow.cpp:136-165) and dumps everything it computes as text, for
// tests/test_gpu_cpp_front.py to compare with the oracle.  Also mirrors the reference's one unit test,
// FisherTest.EquivalentWithR (tests/unit/FisherTest.cpp:49-74): prints fisher_exact_tiss(i,1000,10,1000).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "dump.hpp"

static uint64_t g_state = 0x1234567;
static uint32_t rnd()
{
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(g_state >> 33);
}
static double urand() { return rnd() / 2147483648.0; }

static std::vector<std::shared_ptr<Data::ArrayRead>> MakeReads(const std::string& ref, int start0, int n, bool rich)
{
    static const char nt[] = "ACGT";
    std::vector<std::shared_ptr<Data::ArrayRead>> reads;
    const int W = (int)ref.size();
    for (int r = 0; r < n; ++r) {
        int a = 0, b = W;
        if (r > 0 && urand() < 0.3) {
            a = (int)(rnd() % (W / 2));
            b = a + 1 + (int)(rnd() % (W - a));
        }
        const bool minor = urand() < 0.08;  // a minor clone: three fixed codon changes
        std::vector<Data::ArrayBase> bases;
        auto push = [&](char cg, char c) {
            const uint8_t q = (uint8_t)(rnd() % 94);
            if (rich)
                bases.emplace_back(cg, c, q, (uint8_t)(30 + rnd() % 64), (uint8_t)(rnd() % 94), (uint8_t)(rnd() % 94));
            else
                bases.emplace_back(cg, c, q);
        };
        for (int c = a; c < b; ++c) {
            if (c > a && urand() < 0.01) push('I', nt[rnd() % 4]);
            char truth = ref[c];
            if (minor && (c == 31 || c == 62 || c == 63 || c == 121)) truth = truth == 'A' ? 'G' : 'A';
            const double u = urand();
            if (u < 0.004)
                push('D', '-');
            else if (u < 0.006)
                push('X', truth == 'C' ? 'T' : 'C');
            else
                push('=', truth);
        }
        std::printf("READ %d %d ", start0 + a, start0 + b);
        for (const auto& x : bases) std::putchar(x.Cigar);
        std::putchar(' ');
        for (const auto& x : bases) std::putchar(x.Nucleotide);
        for (int k = 0; k < (rich ? 4 : 1); ++k) {
            std::putchar(' ');
            for (size_t i = 0; i < bases.size(); ++i) {
                const auto& x = bases[i];
                std::printf("%s%d", i ? "," : "", k == 0 ? x.QualQV : k == 1 ? x.SubQV : k == 2 ? x.DelQV : x.InsQV);
            }
        }
        std::putchar('\n');
        reads.push_back(std::make_shared<Data::ArrayRead>(r, "read/" + std::to_string(r), start0 + a, start0 + b, bases));
    }
    return reads;
}

static std::string ShowCodon(const std::string& c)  // as oracle/oracle.cpp's report: "" -> ".", ' ' -> '_'
{
    if (c.empty()) return ".";
    std::string o = c;
    for (auto& ch : o)
        if (ch == ' ') ch = '_';
    return o;
}

static void DumpHaplotype(const char* tag, const Juliet::Haplotype& h)
{
    std::printf("%s %s %zu %d %.17g ", tag, h.Name().empty() ? "." : h.Name().c_str(), h.ReadNames().size(), h.Flags(), h.Frequency());
    for (size_t i = 0; i < h.NumCodons(); ++i) std::printf("%s%s", i ? "," : "", ShowCodon(h.Codon((int)i)).c_str());
    std::printf("\nREADS");
    for (const auto& n : h.ReadNames()) std::printf(" %s", n.substr(n.find('/') + 1).c_str());  // "read/<index>"
    std::printf("\n");
}

// AminoAcidCaller::PhaseVariants (src/AminoAcidCaller.cpp:126-323) through the front: the variant positions it
// phases over, the haplotypes in the oracle's report format, and every VariantCodon's haplotypeHit vector.
static void DumpPhase(const char* name, Juliet::AminoAcidCaller& aac)
{
    aac.PhaseVariants();
    std::printf("PHASE %s\n", name);
    for (const auto& vg : aac.VariantGenes())
        for (const auto& pv : vg.relPositionToVariant)
            if (pv.second->IsVariant())
                std::printf("VPOS %s %d %d %s %s\n", Tok(vg.geneName).c_str(), vg.geneOffset, pv.first, ShowCodon(pv.second->refCodon).c_str(),
                            ShowCodon(pv.second->altRefCodon).c_str());
    for (const auto& h : aac.ReconstructedHaplotypes()) DumpHaplotype("GEN", h);
    for (const auto& h : aac.FilteredHaplotypes()) DumpHaplotype("FIL", h);
    const auto c = aac.HaplotypeReadCounts();
    std::printf("CNT %d %d %d %d %d %d\n", c[0], c[1], c[2], c[3], c[4], c[5]);
    for (const auto& vg : aac.VariantGenes())
        for (const auto& pv : vg.relPositionToVariant)
            for (const auto& aa : pv.second->aminoAcidToCodons)
                for (const auto& vc : aa.second) {
                    std::printf("HIT %s %d %s ", Tok(vg.geneName).c_str(), pv.first, vc.codon.c_str());
                    for (const bool b : vc.haplotypeHit) std::putchar(b ? '1' : '0');
                    std::putchar('\n');
                }
    std::printf("ENDPHASE\n");
}

// MSAByRow::Rows() / NameToRow() / MSARow (include/pacbio/data/MSA.h:72-75,96-110): every row's characters and
// insertion map, and -- checked right here -- that the rows reproduce the MSA's own codon counts through
// MSARow::CodingCodonAt exactly as MSAByRow::CodonsAt sums them up in the reference (src/MSA.cpp:126-136).
static void DumpRows(const char* name, const Data::MSAByRow& msa, bool sharedReads)
{
    const auto& rows = msa.Rows();
    std::printf("ROWS %s %zu %d\n", name, rows.size(), msa.Width());
    for (size_t r = 0; r < rows.size(); ++r) {
        std::string b(rows[r]->Bases.begin(), rows[r]->Bases.end());
        for (auto& ch : b)
            if (ch == ' ') ch = '_';
        std::printf("ROW %zu %s", r, b.c_str());
        for (const auto& kv : rows[r]->Insertions) std::printf(" %d:%s", kv.first, kv.second.c_str());
        std::putchar('\n');
    }
    bool codonsEqual = true, byName = true, readsShared = true;
    std::string codon;
    for (int i = -1; i <= msa.Width(); ++i) {
        std::map<std::string, int> fromRows;
        for (const auto& row : rows)
            if (row->CodingCodonAt(i, &codon)) ++fromRows[codon];
        codonsEqual = codonsEqual && fromRows == msa.CodonsAt(i);
    }
    for (size_t r = 0; r < rows.size(); ++r) {
        byName = byName && msa.NameToRow(msa.ReadNames()[r]) == rows[r];
        readsShared = readsShared && (sharedReads ? (rows[r]->Read && rows[r]->Read->Name() == msa.ReadNames()[r]) : !rows[r]->Read);
    }
    bool unknownThrows = false;
    try {
        msa.NameToRow("no such read");
    } catch (const std::out_of_range&) {
        unknownThrows = true;
    }
    // CodonAt keeps the reference's `i > 0`: at 0 two characters, inside three, at the right edge what is left
    const std::string first = rows.empty() ? "" : rows[0]->CodonAt(0), last = rows.empty() ? "" : rows[0]->CodonAt(msa.Width() - 2);
    std::printf("ROWCHECK %s codons_equal=%d name_to_row=%d reads=%d unknown_throws=%d codonat0=%zu codonat_edge=%zu\n", name, codonsEqual,
                byName, readsShared, unknownThrows, first.size(), last.size());
    std::printf("ENDROWS\n");
}

int main()
{
    // MSQ_TEST_COMM=1: the same run as the one rank of a "sharded" sample -- the library's RCCL communicator, the
    // share rule, the bounds / count all-reduces and the all-gather of the phasing keys are all on the path, and
    // every line of the dump must be what the plain run prints
    // MSQ_TEST_COMM=<rank>/<world> with MSQ_TEST_COMM_ID=<hex of the communicator id> (and MSQ_COMM_TRANSPORT=shm on a box
    // with one GPU): one of several processes that share the sample -- EVERY rank must print the plain run's dump
    if (const char* spec = std::getenv("MSQ_TEST_COMM")) {
        uint8_t id[MSQ_COMM_ID_BYTES] = {0};
        int rank = 0, world = 1;
        if (std::sscanf(spec, "%d/%d", &rank, &world) == 2) {
            const char* hex = std::getenv("MSQ_TEST_COMM_ID");
            if (!hex) throw std::runtime_error("MSQ_TEST_COMM=<rank>/<world> needs MSQ_TEST_COMM_ID");
            for (size_t k = 0; k < MSQ_COMM_ID_BYTES && hex[2 * k] && hex[2 * k + 1]; ++k) {
                unsigned v = 0;
                std::sscanf(hex + 2 * k, "%2x", &v);
                id[k] = (uint8_t)v;
            }
        } else {
            rank = 0;
            world = 1;
            Msq::Check(msq_comm_unique_id(id));
        }
        Msq::SetCommunicator(rank, world, id);
        std::fprintf(stderr, "communicator: rank %d of %d\n", Msq::CommRank(), Msq::CommWorld());
    }
    // FisherTest.EquivalentWithR
    for (int i = 0; i <= 100; ++i) std::printf("FISHER %d %.17g\n", i, Statistics::Fisher::fisher_exact_tiss(i, 1000, 10, 1000));

    {  // the serializer's fixed points (msq_json.hpp): empty containers, escapes, integral / small / negative doubles
        JSON::Json j;
        j["b"] = std::vector<JSON::Json>{};
        j["a"] = JSON::Json();
        j["s"] = std::string("q\"\\\n\t\x01z");
        j["d"] = std::vector<double>{1.0, 0.1, -2.5e-7, 1e21, 0.0, 123456789.125};
        j["n"]["deep"] = std::vector<bool>{true, false};
        j["u"] = size_t(18446744073709551615ull);
        j["i"] = -7;
        std::printf("JSONFORM %s\n", j.dump().c_str());
        std::printf("JSON form\n%s\nENDJSON\n", j.dump(2).c_str());
    }

    std::string ref;
    for (int i = 0; i < 180; ++i) ref += "ACGT"[rnd() % 4];
    const int start0 = 40;
    std::string full(start0, 'A');  // reference in TargetConfig coordinates: 40 filler bases, then the window
    full += ref;
    full += std::string(30, 'C');
    std::printf("REF %s\n", full.c_str());

    const Juliet::ErrorEstimates error("S/P2-C2");
    {
        std::printf("READS plain\n");
        auto reads = MakeReads(ref, start0, 1500, false);
        Juliet::JulietSettings s;  // no config: one ORF over the window
        Juliet::AminoAcidCaller aac(reads, error, s);
        Dump("noconfig", aac);
        DumpPhase("noconfig", aac);
        DumpReports("noconfig", aac);
        DumpRows("plain", aac.Rows(), true);

        Juliet::JulietSettings c;  // two genes in different frames + reference + DRMs + expected minors
        Juliet::DRM drm1{"drugA", {Juliet::DMutation('*', 11, '*')}};
        Juliet::DRM drm2{"drugB", {Juliet::DMutation('*', 11, '*'), Juliet::DMutation('*', 21, '*')}};
        c.TargetConfigUser.targetGenes.emplace_back(41, 161, "geneA", std::vector<Juliet::DRM>{drm1, drm2});
        c.TargetConfigUser.targetGenes.emplace_back(102, 230, "geneB", std::vector<Juliet::DRM>{});
        c.TargetConfigUser.referenceSequence = full;
        Juliet::AminoAcidCaller aac2(reads, error, c);
        Dump("config", aac2);
        DumpPhase("config", aac2);
        DumpReports("config", aac2);
        c.DRMOnly = true;
        Juliet::AminoAcidCaller aac3(reads, error, c);
        Dump("drmonly", aac3);
        c.DRMOnly = false;
        c.Debug = true;
        c.MaximalPerc = 90;
        Juliet::AminoAcidCaller aac4(reads, error, c);
        Dump("debug_maxperc90", aac4);

        // expected minors (an HIV-PB-style configuration): MeasurePerformance's predictor logic and the store
        // branches that depend on it (src/AminoAcidCaller.cpp:341-378, 519-539).  The configured reference differs
        // from the sample in two codons of geneA (amino acids 5 and 7), so the sample's own majority codon is tested
        // there at a frequency above 0.8: listed as an expected minor it is a predictor at a non-variable site
        // (stored, a true positive), unlisted it is neither and must NOT be stored.  The planted 8 % clone gives a
        // predictor at a variable site (amino acid 11) and variable non-predictors (amino acid 21, geneB); a
        // one-substitution neighbour seen a few times is a predictor that stays insignificant (false negative).
        Juliet::JulietSettings e;
        e.TargetConfigUser = c.TargetConfigUser;
        std::string fullx = full;
        auto other = [](char ch) { return ch == 'A' ? 'C' : 'A'; };
        fullx[start0 + 12] = other(fullx[start0 + 12]);  // amino acid 5 of geneA: window columns 12..14
        fullx[start0 + 19] = other(fullx[start0 + 19]);  // amino acid 7: columns 18..20
        e.TargetConfigUser.referenceSequence = fullx;
        std::string clone11 = ref.substr(30, 3), noise15 = ref.substr(42, 3);
        clone11[1] = clone11[1] == 'A' ? 'G' : 'A';       // the clone's change at column 31 (MakeReads)
        noise15[1] = noise15[1] == 'C' ? 'T' : 'C';       // the 'X' substitution MakeReads writes at column 43
        const std::string major7 = ref.substr(18, 3);
        auto minor = [](int pos, const std::string& codon) {
            return Juliet::ExpectedMinor{pos, std::string(1, Juliet::AminoAcidTable::At(codon)), codon};
        };
        e.TargetConfigUser.targetGenes[0].minors = {minor(11, clone11), minor(7, major7), minor(15, noise15), minor(30, "AAA")};
        std::printf("REFX %s\n", fullx.c_str());
        for (const auto& m : e.TargetConfigUser.targetGenes[0].minors)
            std::printf("MINOR geneA %d %s %s\n", m.position, m.aminoacid.c_str(), m.codon.c_str());
        Juliet::AminoAcidCaller aac5(reads, error, e);
        Dump("expectedminors", aac5);
        DumpReports("expectedminors", aac5);
    }
    {
        std::printf("READS rich\n");
        auto reads = MakeReads(ref, start0, 600, true);
        Juliet::JulietSettings s;
        Juliet::AminoAcidCaller aac(reads, error, s);  // SUBQV defaults to 42 (src/QvThresholds.cpp:53)
        Dump("rich_default_subqv42", aac);
        DumpPhase("rich_default_subqv42", aac);
        DumpReports("rich_default_subqv42", aac);
        DumpRows("rich", aac.Rows(), true);
    }
    // fuse (src/tools/Fuse.cpp:60-140): planted insertions -- in frame and strong, in frame but weaker and within 20
    // positions of a stronger one, out of frame, below half the coverage -- a majority deletion, noise; then too few reads
    for (const int n_reads : {400, 30}) {
        const int W = 300, start0 = 7;
        std::string fref;
        for (int i = 0; i < W; ++i) fref += "ACGT"[rnd() % 4];
        std::printf("READS fuse%d\n", n_reads);
        std::vector<Data::ArrayRead> reads;
        for (int r = 0; r < n_reads; ++r) {
            int a = 0, b = W;
            if (r > 0 && urand() < 0.2) {
                a = (int)(rnd() % (W / 2));
                b = a + 1 + (int)(rnd() % (W - a));
            }
            std::vector<Data::ArrayBase> bases;
            auto insert = [&](const char* s) {
                for (const char* p = s; *p; ++p) bases.emplace_back('I', *p, (uint8_t)(rnd() % 94));
            };
            for (int c = a; c < b; ++c) {
                if (c == 100 && urand() < 0.8) insert("GGA");
                if (c == 110 && urand() < 0.6) insert("TTTCCC");
                if (c == 200 && urand() < 0.9) insert("AC");
                if (c == 250 && urand() < 0.4) insert("CAT");
                if (c == 280 && urand() < 0.7) insert(urand() < 0.5 ? "AAA" : "AAC");
                // strings the packed 2-bit form cannot hold: 69 bases, and a non-ACGT char (src/MSA.cpp:151-156,172-174)
                if (c == 150 && r % 40 == 3) insert("ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTT");
                if (c == 160 && r % 40 == 5) insert("GNT");
                if (c > a && urand() < 0.01) insert(urand() < 0.5 ? "G" : "TCA");
                if (c == 50 && urand() < 0.7)
                    bases.emplace_back('D', '-', (uint8_t)0);
                else if (urand() < 0.01)
                    bases.emplace_back('X', fref[c] == 'C' ? 'T' : 'C', (uint8_t)(rnd() % 94));
                else
                    bases.emplace_back('=', fref[c], (uint8_t)(rnd() % 94));
            }
            if (urand() < 0.1) insert("TT");  // trailing: dropped
            std::printf("READ %d %d ", start0 + a, start0 + b);
            for (const auto& x : bases) std::putchar(x.Cigar);
            std::putchar(' ');
            for (const auto& x : bases) std::putchar(x.Nucleotide);
            std::putchar(' ');
            for (size_t i = 0; i < bases.size(); ++i) std::printf("%s%d", i ? "," : "", bases[i].QualQV);
            std::putchar('\n');
            reads.emplace_back(r, "read/" + std::to_string(r), start0 + a, start0 + b, bases);
        }
        const Fuse::Fuse fuse(reads);
        std::printf("FUSE fuse%d 50 %s\n", n_reads, fuse.ConsensusSequence().c_str());
        const Fuse::Fuse fuse10(reads, 10);
        std::printf("FUSE fuse%d 10 %s\n", n_reads, fuse10.ConsensusSequence().c_str());
        const Data::MSAByRow all(reads, Data::InsertionMode::All);  // what MSAColumn::Insertions() holds in the reference
        const Data::MSAByColumn cols(all);
        DumpRows(("fuse" + std::to_string(n_reads)).c_str(), all, false);  // the plain-layout set is resident: no second pack
        int k = 0;
        for (const auto& c : cols) {
            if (!c.Insertions().empty()) {
                std::printf("INS fuse%d %d", n_reads, k);
                for (const auto& kv : c.Insertions()) std::printf(" %s:%d", kv.first.c_str(), kv.second);
                std::putchar('\n');
            }
            ++k;
        }
    }
    {  // MSAColumn's result slots (include/pacbio/data/MSA.h:178-191, MSA.inl:82-112, src/MSA.cpp:223-237)
        std::vector<Data::ArrayRead> two{Data::ArrayRead(0, "a", 0, 2, {Data::ArrayBase('=', 'A', 9), Data::ArrayBase('I', 'G', 9), Data::ArrayBase('=', 'C', 9)}),
                                         Data::ArrayRead(1, "b", 0, 2, {Data::ArrayBase('D', '-', 0), Data::ArrayBase('I', 'T', 9), Data::ArrayBase('=', 'C', 9)})};
        const Data::MSAByRow rows(two, Data::InsertionMode::All);
        Data::MSAByColumn cols(rows);
        Data::MSAColumn& col = cols[1];
        std::cout << "COLUMN before " << col.PValue('A') << " " << col.PValue('-') << " " << col.SignificantInsertions().size() << " | " << cols[0] << "|\n";
        Data::FisherResult f;
        f.PValues = {{0.5, 0.25, 0.125, 1, 0.001, 1}};
        f.Mask = {{0, 0, 0, 0, 1, 0}};
        f.Hit = true;
        f.ArgMax = 1;
        col.AddFisherResult(f);
        col.AddFisherResult(std::map<std::string, double>{{"G", 0.5}, {"T", 0.001}});
        std::cout << "COLUMN after " << col.PValue('A') << " " << col.PValue('-') << " " << col.SignificantInsertions().size() << " "
                  << col.SignificantInsertions().at(0) << " | ";
        col.InDels(std::cout);
    }
    // exceptions keep the reference's types and texts
    try {
        std::vector<Data::ArrayRead> bad{Data::ArrayRead(0, "x", 0, 3, {Data::ArrayBase('=', 'A', 9), Data::ArrayBase('N', 'C', 9), Data::ArrayBase('=', 'G', 9)})};
        Data::MSAByRow m(bad);
        std::printf("ERR none\n");
    } catch (const std::runtime_error& e) {
        std::printf("ERR runtime_error %s\n", e.what());
    }
    try {
        std::vector<Data::ArrayRead> bad{Data::ArrayRead(0, "x", 0, 3, {Data::ArrayBase('=', 'A', 9), Data::ArrayBase('=', 'R', 9), Data::ArrayBase('=', 'G', 9)})};
        Data::MSAByRow m(bad);
        std::printf("ERR none\n");
    } catch (const std::runtime_error& e) {
        std::printf("ERR runtime_error %s\n", e.what());
    }
    try {
        std::printf("READS out_of_range\n");
        auto reads = MakeReads(ref.substr(0, 30), start0, 5, false);
        Juliet::JulietSettings c;
        c.TargetConfigUser.targetGenes.emplace_back(41, 400, "long", std::vector<Juliet::DRM>{});
        c.TargetConfigUser.referenceSequence = full.substr(0, 100);
        Juliet::AminoAcidCaller aac(reads, error, c);
        std::printf("ERR none\n");
    } catch (const std::out_of_range& e) {
        std::printf("ERR out_of_range %s\n", e.what());
    }
    try {
        const Fuse::Fuse none(std::vector<Data::ArrayRead>{});
        std::printf("ERR none\n");
    } catch (const std::runtime_error& e) {
        std::printf("ERR runtime_error %s\n", e.what());
    }
    return 0;
}
