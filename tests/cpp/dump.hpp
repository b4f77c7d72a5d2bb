// dump.hpp -- the text dump of everything an AminoAcidCaller computed, shared by front_test.cpp and bam_test.cpp
// (parsed by tests/test_gpu_cpp_front.py).
#pragma once
#include <cstdio>
#include <sstream>
#include <string>

#include "../../minorseq_amd/cpp/msq_front.hpp"

using namespace PacBio;

static std::string Tok(std::string s)  // one token per name in the dump
{
    for (auto& c : s)
        if (c == ' ') c = '_';
    return s;
}

static void Dump(const char* name, Juliet::AminoAcidCaller& aac)
{
    const auto& rows = aac.Rows();
    std::printf("CASE %s\nBEGIN %d END %d\n", name, rows.BeginPos(), rows.EndPos());
    for (const auto& col : aac.msaByColumn_)
        std::printf("NUC %d %d %d %d %d %d %d COV %d MAX %c\n", col.RefPos(), col['A'], col['C'], col['G'], col['T'], col['-'],
                    col['N'], col.Coverage(), col.MaxBase() == ' ' ? '_' : col.MaxBase());
    for (int i = 0; i < rows.Width(); ++i) {
        const auto codons = rows.CodonsAt(i);
        if (codons.empty()) continue;
        std::printf("CODON %d", i);
        for (const auto& kv : codons) std::printf(" %s:%d", kv.first.c_str(), kv.second);
        std::putchar('\n');
    }
    std::printf("NTESTS %d\n", aac.NumberOfTests());
    for (const auto& vg : aac.VariantGenes())
        for (const auto& pv : vg.relPositionToVariant) {
            const auto& vp = *pv.second;
            for (const auto& aa : vp.aminoAcidToCodons)
                for (const auto& vc : aa.second)
                    std::printf("VAR %s %d %s %c %s %.17g %.17g %s COV %d\n", Tok(vg.geneName).c_str(), pv.first, vp.refCodon.c_str(),
                                aa.first, vc.codon.c_str(), vc.frequency, vc.pValue, vc.knownDRM.empty() ? "-" : Tok(vc.knownDRM).c_str(),
                                vp.coverage);
            for (const auto& m : vp.msa)
                std::printf("MSA %s %d %d %d %d %d %d %d %d %d %s\n", Tok(vg.geneName).c_str(), pv.first, m.rel_pos, m.abs_pos, m.A, m.C, m.G,
                            m.T, m.gap, m.N, m.wt == " " ? "_" : m.wt.c_str());
        }
    const auto& pm = aac.Performance();
    std::printf("PERF %g %g %g %g\n", pm.TruePositives, pm.FalsePositives, pm.FalseNegative, pm.TrueNegative);
}

// The report (AminoAcidCaller::JSON().dump(2), src/tools/JulietWorkflow.cpp:165-170), the --msa table (:178-191) and
// the --mode-error lines (:193-213), each between marker lines.
[[maybe_unused]] static void DumpReports(const char* name, const Juliet::AminoAcidCaller& aac)
{
    std::printf("JSON %s\n", name);
    std::ostringstream js;
    Juliet::JulietWorkflow::WriteJson(js, aac);
    std::fputs(js.str().c_str(), stdout);
    std::printf("ENDJSON\nMSATABLE %s\n", name);
    std::ostringstream ms;
    Juliet::JulietWorkflow::WriteMsa(ms, aac.msaByColumn_);
    std::fputs(ms.str().c_str(), stdout);
    std::printf("ENDMSATABLE\nERRMODE %s\n", name);
    std::ostringstream es;
    try {
        Juliet::JulietWorkflow::Error(es, "input.bam", aac.msaByColumn_);
        std::fputs(es.str().c_str(), stdout);
    } catch (const std::out_of_range&) {
        // a column whose majority is 'N' has MaxBase() ' ', and Frequency(' ') is counts_.at(255): the reference throws
        // the same std::out_of_range (NucleotideConversion.h:84-85, MSA.inl:58-61)
        std::printf("out_of_range\n");
    }
    std::printf("ENDERRMODE\n");
}
