// dump.hpp -- the text dump of everything an AminoAcidCaller computed, shared by front_test.cpp and bam_test.cpp
// (parsed by tests/test_gpu_cpp_front.py).
#pragma once
#include <cstdio>
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
