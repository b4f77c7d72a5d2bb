// bam_test.cpp -- drives the BAM decode front (minorseq_amd/cpp/msq_bam.hpp) the way juliet's workflow does
// (src/tools/JulietWorkflow.cpp:136-138: BamUtils::BamToArrayReads(path, regionStart, regionEnd)) and prints every
// ArrayRead it produced, for tests/test_bam_front.py to compare with expectations computed from the records the
// test itself wrote.  With --call the reads also go through AminoAcidCaller (GPU) and the result is dumped in
// front_test's format.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../minorseq_amd/cpp/msq_bam.hpp"
#include "dump.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: bam_test file.bam regionStart regionEnd [--call] [--threads N]\n");
        return 2;
    }
    bool call = false;
    unsigned threads = 4;
    for (int i = 4; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--call")) call = true;
        if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = (unsigned)std::atoi(argv[++i]);
    }
    try {
        const auto reads = IO::BamUtils::BamToArrayReads(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), threads);
        std::printf("READS bam\n");
        for (const auto& r : reads) {
            const auto& bases = r->Bases();
            std::printf("READ %d %d ", r->ReferenceStart(), r->ReferenceEnd());
            for (const auto& x : bases) std::putchar(x.Cigar);
            std::putchar(' ');
            for (const auto& x : bases) std::putchar(x.Nucleotide);
            const bool rich = !bases.empty() && bases[0].HasRichQVs;
            for (int k = 0; k < (rich ? 4 : 1); ++k) {
                std::putchar(' ');
                for (size_t i = 0; i < bases.size(); ++i) {
                    const auto& x = bases[i];
                    std::printf("%s%d", i ? "," : "", k == 0 ? x.QualQV : k == 1 ? x.SubQV : k == 2 ? x.DelQV : x.InsQV);
                }
            }
            std::printf(" NAME %s\n", r->Name().c_str());
        }
        if (call) {
            const Juliet::ErrorEstimates error("S/P2-C2");
            Juliet::JulietSettings s;
            Juliet::AminoAcidCaller aac(reads, error, s);
            Dump("bam", aac);
        }
    } catch (const std::exception& e) {
        std::printf("ERR %s\n", e.what());
        return 1;
    }
    return 0;
}
