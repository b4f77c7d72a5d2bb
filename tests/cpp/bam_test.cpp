// bam_test.cpp -- drives the BAM decode front (minorseq_amd/cpp/msq_bam.hpp) the way juliet's workflow does
// (src/tools/JulietWorkflow.cpp:136-138: BamUtils::BamToArrayReads(path, regionStart, regionEnd)) and prints every
// ArrayRead it produced, for tests/test_bam_front.py to compare with expectations computed from the records the
// test itself wrote.  With --call the reads also go through AminoAcidCaller (GPU) and the result is dumped in
// front_test's format.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <malloc.h>

#include "../../minorseq_amd/cpp/msq_bam.hpp"
#include "dump.hpp"

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::fprintf(stderr, "usage: bam_test file.bam regionStart regionEnd [--call] [--time] [--threads N]\n");
        return 2;
    }
    bool call = false, time_only = false, direct = false, on_device = false;
    unsigned threads = 4;
    for (int i = 4; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--call")) call = true;
        if (!std::strcmp(argv[i], "--direct")) direct = true;
        if (!std::strcmp(argv[i], "--device")) direct = on_device = true;  // the file decoded on the GPU (BamToReadSourceDevice)
        if (!std::strcmp(argv[i], "--time")) time_only = true;
        if (!std::strcmp(argv[i], "--top-pad") && i + 1 < argc) mallopt(M_TOP_PAD, std::atoi(argv[++i]) << 20);  // experiment: arena growth step, MB
        if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = (unsigned)std::atoi(argv[++i]);
    }
    bool scan_check = false;
    for (int i = 4; i < argc; ++i)
        if (!std::strcmp(argv[i], "--scan-check")) scan_check = true;
    try {
        IO::BamTimings tm;
        if (scan_check) {  // the BGZF member table found in ranges by several threads against the serial walk's
            using namespace IO::detail;
            const MappedFile file(argv[1]);
            size_t total_serial = 0, total_ranges = 0;
            bool in_ranges = false;
            std::string error_serial, error_ranges;
            std::vector<BgzfBlock> serial, ranges;
            try {
                serial = ScanBgzfSerial(file, &total_serial);
            } catch (const std::exception& e) {
                error_serial = e.what();
            }
            try {
                ranges = ScanBgzf(file, &total_ranges, threads, &in_ranges);
            } catch (const std::exception& e) {
                error_ranges = e.what();
            }
            bool same = serial.size() == ranges.size() && total_serial == total_ranges && error_serial == error_ranges;
            for (size_t b = 0; same && b < serial.size(); ++b)
                same = serial[b].in_off == ranges[b].in_off && serial[b].in_len == ranges[b].in_len && serial[b].out_off == ranges[b].out_off &&
                       serial[b].out_len == ranges[b].out_len;
            std::printf("SCAN members %zu total %zu same %d in_ranges %d error '%s'\n", serial.size(), total_serial, (int)same, (int)in_ranges, error_serial.c_str());
            return same ? 0 : 1;
        }
        if (direct) {  // BAM records decoded straight into the device packer's staging: no ArrayRead objects
            const Juliet::ErrorEstimates error("S/P2-C2");
            Juliet::JulietSettings s;
            double call_s[2] = {0, 0}, load_s[2] = {0, 0};
            for (int rep = 0; rep < (time_only ? 2 : 1); ++rep) {
                const auto l0 = std::chrono::steady_clock::now();
                std::vector<std::string> chems;
                bool used_device = false;
                const Data::ReadSource source = on_device ? IO::BamUtils::BamToReadSourceDevice(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), threads, &tm, &chems, &used_device)
                                                          : IO::BamUtils::BamToReadSource(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), threads, &tm, &chems);
                if (on_device && !time_only && rep == 0) std::printf("FRONT %s\n", used_device ? "device" : "host");
                const auto c0 = std::chrono::steady_clock::now();
                Juliet::AminoAcidCaller aac(source, error, s);
                const auto c1 = std::chrono::steady_clock::now();
                load_s[rep] = std::chrono::duration<double>(c0 - l0).count();
                call_s[rep] = std::chrono::duration<double>(c1 - c0).count();
                if (!time_only) {
                    std::printf("READS bam\n");
                    for (size_t r = 0; r < source.names.size(); ++r)
                        std::printf("SRC %d %d %lld NAME %s\n", source.referenceStart[r], source.referenceEnd[r], (long long)source.numBases[r],
                                    source.names[r].c_str());
                    for (const auto& c : chems) std::printf("CHEMS %s\n", c.empty() ? "." : c.c_str());
                    Dump("bam", aac);
                    aac.PhaseVariants();
                    DumpReports("bam", aac);
                }
            }
            if (time_only) {
                size_t bases = 0;
                std::printf("TIMING threads %u file_bytes %zu bam_bytes %zu records %zu kept %zu bases %zu read_s %.4f inflate_s %.4f index_s %.4f parse_s %.4f "
                            "call_first_s %.4f call_s %.4f load_s %.4f\n",
                            threads, tm.file_bytes, tm.bam_bytes, tm.records, tm.kept, bases, tm.read, tm.inflate, tm.index, tm.parse, call_s[0], call_s[1],
                            load_s[1]);
            }
            return 0;
        }
        const auto reads = IO::BamUtils::BamToArrayReads(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), threads, &tm);
        if (time_only) {  // decode rate of the front on its own (host only)
            size_t bases = 0;
            for (const auto& r : reads) bases += r->Bases().size();
            double call_s = 0, call2_s = 0;
            if (call) {  // + ArrayReads -> packed upload -> pileup -> Fisher (AminoAcidCaller's constructor), twice
                const Juliet::ErrorEstimates error("S/P2-C2");
                Juliet::JulietSettings s;
                for (int rep = 0; rep < 2; ++rep) {
                    const auto c0 = std::chrono::steady_clock::now();
                    Juliet::AminoAcidCaller aac(reads, error, s);
                    (rep ? call2_s : call_s) = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
                    if (rep) {
                        const auto p0 = std::chrono::steady_clock::now();
                        aac.PhaseVariants();
                        const auto p1 = std::chrono::steady_clock::now();
                        const std::string json = aac.JSON().dump(2);
                        const auto p2 = std::chrono::steady_clock::now();
                        std::printf("PHASE_S %.4f JSON_S %.4f haplotypes %zu json_bytes %zu\n", std::chrono::duration<double>(p1 - p0).count(),
                                    std::chrono::duration<double>(p2 - p1).count(), aac.ReconstructedHaplotypes().size(), json.size());
                    }
                }
            }
            std::printf("TIMING threads %u file_bytes %zu bam_bytes %zu records %zu kept %zu bases %zu read_s %.4f inflate_s %.4f index_s %.4f parse_s %.4f "
                        "call_first_s %.4f call_s %.4f\n",
                        threads, tm.file_bytes, tm.bam_bytes, tm.records, tm.kept, bases, tm.read, tm.inflate, tm.index, tm.parse, call_s, call2_s);
            return 0;
        }
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
            std::printf(" NAME %s CHEM %s\n", r->Name().c_str(), r->SequencingChemistry().empty() ? "." : r->SequencingChemistry().c_str());
        }
        if (call) {
            const Juliet::ErrorEstimates error("S/P2-C2");
            Juliet::JulietSettings s;
            Juliet::AminoAcidCaller aac(reads, error, s);
            Dump("bam", aac);
            aac.PhaseVariants();
            DumpReports("bam", aac);
        }
    } catch (const std::exception& e) {
        std::printf("ERR %s\n", e.what());
        return 1;
    }
    return 0;
}
