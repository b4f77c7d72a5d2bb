// config_test.cpp -- host-only: the target configuration parser of the front (TargetConfig(input),
// src/TargetConfig.cpp:120-265) and the JSON reader under it.  Reads one configuration input per command-line
// argument (a JSON string or a file path) and dumps what was parsed; `--json TEXT` parses TEXT and prints it back
// compactly.  No GPU call is made.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../minorseq_amd/cpp/msq_front.hpp"

using namespace PacBio;

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        try {
            if (!std::strcmp(argv[i], "--json") && i + 1 < argc) {
                std::printf("JSON %s\n", JSON::Json::parse(argv[++i]).dump().c_str());
                continue;
            }
            const Juliet::TargetConfig tc{std::string(argv[i])};
            std::printf("CONFIG name=%s version=%s db=%s ref=%s genes=%zu minors=%zu\n", tc.referenceName.c_str(), tc.version.c_str(),
                        tc.dbVersion.c_str(), tc.referenceSequence.c_str(), tc.targetGenes.size(), tc.NumExpectedMinors());
            for (const auto& g : tc.targetGenes) {
                std::printf("GENE %s %d %d\n", g.name.c_str(), g.begin, g.end);
                for (const auto& d : g.drms) {
                    std::printf("DRM %s", d.name.c_str());
                    for (const auto& m : d.positions) std::printf(" %c%d%c", m.refAA, m.pos, m.curAA);
                    std::printf("\n");
                }
                for (const auto& m : g.minors) std::printf("MINOR %d %s %s\n", m.position, m.aminoacid.c_str(), m.codon.c_str());
            }
        } catch (const std::exception& e) {
            std::printf("ERR %s\n", e.what());
        }
    }
    return 0;
}
