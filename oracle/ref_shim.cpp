// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" doorway into the reference's OWN code, compiled unmodified from where it lies under
// $MINORSEQ_REF (default /root/reference) by oracle/Makefile target `ref` into oracle/_ref/libmsq_ref.so.
// Only the four reference sources that build with nothing but the reference's include/ directory are
// used: src/Fisher.cpp, src/AminoAcidTable.cpp, src/ErrorEstimates.cpp, src/NucleotideConversion.cpp.
// (MSA.cpp / AminoAcidCaller.cpp need boost, pbbam and pbcopper, absent from this image: unbuildable.)
// No reference source is copied into this repository; this file only calls the reference's public API.

#include <cstdint>
#include <stdexcept>
#include <string>

#include <pacbio/data/NucleotideConversion.h>
#include <pacbio/juliet/AminoAcidTable.h>
#include <pacbio/juliet/ErrorEstimates.h>
#include <pacbio/statistics/Fisher.h>

extern "C" {

// include/pacbio/statistics/Fisher.h:15
double ref_fisher_exact_tiss(int chi11, int chi12, int chi21, int chi22)
{
    return PacBio::Statistics::Fisher::fisher_exact_tiss(chi11, chi12, chi21, chi22);
}

// src/AminoAcidTable.cpp:43-54.  Returns 0 when the 3-char string is not a key.
char ref_amino_acid(const char* codon3)
{
    const auto& tab = PacBio::Juliet::AminoAcidTable::FromCodon;
    const auto it = tab.find(std::string(codon3, 3));
    return it == tab.cend() ? 0 : it->second;
}
int ref_amino_acid_table_size() { return (int)PacBio::Juliet::AminoAcidTable::FromCodon.size(); }

// src/ErrorEstimates.cpp:44-49 and :61-67.  out = {Match, Substitution, Deletion, Insertion}
void ref_error_estimates_chemistry(const char* chemistry, double* out)
{
    const PacBio::Juliet::ErrorEstimates e{std::string(chemistry)};
    out[0] = e.Match; out[1] = e.Substitution; out[2] = e.Deletion; out[3] = e.Insertion;
}
void ref_error_estimates_rates(double sub, double del, double* out)
{
    const PacBio::Juliet::ErrorEstimates e(sub, del);
    out[0] = e.Match; out[1] = e.Substitution; out[2] = e.Deletion; out[3] = e.Insertion;
}

// include/pacbio/data/NucleotideConversion.h:44-45 (C++11 declarations; src/NucleotideConversion.cpp)
// The C++11 definitions throw on unknown input; reported here as 255 / '?'.
int ref_nucleotide_to_tag(char c)
{
    try { return PacBio::Data::NucleotideToTag(c); } catch (const std::exception&) { return 255; }
}
char ref_tag_to_nucleotide(int t)
{
    try { return PacBio::Data::TagToNucleotide((uint8_t)t); } catch (const std::exception&) { return '?'; }
}
}
