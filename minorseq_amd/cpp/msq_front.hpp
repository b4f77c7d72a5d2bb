// msq_front.hpp -- C++ host front of the MI355X pileup + Fisher layer: the reference's MSA / Fisher /
// AminoAcidCaller interface (same names, argument meaning and error behaviour) on top of the C ABI
// (include/msq_abi.h).  A maintainer swaps the reference's data/statistics layer for this header + libmsq_hip.so;
// the callers (JulietWorkflow::AminoPhasing / ::Error, Fuse) keep compiling against the same class names.
//
// What is mirrored (reference file:line):
//   PacBio::Data::ArrayBase / ArrayRead        include/pacbio/data/ArrayRead.h:60-119 (boost::optional QVs -> has* flags)
//   PacBio::Data::QvThresholds                 include/pacbio/data/QvThresholds.h:44-54, src/QvThresholds.cpp:45-56
//   PacBio::Data::MSAByRow                     include/pacbio/data/MSA.h:59-93   (BeginPos/EndPos 1-based, CodonsAt)
//   PacBio::Data::MSAByColumn / MSAColumn      include/pacbio/data/MSA.h:115-214 (operator[](abs0), Coverage, Frequency, MaxBase)
//   PacBio::Statistics::Fisher                 include/pacbio/statistics/Fisher.h:12-25
//   PacBio::Juliet::ErrorEstimates             include/pacbio/juliet/ErrorEstimates.h:49-61
//   PacBio::Juliet::TargetGene/DRM/DMutation/ExpectedMinor/TargetConfig (data members only) TargetConfig.h:49-139
//   PacBio::Juliet::VariantGene                include/pacbio/juliet/VariantGene.h:49-81 (msa context as plain structs)
//   PacBio::Juliet::AminoAcidCaller            include/pacbio/juliet/AminoAcidCaller.h:90-154 (CallVariants + store policy)
//   PacBio::Data::MSARow                       include/pacbio/data/MSA.h:96-110  (Bases, Insertions, Read, CodonAt, CodingCodonAt)
// Per-read rows (MSAByRow::Rows() / NameToRow) are decoded on first use from the packed reads still in HBM
// (msq_rows_decode + the insertion scan); the path itself never needs them -- PhaseVariants gets the codons it asks
// for from the device (MSAByRow::PhaseKeys).  Also here: the JSON report
// (msq_json.hpp), the --msa / --mode-error writers, MSAColumn::Insertions() and PacBio::Fuse::Fuse
// (include/pacbio/fuse/Fuse.h:50-72), TargetConfig(input) from a JSON string or file.  Not mirrored: the HTML
// report, the reference's built-in HIV configuration tables.
//
// Errors: ABI status codes come back as the reference's exceptions -- std::runtime_error("Unexpected cigar ..") /
// ("Unexpected base ..") (src/MSA.cpp:85,182), std::out_of_range (referenceSequence.substr,
// src/AminoAcidCaller.cpp:446); anything else (no GPU, HIP failure) is a std::runtime_error carrying
// msq_last_error().  There is no CPU fallback.
#pragma once

#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <cstddef>
#include <chrono>
#include <cstdint>
#include <exception>
#include <functional>
#include <limits>
#include <mutex>
#include <map>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/msq_abi.h"
#include "msq_json.hpp"

namespace PacBio {
namespace Msq {
// ---- plumbing: one context per process (the reference is single-threaded)
inline void Check(int rc)
{
    if (rc == MSQ_OK) return;
    const std::string text = msq_last_error();
    if (rc == MSQ_ERR_RANGE) throw std::out_of_range(text);
    throw std::runtime_error(text);
}
inline msq_ctx* Context()
{
    static msq_ctx* ctx = [] {
        msq_ctx* c = nullptr;
        Check(msq_ctx_create(0, &c));
        return c;
    }();
    return ctx;
}
/// One rank of a read-sharded run (one process per GPU): binds the library's RCCL communicator to this process's
/// context.  `id` = the MSQ_COMM_ID_BYTES that msq_comm_unique_id() produced on rank 0, handed to every rank by the
/// caller (MPI, a file, ...).  From then on every MSAByRow / AminoAcidCaller takes the reads it is given as the WHOLE
/// sample, piles up only its own share [rank * N / world, (rank + 1) * N / world) and merges window bounds and count
/// matrices over the ranks (msq_comm_allreduce_*), so that every rank ends up with the same matrices, calls and
/// haplotypes as a single-GPU run.  Call it before the first MSA is built.
inline void SetCommunicator(int rank, int world, const uint8_t* id) { Check(msq_comm_create(Context(), rank, world, id)); }
inline int CommRank() { return msq_comm_rank(Context()); }
inline int CommWorld() { return msq_comm_world(Context()); }
inline bool CommActive() { return msq_comm_active(Context()) != 0; }
template <typename T>
class DeviceBuffer
{
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) : n_(n)
    {
        void* p = nullptr;
        Check(msq_dev_malloc(Context(), std::max<size_t>(n, 1) * sizeof(T), &p));
        p_ = static_cast<T*>(p);
        Check(msq_dev_memset(Context(), p_, 0, std::max<size_t>(n, 1) * sizeof(T)));
    }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer()
    {
        if (p_) msq_dev_free(Context(), p_);
    }
    T* get() const { return p_; }
    std::vector<T> ToHost() const { return ToHost(n_); }
    std::vector<T> ToHost(size_t count) const  // the first `count` elements
    {
        std::vector<T> h(std::min(count, n_));
        if (!h.empty()) Check(msq_memcpy_d2h(Context(), h.data(), p_, h.size() * sizeof(T)));
        return h;
    }

private:
    T* p_ = nullptr;
    size_t n_ = 0;
};
inline int CodonIndex(const std::string& s)  // -1 unless a key of AminoAcidTable::FromCodon
{
    if (s.size() != 3) return -1;
    int idx = 0;
    for (char c : s) {
        const int t = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
        if (t < 0) return -1;
        idx = idx * 4 + t;
    }
    return idx;
}
/// {"key", value} pairs -> a JSON object
inline JSON::Json Obj(std::initializer_list<std::pair<const char*, JSON::Json>> members)
{
    JSON::Json o;
    for (const auto& m : members) o[m.first] = m.second;
    return o;
}
inline std::string CodonString(int idx)
{
    static const char b[] = "ACGT";
    return std::string{b[(idx >> 4) & 3], b[(idx >> 2) & 3], b[idx & 3]};
}
}  // namespace Msq

namespace Juliet {
// AminoAcidTable::FromCodon (src/AminoAcidTable.cpp:43-54): standard code, stop -> 'X'
struct AminoAcidTable
{
    static char FromCodonIndex(int idx)
    {
        static const char* tcag = "FFLLSSSSYYXXCCXWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
        static const int t[4] = {2, 1, 3, 0};
        return tcag[16 * t[(idx >> 4) & 3] + 4 * t[(idx >> 2) & 3] + t[idx & 3]];
    }
    static bool Has(const std::string& codon) { return Msq::CodonIndex(codon) >= 0; }
    static char At(const std::string& codon)
    {
        const int i = Msq::CodonIndex(codon);
        if (i < 0) throw std::out_of_range("AminoAcidTable::FromCodon.at(" + codon + ")");
        return FromCodonIndex(i);
    }
};
}  // namespace Juliet

namespace Data {
// ---------------------------------------------------------------------------------------------- ArrayRead
struct QvThresholds  // include/pacbio/data/QvThresholds.h:44-54; -1 == boost::none
{
    QvThresholds()
    {
        msq_qv_thresholds t;
        const int rc = msq_qv_thresholds_from_env(&t);  // getenv + std::stoi, SUBQV default 42
        if (rc != MSQ_OK) throw std::invalid_argument(msq_last_error());
        DelQV = t.del; SubQV = t.sub; InsQV = t.ins; QualQV = t.qual;
    }
    int DelQV, SubQV, InsQV, QualQV;
};

struct ArrayBase  // include/pacbio/data/ArrayRead.h:96-119
{
    ArrayBase(char cigar, char nucleotide, uint8_t qualQV, uint8_t subQV, uint8_t delQV, uint8_t insQV)
        : Cigar(cigar), Nucleotide(nucleotide), HasQualQV(true), HasRichQVs(true), QualQV(qualQV), DelQV(delQV), SubQV(subQV), InsQV(insQV)
    {
    }
    ArrayBase(char cigar, char nucleotide, uint8_t qualQV) : Cigar(cigar), Nucleotide(nucleotide), HasQualQV(true), QualQV(qualQV) {}
    ArrayBase(char cigar, char nucleotide) : Cigar(cigar), Nucleotide(nucleotide) {}
    char Cigar;
    char Nucleotide;
    bool HasQualQV = false, HasRichQVs = false;
    uint8_t QualQV = 0, DelQV = 0, SubQV = 0, InsQV = 0;
};

class ArrayRead  // include/pacbio/data/ArrayRead.h:60-81
{
public:
    ArrayRead(const int idx = -1, const std::string& name = "") : idx_(idx), name_(name) {}
    ArrayRead(const int idx, const std::string& name, size_t referenceStart, size_t referenceEnd, std::vector<ArrayBase> bases)
        : bases_(std::move(bases)), idx_(idx), name_(name), referenceStart_(referenceStart), referenceEnd_(referenceEnd)
    {
    }
    virtual ~ArrayRead() = default;
    int ReferenceStart() const { return static_cast<int>(referenceStart_); }
    int ReferenceEnd() const { return static_cast<int>(referenceEnd_); }
    const std::vector<ArrayBase>& Bases() const { return bases_; }
    const std::string& Name() const { return name_; }
    virtual std::string SequencingChemistry() const { return ""; }

protected:
    std::vector<ArrayBase> bases_;
    const int idx_;
    const std::string name_;
    size_t referenceStart_ = 0;
    size_t referenceEnd_ = 0;
};

// ---------------------------------------------------------------------------------------------------- MSA
class MSAByColumn;

/// A particular row of a MSA (include/pacbio/data/MSA.h:96-110): what AddRead makes of one read.
struct MSARow
{
    MSARow(const int size) : Bases(size, ' ') {}
    /// Individual bases with '-' as deletion, 'N' for a base under a QV threshold, ' ' where the read has nothing.
    std::vector<char> Bases;
    /// Position to insertion string.
    std::map<int, std::string> Insertions;
    /// The underlying ArrayRead (set by the shared_ptr constructor of MSAByRow, as in src/MSA.cpp:99-105).
    std::shared_ptr<Data::ArrayRead> Read;

    /// The row's characters at window columns pos, pos + 1, pos + 2, as far as they exist -- and column 0 never
    /// does: the reference's loop asks for `i > 0` (src/MSA.cpp:188-195), which parity keeps.
    std::string CodonAt(const int pos) const
    {
        std::string codon;
        const int length = static_cast<int>(Bases.size());
        for (int i = std::max(pos, 1); i < std::min(pos + 3, length); ++i) codon += Bases[i];
        return codon;
    }
    /// true and *codon set when the read covers the three columns without a gap and CodonAt is one of the 64
    /// codons of the amino-acid table (src/MSA.cpp:197-219); at winPos 0 CodonAt has two characters and never is.
    bool CodingCodonAt(const int winPos, std::string* codon) const
    {
        if (winPos < 0 || winPos + 2 >= static_cast<int>(Bases.size())) return false;
        for (int i = winPos; i < winPos + 3; ++i)
            if (Bases[i] == ' ' || Bases[i] == '-') return false;
        const std::string proposed = CodonAt(winPos);
        if (!Juliet::AminoAcidTable::Has(proposed)) return false;
        *codon = proposed;
        return true;
    }
};

/// MSAByRow: the pileup.  The reads are packed (msq_pack_reads), uploaded, and one msq_pileup builds both count
/// matrices on the GPU; CodonsAt() and MSAByColumn read them back.
/// Which per-column insertion histograms (MSARow::Insertions -> MSAColumn::Insertions(), src/MSA.cpp:88-90,151-156)
/// an MSAByRow collects.  juliet never reads them (None); fuse only reads strings above half a column's coverage
/// (Majority: the columns whose insertion count can reach that); All = every column, as the reference always does.
enum class InsertionMode { None, Majority, All };

/// Reads that never become ArrayRead objects: a producer (the BAM front, msq_bam.hpp) that knows every read's
/// name, reference interval and number of ArrayBases, and writes a read's ArrayBases where it is told to -- the
/// pinned staging buffer of the device packer.  What BAMArrayRead's constructor computes (src/ArrayRead.cpp:53-99),
/// without the per-read vectors in between.  fill is called concurrently from several threads.
struct ReadSource
{
    std::vector<std::string> names;
    std::vector<int32_t> referenceStart, referenceEnd;
    std::vector<int64_t> numBases;
    std::function<void(uint64_t read, ArrayBase* dst)> fill;
    /// Optional: the same read as one translated byte per base (msq_pack_reads_device_fill_compact in msq_abi.h: the
    /// table over (Cigar, Nucleotide), msq_compact_fail on a base that misses a threshold); returns -1 or
    /// (index of the first base with an unknown cigar) << 8 | that cigar.  A producer that decodes and translates in
    /// one go (the BAM front) saves the 8-byte records; used when present.
    std::function<int64_t(uint64_t read, uint8_t* dst, const msq_qv_thresholds& thr, uint32_t* presence)> fillCompact;
    /// Optional: the reads are in HBM already (the BAM front on the device, BamUtils::BamToReadSourceDevice): packs reads
    /// [first, first + count) there, thresholds applied (msq_bam_pack_device); used when present, and fill / fillCompact
    /// then only serve the rare look-ups of literal bases.
    std::function<int(uint32_t packFlags, size_t first, size_t count, const msq_qv_thresholds& thr, msq_packed_dev** dev)> packDevice;
};

class MSAByRow  // include/pacbio/data/MSA.h:59-93
{
public:
    MSAByRow() = default;
    /// The source must stay alive while Rows() / NameToRow() may still be called for the first time (they decode
    /// lazily) and while ReadNames() is used (the names are the source's, not copied); the counts, CodonsAt and
    /// MSAByColumn need nothing from it after construction.
    MSAByRow(const ReadSource& source, InsertionMode insertions = InsertionMode::None)
    {
        const QvThresholds qv;
        thr_ = msq_qv_thresholds{qv.QualQV, qv.DelQV, qv.SubQV, qv.InsQV};
        const auto tr0 = std::chrono::steady_clock::now();
        numReads_ = source.names.size();
        sourceNames_ = &source.names;  // not copied (10^5 strings are 7 ms): the source outlives the MSA's use of them, see above
        const auto tr1 = std::chrono::steady_clock::now();
        const ReadSource* src = &source;
        TakeShare(insertions, [src](size_t k) { return (uint64_t)src->numBases[k]; });
        pack_ = [src](uint32_t packFlags, size_t first, size_t count, const msq_qv_thresholds& thr, msq_packed_dev** dev,
                      std::exception_ptr* thrown) { return PackSource(*src, packFlags, first, count, thr, dev, thrown); };
        basesOf_ = [src](uint64_t read, uint32_t first, uint32_t len) {
            std::vector<ArrayBase> bases((size_t)src->numBases[read], ArrayBase('P', '*'));
            src->fill(read, bases.data());
            std::string s(len, 'A');
            for (uint32_t i = 0; i < len; ++i) s[i] = bases[first + i].Nucleotide;
            return s;
        };
        msq_packed_dev* packed = PackShare(insertions == InsertionMode::None ? 0u : MSQ_PACK_PLAIN);
        const auto tr2 = std::chrono::steady_clock::now();
        Finish(packed, numReads_ == 0, insertions);
        if (std::getenv("MSQ_BAM_TRACE")) {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "MSAByRow(ReadSource): names %.2f ms, pack %.2f, bounds + pileup + counts %.2f\n", ms(tr0, tr1), ms(tr1, tr2), ms(tr2, std::chrono::steady_clock::now()));
        }
    }
    MSAByRow(const std::vector<std::shared_ptr<Data::ArrayRead>>& reads, InsertionMode insertions = InsertionMode::None)
    {
        std::vector<const ArrayRead*> p;
        for (const auto& r : reads) p.push_back(r.get());
        kept_ = reads;  // row.Read = r (src/MSA.cpp:101): the rows share the reads
        Build(std::move(p), insertions);
    }
    /// The vector must stay alive while Rows() / NameToRow() may still be called for the first time (see above).
    MSAByRow(const std::vector<Data::ArrayRead>& reads, InsertionMode insertions = InsertionMode::None)
    {
        std::vector<const ArrayRead*> p;
        for (const auto& r : reads) p.push_back(&r);
        Build(std::move(p), insertions);
    }
    /// Per window column the insertion strings and their row counts (empty vector with InsertionMode::None).
    const std::vector<std::map<std::string, int>>& Insertions() const { return insertions_; }
    /// The left-most / right-most position of all reads, 1-based after construction (src/MSA.cpp:107-108).
    int BeginPos() const { return beginPos_; }
    int EndPos() const { return endPos_; }
    int Width() const { return width_; }
    /// The individual rows of the MSA (MSA.h:72).  The counts never need them, so they are decoded on first use: the
    /// characters by msq_rows_decode from the packed reads still resident in HBM, the insertion strings by the
    /// insertion scan (msq_insertion_records) over a plain-layout set -- the resident one when this MSA collects
    /// insertions, else one packed again from the reads for this purpose.
    const std::vector<std::shared_ptr<MSARow>>& Rows() const
    {
        Materialise();
        return rows_;
    }
    /// Easy way to access a row by its name (MSA.h:75; std::out_of_range for an unknown name, as .at() there).
    std::shared_ptr<MSARow> NameToRow(const std::string& name) const
    {
        Materialise();
        return nameToRow_.at(name);
    }
    /// Codon histogram at window column i (src/MSA.cpp:126-136): only observed codons appear, in codon order.
    std::map<std::string, int> CodonsAt(const int i) const
    {
        std::map<std::string, int> codons;
        if (i < 0 || i >= width_) return codons;
        for (int k = 0; k < 64; ++k)
            if (const uint32_t n = codon_[(size_t)i * 64 + k]) codons[Msq::CodonString(k)] = static_cast<int>(n);
        return codons;
    }
    const std::vector<uint32_t>& NucleotideCounts() const { return nuc_; }  // [W][6]
    const std::vector<uint32_t>& CodonCounts() const { return codon_; }     // [W][64]
    const uint32_t* DeviceCodonCounts() const { return dCounts_ ? dCounts_->get() + (size_t)width_ * 6 : nullptr; }
    size_t NumReads() const { return numReads_; }
    /// Names of the reads, in row order (MSARow::Read->Name(), src/MSA.cpp:100-104).
    const std::vector<std::string>& ReadNames() const { return sourceNames_ ? *sourceNames_ : names_; }
    /// MSARow::CodonAt(pos[k]) of every row, as msq_phase_codons keys: [row][k], three 3-bit symbols each
    /// (0 ' ', 1..4 ACGT, 5 '-', 6 'N', 7 = index left out by CodonAt); src/MSA.cpp:188-195.
    std::vector<uint16_t> PhaseKeys(const std::vector<int32_t>& pos) const
    {
        std::vector<uint16_t> keys(numReads_ * pos.size());
        if (keys.empty()) return keys;
        Msq::DeviceBuffer<uint16_t> dKeys(keys.size());
        GatherKeys(pos, dKeys.get());
        return dKeys.ToHost();
    }
    /// The rows of the MSA grouped by their codons at the given positions: rows with equal CodonAt strings at every
    /// position form one group; groups are numbered by their first row (the order in which the reference's pairwise
    /// comparison opens its observations, src/AminoAcidCaller.cpp:162-196).  Gather and grouping run on the GPU
    /// (msq_phase_codons, msq_phase_cluster); what comes back is one id per row and one key row per group.
    struct RowGroups
    {
        size_t numPositions = 0;
        std::vector<uint32_t> groupOfRow;  // [rows]
        std::vector<uint32_t> firstRow;    // [groups]
        std::vector<uint32_t> size;        // [groups]
        std::vector<uint16_t> keys;        // [groups][numPositions]
        size_t NumGroups() const { return firstRow.size(); }
    };
    RowGroups GroupRowsByCodons(const std::vector<int32_t>& pos) const
    {
        RowGroups g;
        g.numPositions = pos.size();
        if (numReads_ == 0) return g;
        const size_t K = pos.size();
        msq_ctx* ctx = Msq::Context();
        Msq::DeviceBuffer<uint16_t> dKeys(numReads_ * K);
        if (K) GatherKeys(pos, dKeys.get());  // every read's keys, on every rank
        Msq::DeviceBuffer<uint32_t> dGroup(numReads_), dFirst(numReads_), dSize(numReads_);
        Msq::DeviceBuffer<uint16_t> dRows(numReads_ * K);
        uint64_t n = 0;
        const int rc = msq_phase_cluster(ctx, dKeys.get(), numReads_, (int32_t)K, dGroup.get(), dFirst.get(), dSize.get(), dRows.get(), numReads_, &n);
        if (rc == MSQ_ERR_RANGE) {  // two different key rows with one 64-bit hash: group on the host
            const std::vector<uint16_t> keys = dKeys.ToHost();
            std::unordered_map<std::string, uint32_t> seen;
            g.groupOfRow.resize(numReads_);
            for (size_t r = 0; r < numReads_; ++r) {
                const std::string row(reinterpret_cast<const char*>(keys.data() + r * K), K * sizeof(uint16_t));
                const auto at = seen.emplace(row, (uint32_t)g.firstRow.size());
                if (at.second) {
                    g.firstRow.push_back((uint32_t)r);
                    g.size.push_back(0);
                    g.keys.insert(g.keys.end(), keys.begin() + r * K, keys.begin() + (r + 1) * K);
                }
                g.groupOfRow[r] = at.first->second;
                ++g.size[at.first->second];
            }
            return g;
        }
        Msq::Check(rc);
        g.groupOfRow = dGroup.ToHost();
        g.firstRow = dFirst.ToHost(n);
        g.size = dSize.ToHost(n);
        g.keys = dRows.ToHost(n * K);
        return g;
    }
    /// The string MSARow::CodonAt returns for one key of PhaseKeys.
    static std::string KeyToCodon(uint16_t key)
    {
        static const char sym[] = " ACGT-N";
        std::string codon;
        for (int j = 0; j < 3; ++j) {
            const int s = (key >> (3 * j)) & 7;
            if (s != 7) codon += sym[s];
        }
        return codon;
    }
    /// The read ranges of the ranks of a sharded MSA: rank r holds reads [ShareBounds()[r], ShareBounds()[r + 1]).
    const std::vector<uint64_t>& ShareBounds() const { return shareBounds_; }

private:
    using PackFn = std::function<int(uint32_t packFlags, size_t first, size_t count, const msq_qv_thresholds& thr, msq_packed_dev** dev,
                                     std::exception_ptr* thrown)>;
    static void FreeDev(msq_packed_dev* d)
    {
        if (d) msq_packed_dev_free(Msq::Context(), d);
    }
    // One status for all ranks of a sharded MSA (msq_comm_agree_status): a rank that alone sees an error must not throw
    // before the others know, or they wait for it in the next collective; every rank gets the code and text of the
    // lowest failing rank -- the error a one-process run reports first -- and throws the same exception.
    int Agree(int rc, const char* text = nullptr) const
    {
        if (!sharded_) return rc;
        int agreed = rc;
        Msq::Check(msq_comm_agree_status(Msq::Context(), rc, text, &agreed));
        return agreed;
    }
    // This rank's share of a sample of numReads_ reads: contiguous read ranges balanced by BASE count
    // (msq_comm_shard_bases) -- reads of an amplicon all span the window and the ranges are the equal read counts
    // of msq_comm_shard; staggered or short reads would leave equal read counts up to 2x out of balance.  The
    // insertion histogram is per read, not additive per column: an MSA that collects it keeps every read on every rank.
    template <typename SizeOf>
    void TakeShare(InsertionMode insertions, SizeOf sizeOf)
    {
        shareFirst_ = 0;
        shareCount_ = numReads_;
        sharded_ = Msq::CommActive() && insertions == InsertionMode::None;
        shareBounds_ = {0, (uint64_t)numReads_};
        if (sharded_) {
            const int world = Msq::CommWorld();
            std::vector<uint64_t> prefix(numReads_ + 1, 0);
            for (size_t k = 0; k < numReads_; ++k) prefix[k + 1] = prefix[k] + sizeOf(k);
            shareBounds_.assign((size_t)world + 1, (uint64_t)numReads_);
            for (int r = 0; r < world; ++r) {
                uint64_t first = 0, count = 0;
                Msq::Check(msq_comm_shard_bases(prefix.data(), numReads_, r, world, &first, &count));
                shareBounds_[(size_t)r] = first;
            }
            shareFirst_ = (size_t)shareBounds_[(size_t)Msq::CommRank()];
            shareCount_ = (size_t)(shareBounds_[(size_t)Msq::CommRank() + 1] - shareBounds_[(size_t)Msq::CommRank()]);
        }
    }
    // keys of all numReads_ reads into dKeys [numReads_][K]: this rank gathers its own reads' keys; on a sharded sample
    // the shares (padded to the largest) are all-gathered and laid out in read order
    void GatherKeys(const std::vector<int32_t>& pos, uint16_t* dKeys) const
    {
        msq_ctx* ctx = Msq::Context();
        const size_t K = pos.size();
        const bool sharded = sharded_;
        const int world = sharded ? Msq::CommWorld() : 1;
        size_t most = numReads_;  // the largest share
        if (sharded) {
            most = 0;
            for (int r = 0; r < world; ++r) most = std::max(most, (size_t)(shareBounds_[(size_t)r + 1] - shareBounds_[(size_t)r]));
        }
        Msq::DeviceBuffer<uint16_t> dMine(sharded ? most * K : 0), dAll(sharded ? most * K * world : 0);
        uint16_t* const mine = sharded ? dMine.get() : dKeys;
        int rc = msq_phase_codons(ctx, msq_packed_dev_view(dev_.get()), &thr_, beginPos_ - 1, width_, (int32_t)K, pos.data(), mine);
        if (rc == MSQ_OK) rc = msq_pileup_check(ctx);
        Msq::Check(Agree(rc));  // before the collective: every rank gathers, or every rank throws
        if (!sharded) return;
        Msq::Check(msq_comm_allgather(ctx, dMine.get(), dAll.get(), most * K * sizeof(uint16_t)));
        for (int r = 0; r < world; ++r) {
            const uint64_t first = shareBounds_[(size_t)r], count = shareBounds_[(size_t)r + 1] - first;
            Msq::Check(msq_memcpy_d2d(ctx, dKeys + first * K, dAll.get() + (size_t)r * most * K, count * K * sizeof(uint16_t)));
        }
        Msq::Check(msq_ctx_sync(ctx));
    }
    // The insertion histogram: count the recorded insertions per column on the GPU, pick the columns
    // that matter, fetch their (column, string) records and add them up.
    // basesOf_(read, first, len): the literal nucleotide chars of ArrayBases [first, first + len) of a read -- asked
    // for the few runs the packed 2-bit form cannot hold (longer than 64 bases, or with a non-ACGT base).
    /// visit(record, string) for every insertion a read records at a column the filter keeps (all columns: NULL)
    template <typename Visit>
    void ForEachInsertion(msq_ctx* ctx, const msq_packed_reads* view, int32_t b0, const std::vector<uint32_t>* filter, uint64_t capacity,
                          Visit visit, size_t setFirstRead = SIZE_MAX) const
    {
        if (capacity == 0) return;
        Msq::DeviceBuffer<uint32_t> dFilter(filter ? filter->size() : 1);
        if (filter) Msq::Check(msq_memcpy_h2d(ctx, dFilter.get(), filter->data(), filter->size() * sizeof(uint32_t)));
        Msq::DeviceBuffer<msq_insertion> dRecords(capacity);
        uint64_t n = 0;
        Msq::Check(msq_insertion_records(ctx, view, b0, width_, filter ? dFilter.get() : nullptr, dRecords.get(), capacity, &n));
        const std::vector<msq_insertion> records = dRecords.ToHost();
        static const char nt[] = "ACGT";
        std::string s;
        for (uint64_t k = 0; k < n; ++k) {
            const msq_insertion& r = records[k];
            if (r.len > 64 || (view->flags & MSQ_READS_INS_EXOTIC)) {
                // any length, literal chars (src/MSA.cpp:151-156,172-174); a record counts reads from the set's first read
                s = basesOf_((setFirstRead == SIZE_MAX ? shareFirst_ : setFirstRead) + r.read, r.first_base, r.len);
            } else {
                s.assign(r.len, 'A');
                for (uint32_t i = 0; i < r.len; ++i) s[i] = nt[((i < 32 ? r.seq_lo >> (2 * i) : r.seq_hi >> (2 * (i - 32)))) & 3];
            }
            visit(r, s);
        }
    }
    void CollectInsertions(msq_ctx* ctx, const msq_packed_reads* view, int32_t b0, InsertionMode mode)
    {
        insertions_.assign((size_t)width_, {});
        if (width_ == 0) return;
        Msq::DeviceBuffer<uint32_t> dTotal((size_t)width_);
        Msq::Check(msq_insertion_counts(ctx, view, b0, width_, dTotal.get()));
        const std::vector<uint32_t> total = dTotal.ToHost();
        std::vector<uint32_t> filter(((size_t)width_ + 31) / 32, 0);
        uint64_t capacity = 0;
        for (int c = 0; c < width_; ++c) {
            if (total[c] == 0) continue;
            if (mode == InsertionMode::Majority) {  // a string needs count > Coverage() * 0.5 (src/tools/Fuse.cpp:107-110)
                uint32_t coverage = 0;
                for (int t = 0; t < 6; ++t) coverage += nuc_[(size_t)c * 6 + t];
                if (!(total[c] > coverage * 0.5)) continue;
            }
            filter[c >> 5] |= 1u << (c & 31);
            capacity += total[c];
        }
        ForEachInsertion(ctx, view, b0, &filter, capacity, [this](const msq_insertion& r, const std::string& s) { ++insertions_[(size_t)r.col][s]; });
    }

    // Rows() / NameToRow(): see Rows()
    void Materialise() const
    {
        if (materialised_) return;
        msq_ctx* ctx = Msq::Context();
        const size_t n = numReads_, W = (size_t)width_;
        std::vector<std::shared_ptr<MSARow>> rows;
        rows.reserve(n);
        for (size_t r = 0; r < n; ++r) rows.push_back(std::make_shared<MSARow>(width_));
        if (n != 0 && W != 0) {
            const int32_t b0 = beginPos_ - 1;
            // row.Insertions: the scan needs the genomic order of inserted and column-consuming bases inside a group --
            // the plain layout, which an MSA that collects insertions holds already; otherwise ALL reads are packed once
            // more for this purpose (on one rank's share of a sharded sample too: every rank then decodes every row
            // locally, no collective)
            std::shared_ptr<msq_packed_dev> plain = dev_;
            const msq_packed_reads* view = msq_packed_dev_view(dev_.get());
            if (sharded_ || !(view->flags & MSQ_READS_PAD_MARKED) || (view->flags & MSQ_READS_NORMALIZED)) {
                msq_packed_dev* again = nullptr;
                std::exception_ptr thrown;
                const int rc = pack_(MSQ_PACK_PLAIN, 0, n, thr_, &again, &thrown);
                plain.reset(again, &FreeDev);
                if (thrown) std::rethrow_exception(thrown);
                Msq::Check(rc);
            }
            const msq_packed_reads* pv = msq_packed_dev_view(plain.get());
            const msq_packed_reads* rowsView = sharded_ ? pv : view;  // the resident set when it holds every read
            const size_t per = std::max<size_t>(1, std::min(n, (size_t(64) << 20) / W));  // rows per 64 MB round trip
            Msq::DeviceBuffer<char> dRows(per * W);
            std::vector<char> h(per * W);
            for (size_t first = 0; first < n; first += per) {
                const size_t count = std::min(per, n - first);
                Msq::Check(msq_rows_decode(ctx, rowsView, &thr_, b0, width_, first, count, dRows.get()));
                Msq::Check(msq_memcpy_d2h(ctx, h.data(), dRows.get(), count * W));
                for (size_t k = 0; k < count; ++k) rows[first + k]->Bases.assign(h.begin() + k * W, h.begin() + (k + 1) * W);
            }
            Msq::Check(msq_pileup_check(ctx));
            Msq::DeviceBuffer<uint32_t> dTotal(W);
            Msq::Check(msq_insertion_counts(ctx, pv, b0, width_, dTotal.get()));
            uint64_t capacity = 0;
            for (const uint32_t t : dTotal.ToHost()) capacity += t;
            ForEachInsertion(ctx, pv, b0, nullptr, capacity, [&rows](const msq_insertion& r, const std::string& s) { rows[r.read]->Insertions[(int)r.col] = s; },
                             /* the plain set starts at read */ sharded_ ? 0 : shareFirst_);
        }
        for (size_t r = 0; r < n; ++r) {
            if (r < kept_.size()) rows[r]->Read = kept_[r];
            nameToRow_[ReadNames()[r]] = rows[r];  // a later read of the same name takes the entry (src/MSA.cpp:103)
        }
        rows_ = std::move(rows);
        materialised_ = true;
    }

    // packs reads [first, first + count) of a ReadSource; *thrown = the first exception a fill callback raised
    static int PackSource(const ReadSource& source, uint32_t packFlags, size_t first, size_t count, const msq_qv_thresholds& thr,
                          msq_packed_dev** dev, std::exception_ptr* thrown)
    {
        if (source.packDevice) {  // decoded where they are
            *thrown = nullptr;
            return source.packDevice(packFlags, first, count, thr, dev);
        }
        struct Relay  // the callback crosses a C boundary: keep the first exception and hand it back afterwards
        {
            const ReadSource* source;
            uint64_t first;  // this rank's first read
            msq_qv_thresholds thr;
            std::mutex mutex;
            std::exception_ptr error;
            static void Fill(void* user, uint64_t read, msq_array_base* dst)
            {
                Relay* self = static_cast<Relay*>(user);
                try {
                    self->source->fill(self->first + read, reinterpret_cast<ArrayBase*>(dst));
                } catch (...) {
                    std::lock_guard<std::mutex> lock(self->mutex);
                    if (!self->error) self->error = std::current_exception();
                }
            }
            static int64_t FillCompact(void* user, uint64_t read, uint8_t* dst, uint32_t* presence)
            {
                Relay* self = static_cast<Relay*>(user);
                try {
                    return self->source->fillCompact(self->first + read, dst, self->thr, presence);
                } catch (...) {
                    std::lock_guard<std::mutex> lock(self->mutex);
                    if (!self->error) self->error = std::current_exception();
                }
                return -1;
            }
        } relay{&source, first, thr, {}, nullptr};
        // the thresholds are fixed for the life of the MSA (MSA.h:82): they are applied while packing -- failing match
        // bases become the 'N' AddRead writes (src/MSA.cpp:165-166) and no QV plane ever reaches HBM
        const char* wide = std::getenv("MSQ_PACK_WIDE_STAGING");  // 1: the 8-byte records even from a producer that can translate
        const bool compact = source.fillCompact && !(wide && wide[0] == '1');
        const int rc = compact ? msq_pack_reads_device_fill_compact(Msq::Context(), count, source.referenceStart.data() + first,
                                                                    source.referenceEnd.data() + first, source.numBases.data() + first,
                                                                    &Relay::FillCompact, &relay, packFlags, 0, 0, &thr, dev)
                               : msq_pack_reads_device_fill_baked(Msq::Context(), count, source.referenceStart.data() + first,
                                                                  source.referenceEnd.data() + first, source.numBases.data() + first,
                                                                  &Relay::Fill, &relay, packFlags, 0, 0, &thr, dev);
        *thrown = relay.error;
        return rc;
    }
    // this rank's share packed into HBM; on a sharded MSA the ranks agree on the outcome before any of them throws
    msq_packed_dev* PackShare(uint32_t packFlags)
    {
        msq_packed_dev* dev = nullptr;
        std::exception_ptr thrown;
        int rc = pack_(packFlags, shareFirst_, shareCount_, thr_, &dev, &thrown);
        std::string text;
        if (thrown) {
            rc = MSQ_ERR_ARG;
            try {
                std::rethrow_exception(thrown);
            } catch (const std::exception& e) {
                text = e.what();
            } catch (...) {
                text = "exception in a read source";
            }
        }
        const int agreed = Agree(rc, thrown ? text.c_str() : nullptr);
        if ((agreed != MSQ_OK || thrown) && dev) {
            FreeDev(dev);
            dev = nullptr;
        }
        if (thrown && !sharded_) std::rethrow_exception(thrown);  // one process: the producer's own exception
        Msq::Check(agreed);  // "Unexpected cigar ..." -> std::runtime_error, the same on every rank
        return dev;
    }

    void Build(std::vector<const ArrayRead*> readPointers, InsertionMode insertionMode)
    {
        const QvThresholds qv;  // member qvThresholds_ of the reference: read from the environment per MSA (MSA.h:82)
        thr_ = msq_qv_thresholds{qv.QualQV, qv.DelQV, qv.SubQV, qv.InsQV};
        const auto reads = std::make_shared<const std::vector<const ArrayRead*>>(std::move(readPointers));
        numReads_ = reads->size();
        for (const auto* r : *reads) names_.push_back(r->Name());
        TakeShare(insertionMode, [&reads](size_t k) { return (uint64_t)(*reads)[k]->Bases().size(); });
        // the reads' ArrayBase arrays go to the GPU as they are (8-byte records); msq_pack_reads_device copies them
        // through pinned staging and translates / packs them on the device
        static_assert(sizeof(ArrayBase) == sizeof(msq_array_base) && offsetof(ArrayBase, Cigar) == offsetof(msq_array_base, cigar) &&
                          offsetof(ArrayBase, Nucleotide) == offsetof(msq_array_base, nucleotide) &&
                          offsetof(ArrayBase, HasQualQV) == offsetof(msq_array_base, has_qual) &&
                          offsetof(ArrayBase, HasRichQVs) == offsetof(msq_array_base, has_rich) &&
                          offsetof(ArrayBase, QualQV) == offsetof(msq_array_base, qual) && offsetof(ArrayBase, DelQV) == offsetof(msq_array_base, del) &&
                          offsetof(ArrayBase, SubQV) == offsetof(msq_array_base, sub) && offsetof(ArrayBase, InsQV) == offsetof(msq_array_base, ins),
                      "ArrayBase is the ABI's msq_array_base");
        pack_ = [reads](uint32_t packFlags, size_t first, size_t count, const msq_qv_thresholds& thr, msq_packed_dev** dev, std::exception_ptr*) {
            std::vector<int32_t> rs, re;
            std::vector<const msq_array_base*> ptr;
            std::vector<int64_t> len;
            rs.reserve(count);
            re.reserve(count);
            ptr.reserve(count);
            len.reserve(count);
            for (size_t k = first; k < first + count; ++k) {
                const ArrayRead* r = (*reads)[k];
                rs.push_back(r->ReferenceStart());
                re.push_back(r->ReferenceEnd());
                ptr.push_back(reinterpret_cast<const msq_array_base*>(r->Bases().data()));
                len.push_back(static_cast<int64_t>(r->Bases().size()));
            }
            // the thresholds are applied while packing, as above
            return msq_pack_reads_device_baked(Msq::Context(), count, rs.data(), re.data(), ptr.data(), len.data(), packFlags, 0, 0, &thr, dev);
        };
        basesOf_ = [reads](uint64_t read, uint32_t first, uint32_t len) {
            std::string s(len, 'A');
            for (uint32_t i = 0; i < len; ++i) s[i] = (*reads)[read]->Bases()[first + i].Nucleotide;
            return s;
        };
        // the insertion scan needs the genomic order inside each group (plain layout); the pileup takes both
        Finish(PackShare(insertionMode == InsertionMode::None ? 0u : MSQ_PACK_PLAIN), reads->empty(), insertionMode);
    }

    // window bounds, pileup and (on request) the insertion histogram of the packed reads now resident in HBM
    void Finish(msq_packed_dev* dev, bool noReads, InsertionMode insertionMode)
    {
        struct EmptyTag { bool empty() const { return value; } bool value; } reads{noReads};
        msq_ctx* ctx = Msq::Context();
        dev_.reset(dev, &FreeDev);  // kept for PhaseKeys and Rows()
        {
            // UpdateBoundaries over this rank's reads, left in HBM as {-begin, end}; one MAX over the ranks makes it the
            // sample's window (a no-op without a communicator); then it comes to the host, which sizes the matrices
            Msq::DeviceBuffer<int32_t> dBounds(2);
            Msq::Check(msq_window_bounds_device(ctx, msq_packed_dev_view(dev), dBounds.get()));
            if (sharded_) Msq::Check(msq_comm_allreduce_bounds(ctx, dBounds.get()));
            const std::vector<int32_t> bounds = dBounds.ToHost();
            const int32_t b0 = -bounds[0], e0 = bounds[1];
            width_ = reads.empty() ? 0 : e0 - b0;
            beginPos_ = b0;
            endPos_ = e0;
            // [W][6] nucleotide counts, then [W][64] codon counts, in one buffer: one u32 SUM over the ranks
            dCounts_.reset(new Msq::DeviceBuffer<uint32_t>((size_t)width_ * 70));
            uint32_t* const dNuc = dCounts_->get();
            uint32_t* const dCodon = dNuc + (size_t)width_ * 6;
            int rc = msq_pileup(ctx, msq_packed_dev_view(dev), &thr_, b0, width_, dNuc, dCodon, MSQ_PILEUP_AUTO);
            if (rc == MSQ_OK) rc = msq_pileup_check(ctx);  // "Unexpected base ..." -> std::runtime_error
            Msq::Check(Agree(rc));                        // ... on every rank, before the collective
            if (sharded_) Msq::Check(msq_comm_allreduce_counts(ctx, dNuc, (size_t)width_ * 70));
            const std::vector<uint32_t> counts = dCounts_->ToHost();
            nuc_.assign(counts.begin(), counts.begin() + (size_t)width_ * 6);
            codon_.assign(counts.begin() + (size_t)width_ * 6, counts.end());
            if (insertionMode != InsertionMode::None) CollectInsertions(ctx, msq_packed_dev_view(dev), b0, insertionMode);
        }
        if (!reads.empty()) {  // ++beginPos_, ++endPos_ (src/MSA.cpp:107-108)
            ++beginPos_;
            ++endPos_;
        } else {
            beginPos_ = std::numeric_limits<int>::max();  // MSA.h:83-84 defaults, then the same ++ (wraps in the reference: UB)
            endPos_ = 1;
        }
    }

    int beginPos_ = std::numeric_limits<int>::max();
    int endPos_ = 0;
    int width_ = 0;
    size_t numReads_ = 0;
    std::vector<uint32_t> nuc_, codon_;
    std::shared_ptr<Msq::DeviceBuffer<uint32_t>> dCounts_;
    size_t shareFirst_ = 0, shareCount_ = 0;  // the reads this rank packed and piled up (all of them on one GPU)
    std::vector<uint64_t> shareBounds_{0, 0};  // [world + 1] first read of every rank's share
    bool sharded_ = false;                    // a communicator is bound and this MSA took its share
    std::shared_ptr<msq_packed_dev> dev_;  // the packed reads stay in HBM for the phasing pass
    msq_qv_thresholds thr_{-1, -1, -1, -1};
    std::vector<std::string> names_;
    const std::vector<std::string>* sourceNames_ = nullptr;  // an MSA over a ReadSource reads the source's names
    std::vector<std::map<std::string, int>> insertions_;
    // for Rows(): how to pack the reads again, their literal bases, the reads the rows share, and the rows themselves
    PackFn pack_;
    std::function<std::string(uint64_t read, uint32_t first, uint32_t len)> basesOf_;
    std::vector<std::shared_ptr<Data::ArrayRead>> kept_;
    mutable std::vector<std::shared_ptr<MSARow>> rows_;
    mutable std::map<std::string, std::shared_ptr<MSARow>> nameToRow_;
    mutable bool materialised_ = false;
};

/// Per-column results of a nucleotide-level Fisher test (include/pacbio/data/FisherResult.h:42-48; the reference
/// declares it and juliet never fills one: "NOT YET USED", MSA.h:184-188).
struct FisherResult
{
    std::array<double, 6> PValues = {{0, 0, 0, 0, 0, 0}};
    std::array<double, 6> Mask = {{0, 0, 0, 0, 0, 0}};
    bool Hit = false;
    int ArgMax = 0;
};

/// A single MSA column: counts per nucleotide {A, C, G, T, -, N}, insertions, and the Fisher results a caller
/// may attach.
class MSAColumn  // include/pacbio/data/MSA.h:155-214
{
public:
    MSAColumn(int refPos) : refPos_(refPos) {}
    static int Tag(char c)  // NucleotideToTag, NucleotideConversion.h:69-88
    {
        switch (c) {
            case 'A': return 0;
            case 'C': return 1;
            case 'G': return 2;
            case 'T': return 3;
            case '-': return 4;
            case 'N': return 5;
            default: return 255;
        }
    }
    int operator[](char c) const { return counts_.at(Tag(c)); }
    double Frequency(char c) const { return counts_.at(Tag(c)) / static_cast<double>(Coverage()); }
    explicit operator int() { return Coverage(); }
    int Coverage() const { return std::accumulate(counts_.cbegin(), counts_.cend(), 0); }  // src/MSA.cpp:221
    char MaxBase() const  // MSA.inl:68-80: first maximum; an 'N' majority prints as ' '
    {
        static const char bases[]{'A', 'C', 'G', 'T', '-'};
        const int m = static_cast<int>(std::distance(counts_.begin(), std::max_element(counts_.begin(), counts_.end())));
        return m == 5 ? ' ' : bases[m];
    }
    int RefPos() const { return refPos_; }
    /// Insertions and respective counts (MSA.h:179-180, MSA.inl:101-103); filled when the MSAByRow collected them.
    const std::map<std::string, int>& Insertions() const { return insertions_; }
    void IncInsertion(const std::string& seq) { insertions_[seq]++; }
    /// P-value for given nucleotide: 1 until a FisherResult is attached (MSA.h:197, MSA.inl:105).
    double PValue(const char c) const { return pValues_.at(Tag(c)); }
    /// Insertions whose attached p-value is below 0.01, in string order (src/MSA.cpp:223-229).
    std::vector<std::string> SignificantInsertions() const
    {
        std::vector<std::string> hits;
        for (const auto& entry : insertionsPValues_)
            if (entry.second < 0.01) hits.push_back(entry.first);
        return hits;
    }
    void AddFisherResult(const FisherResult& f)  // src/MSA.cpp:231-237
    {
        pValues_ = f.PValues;
        mask_ = f.Mask;
        hit_ = f.Hit;
        argMax_ = f.ArgMax;
    }
    void AddFisherResult(const std::map<std::string, double>& f) { insertionsPValues_ = f; }  // MSA.inl:82-85
    /// "refPos<TAB>(-,count,p)<TAB>(ins,count,p)<TAB>...": the masked deletion and the significant insertions (MSA.inl:87-97)
    std::ostream& InDels(std::ostream& stream)
    {
        stream << refPos_ << "\t";
        if (mask_.at(4) == 1) stream << "(-," << counts_.at(4) << "," << pValues_.at(4) << ")\t";
        for (const std::string& ins : SignificantInsertions())
            stream << "(" << ins << "," << insertions_.at(ins) << "," << insertionsPValues_.at(ins) << ")\t";
        stream << std::endl;
        return stream;
    }
    /// the five counts A C G T -, each followed by a tab (MSA.inl:107-112)
    friend std::ostream& operator<<(std::ostream& stream, const MSAColumn& r)
    {
        for (const char b : {'A', 'C', 'G', 'T', '-'}) stream << r[b] << "\t";
        return stream;
    }

private:
    std::array<int, 6> counts_{{0, 0, 0, 0, 0, 0}};
    std::map<std::string, int> insertions_;
    std::map<std::string, double> insertionsPValues_;
    std::array<double, 6> pValues_{{1, 1, 1, 1, 1, 1}};
    std::array<double, 6> mask_{{0, 0, 0, 0, 0, 0}};
    bool hit_ = false;
    int argMax_ = 0;
    int refPos_ = -1;
    friend class MSAByColumn;
};

/// MSA by columns; index parameters are in ABSOLUTE 0-based reference space.
class MSAByColumn  // include/pacbio/data/MSA.h:115-150, src/MSA.cpp:55-92
{
    using MsaVec = std::vector<MSAColumn>;

public:
    MSAByColumn(const MSAByRow& msaRows)
    {
        beginPos_ = msaRows.BeginPos() - 1;
        endPos_ = msaRows.EndPos() - 1;
        const int size = msaRows.Width();
        int pos = msaRows.BeginPos();
        counts.reserve(size);
        const auto& nuc = msaRows.NucleotideCounts();
        for (int i = 0; i < size; ++i) {
            counts.emplace_back(pos++);
            for (int t = 0; t < 6; ++t) counts.back().counts_[t] = static_cast<int>(nuc[(size_t)i * 6 + t]);
            if (!msaRows.Insertions().empty()) counts.back().insertions_ = msaRows.Insertions()[(size_t)i];  // MSA.cpp:88-90
        }
    }
    MSAColumn& operator[](int i) const { return const_cast<MSAColumn&>(counts[i - beginPos_]); }
    bool has(int i) { return i >= beginPos_ && i < endPos_; }
    MsaVec::iterator begin() { return counts.begin(); }
    MsaVec::iterator end() { return counts.end(); }
    MsaVec::const_iterator begin() const { return counts.begin(); }
    MsaVec::const_iterator end() const { return counts.end(); }
    MsaVec::const_iterator cbegin() const { return counts.cbegin(); }
    MsaVec::const_iterator cend() const { return counts.cend(); }
    int BeginPos() const { return beginPos_; }
    int EndPos() const { return endPos_; }

private:
    MsaVec counts;
    int beginPos_ = std::numeric_limits<int>::max();
    int endPos_ = 0;
};
}  // namespace Data

namespace Statistics {
class Fisher  // include/pacbio/statistics/Fisher.h:12-25
{
public:
    static double fisher_exact_tiss(int chi11, int chi12, int chi21, int chi22)
    {
        double p = 0;
        Msq::Check(msq_fisher(Msq::Context(), chi11, chi12, chi21, chi22, &p));
        return p;
    }
};
}  // namespace Statistics

namespace Juliet {
struct ErrorEstimates  // include/pacbio/juliet/ErrorEstimates.h:49-61, src/ErrorEstimates.cpp:44-67
{
    ErrorEstimates() = default;
    ErrorEstimates(const std::string& /*chemistry*/)
    {
        msq_error_model m;
        msq_error_model_default(&m);
        Match = m.match; Substitution = m.substitution; Deletion = m.deletion; Insertion = 0;
    }
    ErrorEstimates(const double substitutionRate, const double deletionRate)
    {
        msq_error_model m;
        msq_error_model_from_rates(substitutionRate, deletionRate, &m);
        Match = m.match; Substitution = m.substitution; Deletion = m.deletion; Insertion = 0;
    }
    double Match = -1, Substitution = -1, Deletion = -1, Insertion = -1;
};

class DMutation  // include/pacbio/juliet/TargetConfig.h:49-62, src/TargetConfig.cpp:51-68
{
public:
    DMutation(char refAA, int pos, char curAA) : refAA(refAA), pos(pos), curAA(curAA) {}
    /// '*' on either side stands for any amino acid
    bool operator==(const DMutation& a) const
    {
        const auto agree = [](char x, char y) { return x == '*' || y == '*' || x == y; };
        return a.pos == pos && agree(a.curAA, curAA) && agree(a.refAA, refAA);
    }
    char refAA;
    int pos;
    char curAA;
};
struct DRM { std::string name; std::vector<DMutation> positions; };
struct ExpectedMinor { int position; std::string aminoacid; std::string codon; };
struct TargetGene  // TargetConfig.h:95-113; begin/end are 1-based reference coordinates
{
    TargetGene() = default;
    TargetGene(const int begin, const int end, const std::string& name, const std::vector<DRM>& drms,
               const std::vector<ExpectedMinor>& minors = std::vector<ExpectedMinor>())
        : begin(begin), end(end), name(name), drms(drms), minors(minors)
    {
    }
    int begin = 0, end = 0;
    std::string name;
    std::vector<DRM> drms;
    std::vector<ExpectedMinor> minors;
};
struct TargetConfig  // TargetConfig.h:117-139, src/TargetConfig.cpp:120-265
{
    TargetConfig() = default;
    /// From a JSON string ("{...}") or the path of a JSON file, optionally wrapped in <>, "" or '' as the CLI passes
    /// it (DetermineConfigInput, src/TargetConfig.cpp:134-167).  The reference also knows two built-in tags ("HIV",
    /// "HIV-PB": HXB2 with the HIVdb resistance lists); those data are not part of this header -- pass the
    /// configuration as a file or string -- and a tag is reported like any unknown input.
    explicit TargetConfig(const std::string& input)
    {
        const std::string inputString = DetermineConfigInput(input);
        if (!inputString.empty()) {
            const JSON::Json inputJson = JSON::Json::parse(inputString);
            targetGenes = TargetGenesFromJson(inputJson);
            referenceSequence = RootTagFromJson(inputJson, "referenceSequence");
            referenceName = RootTagFromJson(inputJson, "referenceName");
            version = RootTagFromJson(inputJson, "version");
            dbVersion = RootTagFromJson(inputJson, "databaseVersion");
        }
    }
    static std::string DetermineConfigInput(std::string input)
    {
        if (input.empty()) return "";
        if ((input.front() == '<' && input.back() == '>') || (input.front() == '"' && input.back() == '"') ||
            (input.front() == '\'' && input.back() == '\''))
            input = input.substr(1, input.size() - 2);
        std::string output;
        if (!input.empty() && input.front() == '{') {
            output = input;
        } else {
            std::ifstream t(input);
            if (t.good())
                output.assign(std::istreambuf_iterator<char>(t), std::istreambuf_iterator<char>());
            else
                std::cerr << "Warning: Target config -c is not a file, JSON string, or tag: " << input << std::endl;
        }
        if (output.empty())
            std::cerr << "Warning: Empty target config -c, specified region will be treated as coding region" << std::endl;
        return output;
    }
    static std::string RootTagFromJson(const JSON::Json& root, const std::string& tag)
    {
        if (root.empty() || root.find(tag) == root.cend() || root.at(tag).empty()) return "";
        return root.at(tag).AsString();
    }
    /// genes[]: name, begin, end (required); drms[]: name, positions[] -- "M184VI" expands to every (reference amino
    /// acid, position, variant amino acid) combination, a missing side to '*'; expectedminors[]: position, aminoacid,
    /// codon.  The error texts are the reference's (src/TargetConfig.cpp:186-262).
    static std::vector<TargetGene> TargetGenesFromJson(const JSON::Json& root)
    {
        std::vector<TargetGene> genes;
        if (root.empty() || root.find("genes") == root.cend() || root.at("genes").empty()) return genes;
        auto has = [](const JSON::Json& j, const char* key) { return j.find(key) != j.cend(); };
        for (const JSON::Json& jGene : root.at("genes").array()) {
            if (!has(jGene, "name")) throw std::runtime_error("Missing name in gene.");
            if (!has(jGene, "begin")) throw std::runtime_error("Missing begin in gene.");
            if (!has(jGene, "end")) throw std::runtime_error("Missing end in gene.");
            const std::string name = jGene.at("name").AsString();
            std::vector<DRM> drms;
            if (has(jGene, "drms")) {
                for (const JSON::Json& jDrm : jGene.at("drms").array()) {
                    DRM drm;
                    if (!has(jDrm, "name")) throw std::runtime_error("Missing name in drm in gene " + name);
                    drm.name = jDrm.at("name").AsString();
                    if (!has(jDrm, "positions")) throw std::runtime_error("Missing positions in drm in gene " + name);
                    if (jDrm.at("positions").empty()) throw std::runtime_error("Empty positions in drm in gene " + name);
                    for (const JSON::Json& jp : jDrm.at("positions").array()) {
                        const std::string& p = jp.AsString();
                        // the first match of ([a-zA-Z]*)([0-9]+)([a-zA-Z]*): letters, digits, letters
                        size_t d0 = 0;
                        while (d0 < p.size() && !(p[d0] >= '0' && p[d0] <= '9')) ++d0;
                        if (d0 == p.size()) continue;  // no digits: regex_search finds nothing
                        size_t l0 = d0;
                        while (l0 > 0 && std::isalpha(static_cast<unsigned char>(p[l0 - 1]))) --l0;
                        size_t d1 = d0;
                        while (d1 < p.size() && p[d1] >= '0' && p[d1] <= '9') ++d1;
                        size_t l1 = d1;
                        while (l1 < p.size() && std::isalpha(static_cast<unsigned char>(p[l1]))) ++l1;
                        std::string refAAs = p.substr(l0, d0 - l0), curAAs = p.substr(d1, l1 - d1);
                        if (refAAs.empty()) refAAs = "*";
                        if (curAAs.empty()) curAAs = "*";
                        const int pos = std::stoi(p.substr(d0, d1 - d0));
                        for (const char a : refAAs)
                            for (const char b : curAAs) drm.positions.emplace_back(a, pos, b);
                    }
                    drms.emplace_back(std::move(drm));
                }
            }
            TargetGene g(static_cast<int>(jGene.at("begin").AsInt()), static_cast<int>(jGene.at("end").AsInt()), name, drms);
            if (has(jGene, "expectedminors")) {
                for (const JSON::Json& jMinor : jGene.at("expectedminors").array()) {
                    if (!has(jMinor, "position")) throw std::runtime_error("Missing position in expectedminors in gene " + name);
                    if (!has(jMinor, "aminoacid")) throw std::runtime_error("Missing aminoacid in expectedminors in gene " + name);
                    if (!has(jMinor, "codon")) throw std::runtime_error("Missing codon in expectedminors in gene " + name);
                    g.minors.push_back(ExpectedMinor{static_cast<int>(jMinor.at("position").AsInt()), jMinor.at("aminoacid").AsString(),
                                                     jMinor.at("codon").AsString()});
                }
            }
            genes.emplace_back(std::move(g));
        }
        return genes;
    }

    std::vector<TargetGene> targetGenes;
    std::string referenceName, referenceSequence, version, dbVersion;
    size_t NumExpectedMinors() const
    {
        size_t n = 0;
        for (const auto& g : targetGenes) n += g.minors.size();
        return n;
    }
    bool HasExpectedMinors() const { return NumExpectedMinors() > 0; }
};
struct JulietSettings  // the fields AminoAcidCaller reads, include/pacbio/juliet/JulietSettings.h:68-82
{
    TargetConfig TargetConfigUser;
    bool DRMOnly = false, Verbose = false, Debug = false;
    double MinimalPerc = 0.1, MaximalPerc = 100;
};

struct PerformanceMetrics  // include/pacbio/juliet/AminoAcidCaller.h:55-78
{
    PerformanceMetrics(double numberOfTests, double numExpectedMinors) : NumberOfTests(numberOfTests), NumExpectedMinors(numExpectedMinors) {}
    double TruePositives = 0, FalsePositives = 0, FalseNegative = 0, TrueNegative = 0;
    double NumberOfTests, NumExpectedMinors;
    double TruePositiveRate() const { return TruePositives / NumExpectedMinors; }
    double FalsePositiveRate() const { return FalsePositives / (NumberOfTests - NumExpectedMinors); }
    double Accuracy() const { return (TruePositives + TrueNegative) / (TruePositives + FalsePositives + FalseNegative + TrueNegative); }
    /// the one-line validation record (the text of src/AminoAcidCaller.cpp:579-588: ostream formatting, 6 significant digits)
    std::string ToJson() const
    {
        const std::pair<const char*, double> fields[] = {{"true_positive_rate", TruePositiveRate()},
                                                         {"false_positive_rate", FalsePositiveRate()},
                                                         {"num_tests", NumberOfTests},
                                                         {"num_false_positives", FalsePositives},
                                                         {"accuracy", Accuracy()}};
        std::ostringstream text;
        char lead = '{';
        for (const auto& f : fields) {
            text << lead << '"' << f.first << "\":" << f.second;
            lead = ',';
        }
        text << '}';
        return text.str();
    }
    /// "TPR FPR tests accuracy FP" on one line (include/pacbio/juliet/internal/AminoAcidCaller.inl:42-47)
    friend std::ostream& operator<<(std::ostream& out, const PerformanceMetrics& pm)
    {
        const double fields[] = {pm.TruePositiveRate(), pm.FalsePositiveRate(), pm.NumberOfTests, pm.Accuracy(), pm.FalsePositives};
        const char* lead = "";
        for (const double f : fields) {
            out << lead << f;
            lead = " ";
        }
        return out;
    }
};

struct VariantGene  // include/pacbio/juliet/VariantGene.h:49-81
{
    VariantGene(const std::string& name, const int offset) : geneName(name), geneOffset(offset) {}
    std::string geneName;
    int geneOffset;
    struct MsaCounts  // the JSON object of src/AminoAcidCaller.cpp:549-563 as a plain struct
    {
        int rel_pos, abs_pos, A, C, G, T, gap, N;
        std::string wt;
    };
    struct VariantPosition
    {
        std::string refCodon, altRefCodon;
        char refAminoAcid = 0, altRefAminoAcid = 0;
        std::vector<MsaCounts> msa;
        int coverage = 0;
        struct VariantCodon
        {
            std::string codon;
            double frequency, pValue;
            std::string knownDRM;
            std::vector<bool> haplotypeHit;
        };
        std::map<char, std::vector<VariantCodon>> aminoAcidToCodons;
        bool IsVariant() const { return !aminoAcidToCodons.empty(); }  // src/VariantGene.cpp:83
        bool IsHit(const std::string& codon) const                       // src/VariantGene.cpp:84-97
        {
            if (codon == refCodon || codon == altRefCodon) return true;
            for (const auto& kv : aminoAcidToCodons)
                for (const auto& vc : kv.second)
                    if (codon == vc.codon) return true;
            return false;
        }
    };
    std::map<int, std::shared_ptr<VariantPosition>> relPositionToVariant;

    /// The gene's entry of the report.  Key names and nesting are the report format (src/VariantGene.cpp:42-81):
    /// name, variant_positions[] of {ref_position, ref_codon, coverage, ref_amino_acid, variant_amino_acids[] of
    /// {amino_acid, variant_codons[] of {codon, frequency, pValue, known_drm, haplotype_hit}}, msa[]}; positions and
    /// amino acids without a stored codon are left out, and so is variant_positions when nothing is left.
    JSON::Json ToJson() const
    {
        using JSON::Json;
        const auto codonEntry = [](const VariantPosition::VariantCodon& vc) {
            return Msq::Obj({{"codon", vc.codon}, {"frequency", vc.frequency}, {"pValue", vc.pValue}, {"known_drm", vc.knownDRM},
                             {"haplotype_hit", vc.haplotypeHit}});
        };
        const auto columnEntry = [](const MsaCounts& m) {
            return Msq::Obj({{"rel_pos", m.rel_pos}, {"abs_pos", m.abs_pos}, {"A", m.A}, {"C", m.C}, {"G", m.G}, {"T", m.T},
                             {"-", m.gap}, {"N", m.N}, {"wt", m.wt}});
        };
        std::vector<Json> positions;
        for (const auto& entry : relPositionToVariant) {
            const VariantPosition& vp = *entry.second;
            if (!vp.IsVariant()) continue;
            std::vector<Json> aminoAcids, context;
            for (const auto& group : vp.aminoAcidToCodons) {
                if (group.second.empty()) continue;
                std::vector<Json> codons;
                std::transform(group.second.begin(), group.second.end(), std::back_inserter(codons), codonEntry);
                aminoAcids.push_back(Msq::Obj({{"amino_acid", std::string(1, group.first)}, {"variant_codons", codons}}));
            }
            std::transform(vp.msa.begin(), vp.msa.end(), std::back_inserter(context), columnEntry);
            positions.push_back(Msq::Obj({{"ref_position", entry.first}, {"ref_codon", vp.refCodon}, {"coverage", vp.coverage},
                                          {"ref_amino_acid", std::string(1, vp.refAminoAcid)}, {"variant_amino_acids", aminoAcids},
                                          {"msa", context}}));
        }
        Json gene = Msq::Obj({{"name", geneName}});
        if (!positions.empty()) gene["variant_positions"] = positions;
        return gene;
    }
};

enum class HaplotypeType : int  // include/pacbio/juliet/Haplotype.h:46-54
{
    REPORT = 0,
    WITH_GAP = 1,
    WITH_HETERODUPLEX = 2,
    PARTIAL = 4,
    LOW_COV = 8,
    OFFTARGET = 16
};

/// One observed combination of codons over the variant positions and the reads that carry it
/// (include/pacbio/juliet/Haplotype.h:56-126).  The flags a codon's characters raise -- '-' a gap, 'N' a
/// heteroduplex (masked) base, ' ' a read that does not cover the position (src/Haplotype.cpp:50-57) -- come from
/// one table look-up per character.
class Haplotype
{
public:
    Haplotype() = delete;
    Haplotype(std::vector<std::string> readNames, std::vector<std::string> codons, const HaplotypeType& flag)
        : readNames_(std::move(readNames)), codons_(std::move(codons)), flags_(static_cast<int>(flag))
    {
        for (const std::string& codon : codons_)
            for (const char c : codon) flags_ |= CharFlag(c);
    }
    Haplotype(const std::string& readName, const std::vector<std::string>& codons, const HaplotypeType& flag)
        : Haplotype(std::vector<std::string>{readName}, codons, flag)
    {
    }
    double Size() const { return static_cast<double>(readNames_.size()) + softCollapses_; }
    std::string ConcatCodons() const
    {
        std::string all;
        for (const auto& c : codons_) all += c;
        return all;
    }
    const std::vector<std::string>& ReadNames() const { return readNames_; }
    const std::string& Codon(const int i) const { return codons_.at(i); }
    size_t NumCodons() const { return codons_.size(); }
    int Flags() const { return flags_; }
    std::string Name() const { return name_; }
    double Frequency() const { return frequency_; }
    void AddFlag(const HaplotypeType& flag) { flags_ |= static_cast<int>(flag); }
    void Frequency(const double& freq) { frequency_ = freq; }
    void AddReadName(const std::string& name) { readNames_.push_back(name); }
    void AddSoftReadCount(const double s) { softCollapses_ += s; }
    void Name(const std::string& name) { name_ = name; }
    /// the haplotype's entry of the report (key names: src/Haplotype.cpp:59-70)
    JSON::Json ToJson() const
    {
        return Msq::Obj({{"name", name_}, {"reads_hard", readNames_.size()}, {"reads_soft", Size()}, {"frequency", frequency_},
                         {"read_names", readNames_}, {"codons", codons_}});
    }

private:
    static int CharFlag(char c)
    {
        switch (c) {
            case '-': return static_cast<int>(HaplotypeType::WITH_GAP);
            case 'N': return static_cast<int>(HaplotypeType::WITH_HETERODUPLEX);
            case ' ': return static_cast<int>(HaplotypeType::PARTIAL);
            default: return 0;
        }
    }
    std::string name_;
    std::vector<std::string> readNames_;
    std::vector<std::string> codons_;
    double softCollapses_ = 0;
    double frequency_ = 0;
    int flags_ = 0;
};

/// Given a MSA, target config, and noise model, compute variant amino acids.
class AminoAcidCaller  // include/pacbio/juliet/AminoAcidCaller.h:90-154
{
public:
    AminoAcidCaller(const std::vector<std::shared_ptr<Data::ArrayRead>>& reads, const ErrorEstimates& error,
                    const JulietSettings& settings)
        : msaByRow_(reads)
        , msaByColumn_(msaByRow_)
        , error_(error)
        , targetConfig_(settings.TargetConfigUser)
        , debug_(settings.Debug)
        , drmOnly_(settings.DRMOnly)
        , minimalPerc_(settings.MinimalPerc)
        , maximalPerc_(settings.MaximalPerc)
    {
        CallVariants();
    }
    /// The same caller on reads that are decoded straight into the device packer's staging (no ArrayRead objects).
    AminoAcidCaller(const Data::ReadSource& reads, const ErrorEstimates& error, const JulietSettings& settings)
        : msaByRow_(reads)
        , msaByColumn_(msaByRow_)
        , error_(error)
        , targetConfig_(settings.TargetConfigUser)
        , debug_(settings.Debug)
        , drmOnly_(settings.DRMOnly)
        , minimalPerc_(settings.MinimalPerc)
        , maximalPerc_(settings.MaximalPerc)
    {
        CallVariants();
    }
    const std::vector<VariantGene>& VariantGenes() const { return variantGenes_; }
    int NumberOfTests() const { return numberOfTests_; }
    const PerformanceMetrics& Performance() const { return *pm_; }
    const Data::MSAByRow& Rows() const { return msaByRow_; }

    /// Phase variants into haplotypes (src/AminoAcidCaller.cpp:126-323).
    /// 1. The codon of every read at every variant position, and the grouping of reads with identical codons at all
    ///    of them, run on the GPU over the packed reads still resident in HBM (MSAByRow::GroupRowsByCodons): the same
    ///    partition, in the same first-occurrence order, as the reference's pairwise comparison (:149-196).
    /// 2. A group is OFFTARGET when one of its codons is neither a reference codon nor a stored variant codon (:157),
    ///    LOW_COV below ten reads (:204); its codons add the gap / heteroduplex / partial flags.  Unflagged groups are
    ///    the reported haplotypes ("generators"), the others are only counted.
    /// 3. Both lists are ordered with the reference's own two sorts -- ascending std::sort, then, for the reported
    ///    ones, std::stable_sort under the reference's `>=` comparator (:215-230) -- applied to the group sizes in the
    ///    same sequence of calls, so that ties fall exactly as they do there.
    void PhaseVariants()
    {
        struct Site { int begin1; VariantGene::VariantPosition* variant; };
        std::vector<Site> sites;
        for (const auto& gene : variantGenes_)
            for (const auto& entry : gene.relPositionToVariant)
                if (entry.second->IsVariant()) sites.push_back(Site{gene.geneOffset + entry.first * 3, entry.second.get()});
        std::vector<int32_t> columns;
        for (const Site& site : sites) columns.push_back(site.begin1 - msaByRow_.BeginPos() - 3);  // CodonAt's argument, :154
        const Data::MSAByRow::RowGroups groups = msaByRow_.GroupRowsByCodons(columns);
        const size_t K = sites.size(), G = groups.NumGroups();

        // the reads of every group, in read order
        std::vector<std::vector<std::string>> members(G);
        for (size_t g = 0; g < G; ++g) members[g].reserve(groups.size[g]);
        const auto& names = msaByRow_.ReadNames();
        for (size_t r = 0; r < groups.groupOfRow.size(); ++r) members[groups.groupOfRow[r]].push_back(names[r]);

        std::vector<Haplotype> reported, counted;
        for (size_t g = 0; g < G; ++g) {
            std::vector<std::string> codons(K);
            bool onTarget = true;
            for (size_t k = 0; k < K; ++k) {
                codons[k] = Data::MSAByRow::KeyToCodon(groups.keys[g * K + k]);
                onTarget = onTarget && sites[k].variant->IsHit(codons[k]);
            }
            Haplotype h(std::move(members[g]), std::move(codons), onTarget ? HaplotypeType::REPORT : HaplotypeType::OFFTARGET);
            if (h.Size() < 10) h.AddFlag(HaplotypeType::LOW_COV);
            (h.Flags() == 0 ? reported : counted).push_back(std::move(h));
        }
        const auto bySize = [](const Haplotype& a, const Haplotype& b) { return a.Size() < b.Size(); };
        OrderAs(reported, [&](std::vector<size_t>& order) {
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bySize(reported[a], reported[b]); });
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return reported[a].Size() >= reported[b].Size(); });
        });
        OrderAs(counted, [&](std::vector<size_t>& order) {
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bySize(counted[a], counted[b]); });
        });

        double reportedReads = 0;
        for (const Haplotype& h : reported) reportedReads += h.Size();
        const bool twoLetters = reported.size() > 26;  // A .. Z, else Aa, Ab, ... (:234-247)
        for (size_t n = 0; n < reported.size(); ++n) {
            Haplotype& h = reported[n];
            h.Frequency(h.Size() / reportedReads);
            h.Name(twoLetters ? std::string{static_cast<char>('A' + n / 26), static_cast<char>('a' + n % 26)}
                              : std::string(1, static_cast<char>('A' + n)));
            for (size_t k = 0; k < K; ++k)  // one more bit per stored codon: does this haplotype carry it? (:253-263)
                for (auto& group : sites[k].variant->aminoAcidToCodons)
                    for (auto& vc : group.second) vc.haplotypeHit.push_back(h.Codon((int)k) == vc.codon);
            genCounts_ += static_cast<int>(h.ReadNames().size());
        }
        for (const Haplotype& h : counted) {  // the read counts behind "haplotype_read_counts" (:286-310)
            const int reads = static_cast<int>(h.ReadNames().size()), flags = h.Flags();
            const auto has = [flags](HaplotypeType t) { return (flags & static_cast<int>(t)) != 0; };
            if (has(HaplotypeType::WITH_GAP)) margWithGap_ += reads;
            if (has(HaplotypeType::WITH_HETERODUPLEX)) margWithHetero_ += reads;
            if (has(HaplotypeType::PARTIAL)) margPartial_ += reads;
            if (has(HaplotypeType::OFFTARGET)) margOfftarget_ += reads;
            if (flags == static_cast<int>(HaplotypeType::LOW_COV)) lowCov_ += reads;  // low coverage and nothing else
        }
        reconstructedHaplotypes_ = std::move(reported);
        filteredHaplotypes_ = std::move(counted);
    }
    /// The report juliet writes with dump(2); key names are the report format (src/AminoAcidCaller.cpp:590-619).
    JSON::Json JSON() const
    {
        std::vector<JSON::Json> genes, haplotypes;
        for (const VariantGene& gene : variantGenes_) {
            JSON::Json entry = gene.ToJson();
            if (entry.find("variant_positions") != entry.cend()) genes.push_back(std::move(entry));
        }
        std::transform(reconstructedHaplotypes_.begin(), reconstructedHaplotypes_.end(), std::back_inserter(haplotypes),
                       [](const Haplotype& h) { return h.ToJson(); });
        return Msq::Obj({{"genes", genes},
                         {"haplotypes", haplotypes},
                         {"haplotype_read_counts",
                          Msq::Obj({{"healthy_reported", genCounts_}, {"healthy_low_coverage", lowCov_}, {"all_damaged", margOfftarget_},
                                    {"marginal_with_gaps", margWithGap_}, {"marginal_with_heteroduplexes", margWithHetero_},
                                    {"marginal_partial_reads", margPartial_}})}});
    }
    const std::vector<Haplotype>& ReconstructedHaplotypes() const { return reconstructedHaplotypes_; }
    const std::vector<Haplotype>& FilteredHaplotypes() const { return filteredHaplotypes_; }
    /// {healthy_reported, healthy_low_coverage, all_damaged, marginal_with_gaps, marginal_with_heteroduplexes,
    /// marginal_partial_reads}: the counters behind the "haplotype_read_counts" JSON (src/AminoAcidCaller.cpp:610-617)
    std::array<int, 6> HaplotypeReadCounts() const
    {
        return {{genCounts_, lowCov_, margOfftarget_, margWithGap_, margWithHetero_, margPartial_}};
    }

private:
    Data::MSAByRow msaByRow_;  // declared (and so constructed) before msaByColumn_, as in AminoAcidCaller.h:130-134

public:
    Data::MSAByColumn msaByColumn_;  // public in the reference too (AminoAcidCaller.h:134)

private:
    static constexpr float alpha = 0.01;  // AminoAcidCaller.h:108: a float, compared against doubles

    /// Reorder `items` by a permutation that `sorter` establishes on the index vector 0 .. n-1.
    template <typename T, typename Sorter>
    static void OrderAs(std::vector<T>& items, Sorter sorter)
    {
        std::vector<size_t> order(items.size());
        std::iota(order.begin(), order.end(), size_t(0));
        sorter(order);
        std::vector<T> out;
        out.reserve(items.size());
        for (const size_t i : order) out.push_back(std::move(items[i]));
        items = std::move(out);
    }
    /// " + "-joined names of the drugs of the first gene called geneName whose mutation list holds `mutation`
    /// ('*' = any amino acid); the lookup of src/AminoAcidCaller.cpp:105-124.
    static std::string FindDRMs(const std::string& geneName, const std::vector<TargetGene>& genes, const DMutation& mutation)
    {
        const auto gene = std::find_if(genes.begin(), genes.end(), [&](const TargetGene& g) { return g.name == geneName; });
        std::string summary;
        if (gene == genes.end()) return summary;
        for (const DRM& drug : gene->drms) {
            if (std::find(drug.positions.begin(), drug.positions.end(), mutation) == drug.positions.end()) continue;
            if (!summary.empty()) summary += " + ";
            summary += drug.name;
        }
        return summary;
    }
    /// Is (aaPos, codon) one of the gene's expected minors?  Counts the call into the confusion matrix on the way:
    /// a cell counts when the site is variable or the codon is expected; expected + significant is a true positive,
    /// expected + not significant a false negative, and likewise false positive / true negative for the unexpected
    /// (src/AminoAcidCaller.cpp:341-378).
    static bool MeasurePerformance(const TargetGene& gene, int codonIndex, bool variableSite, int aaPos, double p, PerformanceMetrics* pm)
    {
        const std::string codon = Msq::CodonString(codonIndex);
        const char aminoAcid = AminoAcidTable::FromCodonIndex(codonIndex);
        const bool expected = std::any_of(gene.minors.begin(), gene.minors.end(), [&](const ExpectedMinor& m) {
            return m.position == aaPos && m.aminoacid[0] == aminoAcid && m.codon == codon;
        });
        if (variableSite || expected) {
            double* const cell[2][2] = {{&pm->TrueNegative, &pm->FalsePositives}, {&pm->FalseNegative, &pm->TruePositives}};
            ++*cell[expected][p < alpha];
        }
        return expected;
    }

    void CallVariants()  // src/AminoAcidCaller.cpp:397-577: the cell arithmetic runs on the GPU, the store policy here
    {
        auto genes = targetConfig_.targetGenes;
        if (genes.empty()) genes.emplace_back(TargetGene(msaByRow_.BeginPos(), msaByRow_.EndPos(), "Unnamed ORF", {}));
        msq_ctx* ctx = Msq::Context();
        std::vector<int32_t> gb, ge;
        for (const auto& g : genes) {
            gb.push_back(g.begin);
            ge.push_back(g.end);
        }
        const std::string& ref = targetConfig_.referenceSequence;
        msq_plan* plan = nullptr;
        Msq::Check(msq_plan_create(ctx, (int32_t)gb.size(), gb.data(), ge.data(), ref.empty() ? nullptr : ref.data(),
                                   (int64_t)ref.size(), msaByRow_.BeginPos(), msaByRow_.Width(), &plan));
        const int64_t P = msq_plan_positions(plan);
        std::vector<msq_position> positions;
        std::vector<double> pv;
        try {
            Msq::DeviceBuffer<msq_position> dPos((size_t)P);
            Msq::DeviceBuffer<double> dP((size_t)P * 64), dExp((size_t)P * 64);
            Msq::DeviceBuffer<int32_t> dTests(1);
            Msq::Check(msq_lfact_reserve(ctx, 2 * (int64_t)msaByRow_.NumReads() + 2));
            msq_error_model em{error_.Match, error_.Substitution, error_.Deletion};
            Msq::Check(msq_call_cells(ctx, plan, msaByRow_.DeviceCodonCounts(), &em, maximalPerc_, dPos.get(), dP.get(),
                                      dExp.get(), dTests.get()));  // std::out_of_range when a gene leaves the reference
            Msq::Check(msq_call_check(ctx));
            positions = dPos.ToHost();
            pv = dP.ToHost();
            numberOfTests_ = dTests.ToHost()[0];
        } catch (...) {
            msq_plan_destroy(ctx, plan);
            throw;
        }
        msq_plan_destroy(ctx, plan);

        pm_.reset(new PerformanceMetrics(numberOfTests_, targetConfig_.NumExpectedMinors()));
        const bool hasExpectedMinors = pm_->NumExpectedMinors > 0;
        const bool hasReference = !ref.empty();
        const auto& codonCounts = msaByRow_.CodonCounts();
        size_t k = 0;
        for (size_t g = 0; g < genes.size(); ++g) {
            const auto& gene = genes[g];
            VariantGene curVariantGene(gene.name, gene.begin);
            for (; k < positions.size() && positions[k].gene == (int32_t)g; ++k) {
                const msq_position& pos = positions[k];
                const int i = pos.i, absPos = i - 1, aaPos = pos.aa_pos;
                curVariantGene.relPositionToVariant.emplace(aaPos, std::make_shared<VariantGene::VariantPosition>());
                auto& cur = curVariantGene.relPositionToVariant.at(aaPos);
                if (hasReference && absPos >= 0 && (size_t)absPos <= ref.size()) cur->refCodon = ref.substr(absPos, 3);  // :446
                if (pos.ref_codon >= 0) {
                    cur->refCodon = Msq::CodonString(pos.ref_codon);
                    cur->refAminoAcid = AminoAcidTable::FromCodonIndex(pos.ref_codon);
                }
                if (pos.alt_codon >= 0) {
                    cur->altRefCodon = Msq::CodonString(pos.alt_codon);
                    cur->altRefAminoAcid = AminoAcidTable::FromCodonIndex(pos.alt_codon);
                }
                if (pos.skipped) continue;
                const int coverage = pos.coverage;
                for (int c = 0; c < 64; ++c) {
                    const double p = pv[k * 64 + c];
                    if (p < 0) continue;  // no cell: count 0, reference codon, alternative reference codon
                    const std::string codon = Msq::CodonString(c);
                    const int count = (int)codonCounts[(size_t)pos.win_pos * 64 + c];
                    const double frequency = 1.0 * count / coverage;  // :491
                    const bool variableSite = frequency < 0.8;        // :492
                    const bool expectedMinor = MeasurePerformance(gene, c, variableSite, aaPos, p, pm_.get());
                    // Which cells are kept (the policy of src/AminoAcidCaller.cpp:499-539): with --debug everything;
                    // otherwise only significant cells (p < alpha, alpha a float), and of those -- with --drm-only the
                    // known resistance mutations; with expected minors configured the expected ones and everything at
                    // variable sites; without them all.  A kept cell is stored when --debug or its frequency reaches
                    // --min-perc.
                    const char curAA = AminoAcidTable::FromCodonIndex(c);
                    bool keep = debug_;
                    if (!keep && p < alpha) {
                        if (drmOnly_)
                            keep = !FindDRMs(gene.name, genes, DMutation(cur->refAminoAcid, aaPos, curAA)).empty();
                        else
                            keep = expectedMinor || !hasExpectedMinors || variableSite;
                    }
                    if (keep && (debug_ || frequency * 100 >= minimalPerc_)) {
                        VariantGene::VariantPosition::VariantCodon vc;
                        vc.codon = codon;
                        vc.frequency = frequency;
                        vc.pValue = p;
                        vc.knownDRM = FindDRMs(gene.name, genes, DMutation(cur->refAminoAcid, aaPos, curAA));
                        cur->aminoAcidToCodons[curAA].push_back(vc);
                    }
                }
                if (!cur->aminoAcidToCodons.empty()) {  // :544-566
                    cur->coverage = coverage;
                    for (int j = -3; j < 6; ++j) {
                        if (i + j >= msaByRow_.BeginPos() && i + j < msaByRow_.EndPos()) {
                            const int abs = absPos + j;
                            const auto& col = msaByColumn_[abs];
                            VariantGene::MsaCounts mc{j, abs, col['A'], col['C'], col['G'], col['T'], col['-'], col['N'], ""};
                            mc.wt = hasReference ? std::string(1, ref.at(abs)) : std::string(1, col.MaxBase());
                            cur->msa.push_back(mc);
                        }
                    }
                }
            }
            variantGenes_.emplace_back(std::move(curVariantGene));
        }
        // with expected minors configured: the classification record goes to validation.json in the working directory
        // and to stderr, as the reference writes them (src/AminoAcidCaller.cpp:571-576)
        if (hasExpectedMinors) {
            std::ofstream outValJson("validation.json");
            outValJson << pm_->ToJson();
            std::cerr << *pm_ << std::endl;
        }
    }

    std::vector<VariantGene> variantGenes_;
    const ErrorEstimates error_;
    const TargetConfig targetConfig_;
    const bool debug_, drmOnly_;
    const double minimalPerc_, maximalPerc_;
    int numberOfTests_ = 1;
    std::unique_ptr<PerformanceMetrics> pm_;
    std::vector<Haplotype> reconstructedHaplotypes_, filteredHaplotypes_;  // AminoAcidCaller.h:138-139
    int genCounts_ = 0, margWithGap_ = 0, margWithHetero_ = 0, margPartial_ = 0, lowCov_ = 0, margOfftarget_ = 0;  // :148-153
};

/// The statements of src/tools/JulietWorkflow.cpp that only format what the path computed (the CLI, the HTML
/// report and the file handling around them stay with the caller).
struct JulietWorkflow
{
    /// `jsonStream << json.dump(2) << std::endl;` (JulietWorkflow.cpp:165-170)
    static void WriteJson(std::ostream& jsonStream, const AminoAcidCaller& aac) { jsonStream << aac.JSON().dump(2) << std::endl; }
    /// the --msa table: "pos A C G T - N", then one line per column with its 1-based position (JulietWorkflow.cpp:178-191)
    static void WriteMsa(std::ostream& msaStream, const Data::MSAByColumn& msa)
    {
        msaStream << "pos A C G T - N" << std::endl;
        int pos = msa.BeginPos();
        for (const auto& column : msa) {
            ++pos;
            msaStream << pos;
            for (const auto& c : {'A', 'C', 'G', 'T', '-', 'N'}) msaStream << " " << column[c];
            msaStream << std::endl;
        }
    }
    /// --mode-error (JulietWorkflow.cpp:193-213): mean deletion and substitution frequency over the columns with
    /// coverage > 100; prints "<file>\nsub: <x>\ndel: <y>\n" as the reference does (0/0 = nan without such a column).
    static void Error(std::ostream& out, const std::string& inputFile, const Data::MSAByColumn& msa)
    {
        double sub = 0;
        double del = 0;
        int columnCount = 0;
        for (const auto& column : msa) {
            if (column.Coverage() > 100) {
                del += column.Frequency('-');
                sub += 1.0 - column.Frequency('-') - column.Frequency(column.MaxBase());
                ++columnCount;
            }
        }
        out << inputFile << std::endl;
        out << "sub: " << (sub / columnCount) << std::endl;
        out << "del: " << (del / columnCount) << std::endl;
    }
};
}  // namespace Juliet

namespace Fuse {
/// fuse's consensus on the same pileup (include/pacbio/fuse/Fuse.h:50-72, src/tools/Fuse.cpp:60-140): per column
/// the majority base, plus the most abundant in-frame insertion where it is carried by more than half the column's
/// coverage, strongest first, suppressing weaker ones within 20 positions.  The pileup and the insertion histogram
/// come from the GPU (MSAByRow with InsertionMode::Majority); the few statements below are the host part.
class Fuse
{
public:
    explicit Fuse(const std::vector<Data::ArrayRead>& arrayReads) { consensusSequence_ = CreateConsensus(arrayReads); }
    /// the (ccsInput, minCoverage) constructor of the reference with the reads already decoded
    /// (FetchAlignedReads, src/tools/Fuse.cpp:142-153; see msq_bam.hpp for the BAM side)
    Fuse(const std::vector<Data::ArrayRead>& arrayReads, int minCoverage) : minCoverageRecommended_(minCoverage)
    {
        consensusSequence_ = CreateConsensus(arrayReads);
    }
    std::string ConsensusSequence() const { return consensusSequence_; }

private:
    std::string CreateConsensus(const std::vector<Data::ArrayRead>& arrayReads) const  // Fuse.cpp:70-97
    {
        if (arrayReads.empty()) throw std::runtime_error("Empty input. Could not find records.");
        const Data::MSAByRow rows(arrayReads, Data::InsertionMode::Majority);
        const Data::MSAByColumn msa(rows);
        const int actualCoverage = static_cast<int>(arrayReads.size());
        const int minCoverage = actualCoverage < minCoverageRecommended_ ? 1 : minCoverageRecommended_;  // permissive mode
        auto posInsCov = CollectInsertions(msa);
        std::map<int, std::string> posIns;
        while (!posInsCov.empty()) posIns.insert(FindInsertions(&posInsCov));
        std::string consensus;
        for (const auto& c : msa) {
            const auto ins = posIns.find(c.RefPos());
            if (ins != posIns.cend()) consensus += ins->second;
            if (c.Coverage() >= minCoverage) {
                const char maxBase = c.MaxBase();
                if (maxBase != '-' && maxBase != ' ') consensus += maxBase;
            }
        }
        return consensus;
    }
    /// RefPos -> (insertion, count): the most abundant insertion whose length is a multiple of three and whose count
    /// exceeds half the column's coverage; first in string order among equals (Fuse.cpp:99-119)
    std::map<int, std::pair<std::string, int>> CollectInsertions(const Data::MSAByColumn& msa) const
    {
        std::map<int, std::pair<std::string, int>> posInsCov;
        for (const auto& c : msa) {
            const double minInsertionCoverage = c.Coverage() * minInsertionCoverageFreq_;
            int best = -1;
            const std::string* bestSeq = nullptr;
            for (const auto& ins_count : c.Insertions()) {
                if (ins_count.first.size() % 3 != 0) continue;
                if (ins_count.second > best && ins_count.second > minInsertionCoverage) {
                    best = ins_count.second;
                    bestSeq = &ins_count.first;
                }
            }
            if (bestSeq) posInsCov[c.RefPos()] = std::make_pair(*bestSeq, best);
        }
        return posInsCov;
    }
    /// takes the best-covered entry (the left-most among equals) out of the map together with everything in
    /// [pos - windowSize, pos + windowSize) (Fuse.cpp:121-140)
    std::pair<int, std::string> FindInsertions(std::map<int, std::pair<std::string, int>>* posInsCov, int windowSize = 20) const
    {
        int argMax = -1, maxCov = -1;
        std::string ins;
        for (const auto& kv : *posInsCov)
            if (kv.second.second > maxCov) {
                maxCov = kv.second.second;
                argMax = kv.first;
                ins = kv.second.first;
            }
        posInsCov->erase(posInsCov->lower_bound(std::max(0, argMax - windowSize)), posInsCov->lower_bound(argMax + windowSize));
        return std::make_pair(argMax, ins);
    }

    const int minCoverageRecommended_ = 50;
    const double minInsertionCoverageFreq_ = 0.5;
    std::string consensusSequence_;
};
}  // namespace Fuse
}  // namespace PacBio
