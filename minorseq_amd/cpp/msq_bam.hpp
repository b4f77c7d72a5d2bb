// msq_bam.hpp -- BAM decode front: PacBio::IO::BamUtils::BamToArrayReads without pbbam.
//
// The reference reads aligned CCS records through pbbam (src/BamUtils.cpp:56-78) and turns each into a
// BAMArrayRead (src/ArrayRead.cpp:53-99).  pbbam is a submodule the reference does not vendor
// (.gitmodules:5-8, no pinned commit), so this is a minimal reader written from the SAM/BAM specification
// (BGZF: RFC 1952 members with the BC extra field; BAM records, SAMv1 section 4) that produces what those two
// functions produce:
//   * primary alignments only (flag 0x100 secondary and 0x800 supplementary are skipped, BamUtils.cpp:69-70),
//     overlapping [regionStart - 1, regionEnd - 1) (BamUtils.cpp:61-62,71), clipped to that reference interval
//     (record.Clip(CLIP_TO_REFERENCE, ...), :72);
//   * per remaining alignment column one ArrayBase: cigar char of "MIDNSHP=X" with the clips excised
//     (CigarData(exciseAllClips = true)), the aligned genomic sequence ('-' at D / N, '*' at P), the aligned
//     QUAL (0 at D / N / P; a record without QUAL gets QualQV 0 for every base, ArrayRead.cpp:94-96), and -- when
//     all three of the PacBio tags sq, dq, iq are present (ArrayRead.cpp:74-79) -- the substitution / deletion /
//     insertion QVs, which BAM stores in native orientation and are reversed for reverse-strand records.
// Parity unpinned at this boundary: the reference has no BAM fixture, and pbbam's treatment of an
// insertion that sits exactly on a clip border is not visible from this tree; such an insertion is kept here (it
// becomes a leading / trailing insertion, which MSAByRow::AddRead does not count: src/MSA.cpp:172-185).
// Inflating BGZF blocks is spread over hardware threads (blocks are independent); link with -lz.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <dlfcn.h>
#include <utility>
#include <vector>

#include "msq_front.hpp"

namespace PacBio {
namespace IO {
namespace detail {
inline uint32_t U32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t I32(const uint8_t* p) { return (int32_t)U32(p); }
inline uint16_t U16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct BgzfBlock
{
    size_t in_off, in_len;    // the deflate payload inside the file image
    size_t out_off, out_len;  // where it inflates to (ISIZE)
};

// The file image: mapped, not copied (the inflate threads read it straight from the page cache).
class MappedFile
{
public:
    explicit MappedFile(const std::string& path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw std::runtime_error("cannot stat " + path);
        }
        n_ = (size_t)st.st_size;
        if (n_) {
            void* m = ::mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                ::close(fd);
                throw std::runtime_error("cannot map " + path);
            }
            p_ = static_cast<const uint8_t*>(m);
        }
        ::close(fd);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    ~MappedFile()
    {
        if (p_) ::munmap(const_cast<uint8_t*>(p_), n_);
    }
    size_t size() const { return n_; }
    const uint8_t& operator[](size_t i) const { return p_[i]; }
    const uint8_t* data() const { return p_; }

private:
    const uint8_t* p_ = nullptr;
    size_t n_ = 0;
};

// The inflated BAM stream: an anonymous mapping, 2 MB aligned and advised to use huge pages -- every byte is written
// once by an inflate thread, and with 4 KB pages the first touches (a fault per page, 10^5 of them per 400 MB) cost
// the threads as much as inflating does.
class Bytes
{
public:
    explicit Bytes(size_t n) : n_(n)
    {
        constexpr size_t kHuge = size_t(2) << 20;
        map_len_ = ((n ? n : 1) + kHuge - 1) / kHuge * kHuge + kHuge;
        void* m = ::mmap(nullptr, map_len_, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) throw std::bad_alloc();
        map_ = static_cast<uint8_t*>(m);
        p_ = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(map_) + kHuge - 1) / kHuge * kHuge);
        const char* small = std::getenv("MSQ_BAM_SMALL_PAGES");
        if (!(small && small[0] == '1')) (void)::madvise(p_, map_len_ - kHuge, MADV_HUGEPAGE);  // advice only: small pages where it is refused
    }
    Bytes(Bytes&& o) noexcept : map_(o.map_), p_(o.p_), n_(o.n_), map_len_(o.map_len_) { o.map_ = o.p_ = nullptr; o.n_ = o.map_len_ = 0; }
    Bytes& operator=(Bytes&& o) noexcept
    {
        if (this != &o) {
            if (map_) ::munmap(map_, map_len_);
            map_ = o.map_; p_ = o.p_; n_ = o.n_; map_len_ = o.map_len_;
            o.map_ = o.p_ = nullptr; o.n_ = o.map_len_ = 0;
        }
        return *this;
    }
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
    ~Bytes()
    {
        if (map_) ::munmap(map_, map_len_);
    }
    size_t size() const { return n_; }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    uint8_t& operator[](size_t i) { return p_[i]; }
    const uint8_t& operator[](size_t i) const { return p_[i]; }

private:
    uint8_t* map_ = nullptr;
    uint8_t* p_ = nullptr;
    size_t n_ = 0, map_len_ = 0;
};

// libdeflate, where the system has it (libdeflate.so.0, found at run time: 2-3x zlib's inflate on BGZF blocks, which
// is half of file -> reads); zlib otherwise, or with MSQ_BAM_ZLIB=1 in the environment.  Same bytes either way: a
// block must inflate to exactly its ISIZE.
struct Deflater {
    void* (*alloc)(void) = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    static const Deflater* Get()
    {
        static const Deflater d = [] {
            Deflater x;
            const char* off = std::getenv("MSQ_BAM_ZLIB");
            if (off && off[0] == '1') return x;
            void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            x.alloc = reinterpret_cast<void* (*)(void)>(dlsym(h, "libdeflate_alloc_decompressor"));
            x.decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
            x.release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
            if (!x.alloc || !x.decompress || !x.release) x = Deflater();
            return x;
        }();
        return d.alloc ? &d : nullptr;
    }
};

// one member's payload through zlib: true when it inflates to exactly out_len bytes
inline bool InflateMember(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len)
{
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(in);
    zs.avail_in = (uInt)in_len;
    zs.next_out = out;
    zs.avail_out = (uInt)out_len;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.avail_out == 0;
}

// BGZF (SAMv1 4.1): gzip members with FEXTRA holding the subfield 'B','C',2,BSIZE.  Where the members are and what
// they inflate to; *total = the inflated stream's length.
// One member at `off`: its block (out_off left 0) and its size in the file.  Nothing in the header is trusted: the extra
// field must lie inside the file, every subfield inside the extra field, and the block must hold its header, the 8-byte
// trailer (CRC32, ISIZE) and end inside the file.
inline BgzfBlock ScanBgzfMember(const MappedFile& file, size_t off, size_t* bsizeOut)
{
    if (off + 18 > file.size() || file[off] != 31 || file[off + 1] != 139 || file[off + 2] != 8 || !(file[off + 3] & 4))
        throw std::runtime_error("not a BGZF block");
    const size_t xlen = U16(&file[off + 10]);
    const size_t xend = off + 12 + xlen;
    if (xend + 8 > file.size()) throw std::runtime_error("truncated BGZF block");
    size_t x = off + 12, bsize = 0;
    while (x + 4 <= xend) {
        const size_t slen = U16(&file[x + 2]);
        if (x + 4 + slen > xend) throw std::runtime_error("truncated BGZF block");
        if (file[x] == 'B' && file[x + 1] == 'C' && slen == 2) bsize = (size_t)U16(&file[x + 4]) + 1;
        x += 4 + slen;
    }
    if (!bsize || bsize < 12 + xlen + 8 || off + bsize > file.size()) throw std::runtime_error("truncated BGZF block");
    *bsizeOut = bsize;
    return BgzfBlock{xend, off + bsize - 8 - xend, 0, (size_t)U32(&file[off + bsize - 4])};
}

inline std::vector<BgzfBlock> ScanBgzfSerial(const MappedFile& file, size_t* total)
{
    std::vector<BgzfBlock> blocks;
    size_t off = 0, out = 0;
    while (off < file.size()) {
        size_t bsize = 0;
        BgzfBlock k = ScanBgzfMember(file, off, &bsize);
        k.out_off = out;
        blocks.push_back(k);
        out += k.out_len;
        off += bsize;
    }
    *total = out;
    return blocks;
}

// The members form a chain (each header holds its block's size), and walking it is a page fault or a cache miss per
// member: 4.4 ms for the 7000 members of a 280 MB file.  With threads, the file is cut into ranges; each range but the
// first starts at the first place in it that looks like a member the way every BAM writer makes them (the 16 fixed
// bytes of a gzip header with the one BC subfield), and is walked by its own thread up to the start of the next range.
// A range whose walk ends exactly on the next range's start proves that start to be a real member, given its own is
// (the first range starts at the file's start): then the result is the serial scan's, block for block.  Anything else
// -- a look-alike inside compressed data, a writer with other extra fields, a broken file -- and the serial scan decides.
inline std::vector<BgzfBlock> ScanBgzf(const MappedFile& file, size_t* total, unsigned threads = 1, bool* inRanges = nullptr)
{
    if (inRanges) *inRanges = false;
    const unsigned ranges = std::min<unsigned>(threads, 8);
    if (ranges < 2 || file.size() < (size_t(8) << 20)) return ScanBgzfSerial(file, total);
    static const uint8_t kHead[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    std::vector<size_t> start(ranges + 1, 0);
    start[ranges] = file.size();
    for (unsigned r = 1; r < ranges; ++r) {
        const size_t from = file.size() / ranges * r, upto = std::min(file.size() - 16, from + (size_t(1) << 17));
        size_t at = from;
        while (at < upto && std::memcmp(&file[at], kHead, 16) != 0) {
            const void* next = std::memchr(&file[at + 1], 31, upto - at);
            at = next ? size_t(static_cast<const uint8_t*>(next) - &file[0]) : upto;
        }
        if (at >= upto) return ScanBgzfSerial(file, total);
        start[r] = at;
    }
    std::vector<std::vector<BgzfBlock>> part(ranges);
    std::vector<char> good(ranges, 0);
    std::vector<std::thread> pool;
    for (unsigned r = 0; r < ranges; ++r)
        pool.emplace_back([&, r] {
            try {
                size_t off = start[r], out = 0;
                part[r].reserve((start[r + 1] - start[r]) / 16384 + 16);
                while (off < start[r + 1]) {
                    size_t bsize = 0;
                    BgzfBlock k = ScanBgzfMember(file, off, &bsize);
                    k.out_off = out;
                    part[r].push_back(k);
                    out += k.out_len;
                    off += bsize;
                }
                good[r] = off == start[r + 1];
            } catch (...) {  // the serial scan will say what is wrong, if anything is
            }
        });
    for (std::thread& t : pool) t.join();
    size_t n = 0;
    for (unsigned r = 0; r < ranges; ++r) {
        if (!good[r]) return ScanBgzfSerial(file, total);
        n += part[r].size();
    }
    std::vector<BgzfBlock> blocks;
    blocks.reserve(n);
    size_t out = 0;
    for (unsigned r = 0; r < ranges; ++r) {
        for (BgzfBlock k : part[r]) {
            k.out_off += out;
            blocks.push_back(k);
        }
        if (!part[r].empty()) out = blocks.back().out_off + blocks.back().out_len;
    }
    *total = out;
    if (inRanges) *inRanges = true;
    return blocks;
}

// ... and inflated in parallel on host threads.
inline Bytes InflateBgzf(const MappedFile& file, unsigned threads)
{
    size_t out = 0;
    const std::vector<BgzfBlock> blocks = ScanBgzf(file, &out, threads);
    Bytes data(out);
    const Deflater* fast = Deflater::Get();
    auto work = [&](size_t first, size_t step) {
        void* dec = fast ? fast->alloc() : nullptr;  // one decompressor per thread
        struct Release {
            const Deflater* f;
            void* d;
            ~Release() { if (d) f->release(d); }
        } release{fast, dec};
        for (size_t b = first; b < blocks.size(); b += step) {
            const BgzfBlock& k = blocks[b];
            if (k.out_len == 0) continue;
            if (dec) {
                size_t got = 0;
                const int rc = fast->decompress(dec, &file[k.in_off], k.in_len, &data[k.out_off], k.out_len, &got);
                if (rc != 0 || got != k.out_len) throw std::runtime_error("BGZF block does not inflate to its ISIZE");
                continue;
            }
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
            zs.next_in = const_cast<Bytef*>(&file[k.in_off]);
            zs.avail_in = (uInt)k.in_len;
            zs.next_out = &data[k.out_off];
            zs.avail_out = (uInt)k.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.avail_out != 0) throw std::runtime_error("BGZF block does not inflate to its ISIZE");
        }
    };
    threads = std::max(1u, std::min<unsigned>(threads, (unsigned)blocks.size()));
    if (threads == 1) {
        work(0, 1);
    } else {
        std::vector<std::thread> pool;
        std::vector<std::string> errors(threads);
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                try {
                    work(t, threads);
                } catch (const std::exception& e) {
                    errors[t] = e.what();
                }
            });
        for (auto& th : pool) th.join();
        for (const auto& e : errors)
            if (!e.empty()) throw std::runtime_error(e);
    }
    return data;
}

// PacBio per-base QV tags are 'Z' strings of Phred + 33 in native orientation
inline bool FindZTag(const uint8_t* p, const uint8_t* end, char t0, char t1, const uint8_t** val, size_t* len)
{
    while (p + 3 <= end) {
        const char a = (char)p[0], b = (char)p[1], type = (char)p[2];
        p += 3;
        size_t skip = 0;
        switch (type) {
            case 'A': case 'c': case 'C': skip = 1; break;
            case 's': case 'S': skip = 2; break;
            case 'i': case 'I': case 'f': skip = 4; break;
            case 'Z': case 'H': {
                const uint8_t* s = p;
                while (p < end && *p) ++p;
                if (a == t0 && b == t1 && type == 'Z') {
                    *val = s;
                    *len = (size_t)(p - s);
                    return true;
                }
                skip = 1;  // the terminating NUL
                break;
            }
            case 'B': {
                if (p + 5 > end) return false;
                const char sub = (char)p[0];
                const size_t n = U32(p + 1);
                const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                skip = 5 + n * w;
                break;
            }
            default: return false;
        }
        p += skip;
    }
    return false;
}
}  // namespace detail

/// BAMArrayRead (include/pacbio/data/ArrayRead.h:83-94) without the pbbam record behind it.
class BAMArrayRead : public Data::ArrayRead
{
public:
    BAMArrayRead(int idx, const std::string& name, size_t referenceStart, size_t referenceEnd, std::vector<Data::ArrayBase> bases,
                 std::string chemistry = "")
        : Data::ArrayRead(idx, name, referenceStart, referenceEnd, std::move(bases)), chemistry_(std::move(chemistry))
    {
    }
    /// The reference returns pbbam's chemistry name for the record's read group (src/ArrayRead.cpp:101-104); juliet
    /// only compares these strings between reads and warns on two names (src/tools/JulietWorkflow.cpp:144-148,
    /// src/ErrorEstimates.cpp:50).  pbbam's kit table is not in the reference tree, so this is the read group's
    /// "BINDINGKIT/SEQUENCINGKIT/BASECALLERVERSION" triple as written in its @RG DS field ("" without one): equal
    /// whenever pbbam's names are equal for one triple, never translated to a marketing name.
    std::string SequencingChemistry() const override { return chemistry_; }

private:
    std::string chemistry_;
};

struct BamTimings  // seconds, for tools and benches
{
    double read = 0, inflate = 0, index = 0, parse = 0;
    size_t file_bytes = 0, bam_bytes = 0, records = 0, kept = 0;
};

struct BamUtils
{
    /// src/BamUtils.cpp:56-78.  regionStart / regionEnd as juliet passes them (1-based; its defaults are 0 and
    /// INT_MAX, include/pacbio/juliet/JulietSettings.h:73-74).  Blocks are inflated and records decoded on `threads`
    /// host threads; the result is in file order with idx 0, 1, ... over the kept records, as in the reference.
    static std::vector<std::shared_ptr<Data::ArrayRead>> BamToArrayReads(const std::string& filePath, int regionStart,
                                                                        int regionEnd,
                                                                        unsigned threads = std::thread::hardware_concurrency(),
                                                                        BamTimings* timings = nullptr)
    {
        using namespace detail;
        using Clock = std::chrono::steady_clock;
        auto secs = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        if (threads == 0) threads = 1;
        regionStart = std::max(regionStart - 1, 0);
        regionEnd = std::max(regionEnd - 1, 0);
        const Loaded loaded = Load(filePath, regionStart, regionEnd, threads);
        const Bytes& data = loaded.data;
        const std::vector<Kept>& kept = loaded.kept;
        const auto& readGroups = loaded.readGroups;
        const auto t3 = Clock::now();
        // pass 2 (parallel): one ArrayRead per kept record
        std::vector<std::shared_ptr<Data::ArrayRead>> returnList(kept.size());
        auto parse = [&](size_t first, size_t last) {
            for (size_t i = first; i < last; ++i)
                returnList[i] = DecodeRecord(&data[kept[i].off], kept[i].block_size, kept[i].ref_start, kept[i].ref_end, regionStart,
                                             regionEnd, (int)i, readGroups, filePath);
        };
        const unsigned workers = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, kept.size() / 64 + 1));
        if (workers == 1) {
            parse(0, kept.size());
        } else {
            std::vector<std::thread> pool;
            std::vector<std::string> errors(workers);
            for (unsigned t = 0; t < workers; ++t)
                pool.emplace_back([&, t] {
                    try {
                        parse(kept.size() * t / workers, kept.size() * (t + 1) / workers);
                    } catch (const std::exception& e) {
                        errors[t] = e.what();
                    }
                });
            for (auto& th : pool) th.join();
            for (const auto& e : errors)
                if (!e.empty()) throw std::runtime_error(e);
        }
        const auto t4 = Clock::now();
        if (timings) {
            timings->read = loaded.t_read;
            timings->inflate = loaded.t_inflate;
            timings->index = loaded.t_index;
            timings->parse = secs(t3, t4);
            timings->file_bytes = loaded.file_bytes;
            timings->bam_bytes = data.size();
            timings->records = loaded.n_records;
            timings->kept = kept.size();
        }
        return returnList;
    }

    /// The same reads as BamToArrayReads, as a Data::ReadSource: names, reference intervals and base counts are
    /// worked out here (run arithmetic on the cigars); the ArrayBases of a read are decoded only when the device
    /// packer asks for them, straight into its pinned staging buffer -- no ArrayRead objects, no per-read allocation.
    /// `chemistries` receives the distinct SequencingChemistry() strings in order of appearance (juliet refuses more
    /// than one, src/tools/JulietWorkflow.cpp:144-148).
    static Data::ReadSource BamToReadSource(const std::string& filePath, int regionStart, int regionEnd,
                                            unsigned threads = std::thread::hardware_concurrency(), BamTimings* timings = nullptr,
                                            std::vector<std::string>* chemistries = nullptr)
    {
        using Clock = std::chrono::steady_clock;
        if (threads == 0) threads = 1;
        regionStart = std::max(regionStart - 1, 0);
        regionEnd = std::max(regionEnd - 1, 0);
        const auto loaded = std::make_shared<const Loaded>(Load(filePath, regionStart, regionEnd, threads));
        const auto t3 = Clock::now();
        Data::ReadSource source;
        const size_t n = loaded->kept.size();
        source.names.resize(n);
        source.referenceStart.resize(n);
        source.referenceEnd.resize(n);
        source.numBases.resize(n);
        const unsigned workers = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, n / 4096 + 1));
        std::vector<std::vector<std::string>> chemsOf(workers);  // distinct strings per worker, in order of appearance
        auto tables = [&](unsigned w) {
            for (size_t r = n * w / workers; r < n * (w + 1) / workers; ++r) {
                const Kept& k = loaded->kept[r];
                const uint8_t* rec = &loaded->data[k.off];
                source.names[r] = reinterpret_cast<const char*>(rec + 32);
                source.referenceStart[r] = (int32_t)std::max<int64_t>(k.ref_start, regionStart);
                source.referenceEnd[r] = (int32_t)std::min<int64_t>(k.ref_end, regionEnd);
                source.numBases[r] = CountBases(rec, k.ref_start, k.ref_end, regionStart, regionEnd);
                if (chemistries) {
                    std::string chem = ChemistryOf(rec, k.block_size, loaded->readGroups);
                    if (std::find(chemsOf[w].begin(), chemsOf[w].end(), chem) == chemsOf[w].end()) chemsOf[w].push_back(std::move(chem));
                }
            }
        };
        if (workers == 1) {
            tables(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w) pool.emplace_back(tables, w);
            for (auto& th : pool) th.join();
        }
        if (chemistries)
            for (const auto& part : chemsOf)  // workers own consecutive record ranges: concatenation keeps the order
                for (const auto& chem : part)
                    if (std::find(chemistries->begin(), chemistries->end(), chem) == chemistries->end()) chemistries->push_back(chem);
        source.fill = [loaded, regionStart, regionEnd, filePath](uint64_t read, Data::ArrayBase* dst) {
            const Kept& k = loaded->kept[read];
            DecodeBases(&loaded->data[k.off], k.block_size, k.ref_start, k.ref_end, regionStart, regionEnd, filePath,
                        [&dst](const Data::ArrayBase& b) { *dst++ = b; });
        };
        source.fillCompact = [loaded, regionStart, regionEnd, filePath](uint64_t read, uint8_t* dst, const msq_qv_thresholds& thr, uint32_t* presence) {
            const Kept& k = loaded->kept[read];
            return DecodeBasesCompact(&loaded->data[k.off], k.block_size, k.ref_start, k.ref_end, regionStart, regionEnd, filePath, thr, dst, presence);
        };
        if (timings) {
            timings->read = loaded->t_read;
            timings->inflate = loaded->t_inflate;
            timings->index = loaded->t_index + std::chrono::duration<double>(Clock::now() - t3).count();
            timings->parse = 0;  // the decode happens inside the packer
            timings->file_bytes = loaded->file_bytes;
            timings->bam_bytes = loaded->data.size();
            timings->records = loaded->n_records;
            timings->kept = n;
        }
        return source;
    }

    /// The same reads with the file decoded ON THE GPU (csrc/msq_bgzf.hip, msq_bamdev.hip): the compressed file image
    /// crosses PCIe once, BGZF members are inflated one wave each, the record chain, the filters, the reference intervals
    /// and base counts are worked out there, and MSAByRow packs the reads without the host (ReadSource::packDevice ->
    /// msq_bam_pack_device); only the per-record table and the names come back.  A file the device front declines
    /// (MSQ_ERR_DATA: a block that does not inflate, a malformed record, a cigar that some later step throws on) goes
    /// through BamToReadSource instead, which reports the error as before -- so the two are interchangeable.
    /// fill / fillCompact (literal bases for insertion strings beyond the packed form) load the file on the host on
    /// first use.  `usedDevice`, if given, says which front produced the source.
    static Data::ReadSource BamToReadSourceDevice(const std::string& filePath, int regionStart, int regionEnd,
                                                  unsigned threads = std::thread::hardware_concurrency(), BamTimings* timings = nullptr,
                                                  std::vector<std::string>* chemistries = nullptr, bool* usedDevice = nullptr)
    {
        using namespace detail;
        using Clock = std::chrono::steady_clock;
        auto secs = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        if (usedDevice) *usedDevice = false;
        if (threads == 0) threads = 1;
        const int regionStart0 = std::max(regionStart - 1, 0), regionEnd0 = std::max(regionEnd - 1, 0);
        msq_ctx* ctx = Msq::Context();
        struct DevBam  // the file in HBM, freed with the last copy of the source
        {
            msq_bam_dev* bam = nullptr;
            ~DevBam() { if (bam) msq_bam_dev_free(Msq::Context(), bam); }
        };
        const auto dev = std::make_shared<DevBam>();
        const auto t0 = Clock::now();
        size_t fileBytes = 0, bamBytes = 0;
        uint64_t nRecords = 0, nKept = 0;
        std::vector<std::pair<std::string, std::string>> readGroups;
        Clock::time_point t1, t2, tMapped, tStaging, tScanned;
        uint64_t first = 0;
        int32_t nRef = 0;
        {
            const MappedFile file(filePath);
            fileBytes = file.size();
            // the device front takes a file image of less than 16 GiB at once (msq_bgzf_stage_file)
            if (fileBytes >= (1ull << 34) - 4096) return BamToReadSource(filePath, regionStart, regionEnd, threads, timings, chemistries);
            // the file image starts on its way to the GPU (the library's copy thread) while this thread scans the BGZF
            // headers and reads the BAM header; whatever happens here, an image nobody will use is forgotten before the
            // mapping goes
            struct Staging
            {
                msq_ctx* ctx;
                ~Staging() { (void)msq_bgzf_stage_file(ctx, nullptr, 0); }  // nothing to forget once the inflate call has taken it
            } staging{ctx};
            tMapped = Clock::now();
            if (file.size()) (void)msq_bgzf_stage_file(ctx, file.data(), file.size());  // an error repeats in the inflate call
            tStaging = Clock::now();
            const std::vector<BgzfBlock> blocks = ScanBgzf(file, &bamBytes, threads);
            tScanned = Clock::now();
            std::vector<msq_bgzf_member> members(blocks.size());
            for (size_t b = 0; b < blocks.size(); ++b)
                members[b] = msq_bgzf_member{(uint64_t)blocks[b].in_off, (uint64_t)blocks[b].out_off, (uint32_t)blocks[b].in_len, (uint32_t)blocks[b].out_len};
            // The BAM header (text with the read groups, reference names; the first record sits behind it) is read from the
            // first members on the host -- a few KB of zlib work instead of a round trip after the device's inflate.
            {
                std::vector<uint8_t> head;
                size_t nextBlock = 0;
                auto need = [&](uint64_t n) {  // the first n bytes of the stream, or the host front's refusal
                    if (n > bamBytes) return false;
                    while (head.size() < n) {
                        if (nextBlock >= blocks.size()) return false;
                        const BgzfBlock& k = blocks[nextBlock++];
                        const size_t at = head.size();
                        head.resize(at + k.out_len);
                        if (k.out_len && !InflateMember(&file[k.in_off], k.in_len, head.data() + at, k.out_len)) return false;
                    }
                    return true;
                };
                bool ok = need(12) && std::memcmp(head.data(), "BAM\1", 4) == 0;
                uint64_t off = 4;
                size_t lText = 0;
                if (ok) {
                    lText = U32(&head[off]);
                    ok = need(off + 4 + lText + 4);
                }
                if (ok) {
                    readGroups = ReadGroupChemistries(std::string(reinterpret_cast<const char*>(&head[off + 4]), lText));
                    off += 4 + lText;
                    const size_t n = U32(&head[off]);
                    off += 4;
                    for (size_t r = 0; ok && r < n; ++r) {
                        ok = need(off + 4);
                        if (!ok) break;
                        const size_t lName = U32(&head[off]);
                        ok = lName <= bamBytes && need(off + 4 + lName + 4);
                        off += 4 + lName + 4;
                    }
                    ok = ok && n <= 0x7fffffffu;
                    nRef = (int32_t)n;
                    first = off;
                }
                if (!ok) return BamToReadSource(filePath, regionStart, regionEnd, threads, timings, chemistries);  // "not a BAM file" / "truncated BAM header" / a block that does not inflate
            }
            t1 = Clock::now();
            const int rc = msq_bgzf_inflate_device(ctx, file.data(), file.size(), members.data(), members.size(), &dev->bam);
            if (rc == MSQ_ERR_DATA) return BamToReadSource(filePath, regionStart, regionEnd, threads, timings, chemistries);
            Msq::Check(rc);
            t2 = Clock::now();
        }
        const auto t3 = Clock::now();
        {
            const int rc = msq_bam_index_device(ctx, dev->bam, first, nRef, regionStart0, regionEnd0, &nRecords, &nKept);
            if (rc == MSQ_ERR_DATA) return BamToReadSource(filePath, regionStart, regionEnd, threads, timings, chemistries);
            Msq::Check(rc);
        }
        const auto t4 = Clock::now();
        const msq_bam_record* const kept = msq_bam_records_view(dev->bam);  // the library's table: no copy of 72 bytes a record
        uint64_t nameBytes = 0;
        for (size_t r = 0; r < nKept; ++r) nameBytes += kept[r].l_read_name;
        std::vector<char> nameBuf(nameBytes + 1);
        Msq::Check(msq_bam_names(ctx, dev->bam, nameBuf.data(), nameBytes));
        const auto t5 = Clock::now();
        Data::ReadSource source;
        source.names.resize(nKept);
        source.referenceStart.resize(nKept);
        source.referenceEnd.resize(nKept);
        source.numBases.resize(nKept);
        std::vector<uint64_t> nameAt(nKept + 1, 0);
        for (size_t r = 0; r < nKept; ++r) {
            if (kept[r].rg_len == 254) return BamToReadSource(filePath, regionStart, regionEnd, threads, timings, chemistries);  // a read group id beyond the table's 16 chars
            nameAt[r + 1] = nameAt[r] + kept[r].l_read_name;
        }
        const unsigned workers = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, nKept / 4096 + 1));
        std::vector<std::vector<std::string>> chemsOf(workers);  // distinct strings per worker, in order of appearance
        auto tables = [&](unsigned w) {
            for (size_t r = nKept * w / workers; r < nKept * (w + 1) / workers; ++r) {
                const msq_bam_record& k = kept[r];
                source.names[r] = nameBuf.data() + nameAt[r];  // up to the name's NUL
                source.referenceStart[r] = std::max(k.ref_start, (int32_t)regionStart0);
                source.referenceEnd[r] = std::min(k.ref_end, (int32_t)regionEnd0);
                source.numBases[r] = k.n_bases;
                if (chemistries) {
                    std::string chem;
                    if (k.rg_len != 255) {
                        const std::string id(k.rg, k.rg_len);
                        for (const auto& g : readGroups)
                            if (g.first == id) chem = g.second;
                    }
                    if (std::find(chemsOf[w].begin(), chemsOf[w].end(), chem) == chemsOf[w].end()) chemsOf[w].push_back(std::move(chem));
                }
            }
        };
        if (workers == 1) {
            tables(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w) pool.emplace_back(tables, w);
            for (auto& th : pool) th.join();
        }
        if (chemistries)
            for (const auto& part : chemsOf)  // workers own consecutive record ranges: concatenation keeps the order
                for (const auto& chem : part)
                    if (std::find(chemistries->begin(), chemistries->end(), chem) == chemistries->end()) chemistries->push_back(chem);
        source.packDevice = [dev](uint32_t packFlags, size_t firstRead, size_t count, const msq_qv_thresholds& thr, msq_packed_dev** out) {
            return msq_bam_pack_device(Msq::Context(), dev->bam, firstRead, count, packFlags, &thr, out);
        };
        // literal bases of a read (rare: insertion strings the packed form cannot hold): the host front, loaded on first use
        struct Lazy
        {
            std::string path;
            int regionStart, regionEnd;
            unsigned threads;
            std::mutex mutex;
            std::shared_ptr<const Loaded> loaded;
            const Loaded& Get()
            {
                std::lock_guard<std::mutex> lock(mutex);
                if (!loaded) loaded = std::make_shared<const Loaded>(Load(path, regionStart, regionEnd, threads));
                return *loaded;
            }
        };
        const auto lazy = std::make_shared<Lazy>();
        lazy->path = filePath;
        lazy->regionStart = regionStart0;
        lazy->regionEnd = regionEnd0;
        lazy->threads = threads;
        source.fill = [lazy](uint64_t read, Data::ArrayBase* dst) {
            const Loaded& L = lazy->Get();
            const Kept& k = L.kept[read];
            DecodeBases(&L.data[k.off], k.block_size, k.ref_start, k.ref_end, lazy->regionStart, lazy->regionEnd, lazy->path,
                        [&dst](const Data::ArrayBase& b) { *dst++ = b; });
        };
        source.fillCompact = [lazy](uint64_t read, uint8_t* dst, const msq_qv_thresholds& thr, uint32_t* presence) {
            const Loaded& L = lazy->Get();
            const Kept& k = L.kept[read];
            return DecodeBasesCompact(&L.data[k.off], k.block_size, k.ref_start, k.ref_end, lazy->regionStart, lazy->regionEnd, lazy->path, thr, dst, presence);
        };
        if (usedDevice) *usedDevice = true;
        if (std::getenv("MSQ_BAM_TRACE"))
            std::fprintf(stderr, "BamToReadSourceDevice: map %.2f ms, staging started %.2f, scan %.2f, header %.2f, inflate %.2f, index %.2f, records + names %.2f, source %.2f\n",
                         1e3 * secs(t0, tMapped), 1e3 * secs(tMapped, tStaging), 1e3 * secs(tStaging, tScanned), 1e3 * secs(tScanned, t1), 1e3 * secs(t1, t2), 1e3 * secs(t3, t4), 1e3 * secs(t4, t5), 1e3 * secs(t5, Clock::now()));
        if (timings) {
            timings->read = secs(t0, t1);
            timings->inflate = secs(t1, t2);
            timings->index = secs(t2, Clock::now());
            timings->parse = 0;
            timings->file_bytes = fileBytes;
            timings->bam_bytes = bamBytes;
            timings->records = nRecords;
            timings->kept = nKept;
        }
        return source;
    }

private:
    struct Kept  // a record that passed the filters
    {
        size_t off;  // of the record body in the inflated stream
        size_t block_size;
        int64_t ref_start, ref_end;
    };
    struct Loaded
    {
        detail::Bytes data{0};
        std::vector<Kept> kept;
        std::vector<std::pair<std::string, std::string>> readGroups;
        size_t n_records = 0, file_bytes = 0;
        double t_read = 0, t_inflate = 0, t_index = 0;
    };
    /// map + inflate + header + pass 1 (cheap, sequential): record offsets and filters; regionStart / regionEnd 0-based here
    static Loaded Load(const std::string& filePath, int regionStart, int regionEnd, unsigned threads)
    {
        using namespace detail;
        using Clock = std::chrono::steady_clock;
        auto secs = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const auto t0 = Clock::now();
        const MappedFile file(filePath);
        const auto t1 = Clock::now();
        Loaded L;
        L.data = InflateBgzf(file, threads);
        const Bytes& data = L.data;
        L.file_bytes = file.size();
        const auto t2 = Clock::now();
        if (data.size() < 12 || std::memcmp(data.data(), "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file: " + filePath);
        size_t off = 4;
        const size_t l_text = U32(&data[off]);
        if (off + 4 + l_text + 4 > data.size()) throw std::runtime_error("truncated BAM header");
        L.readGroups =
            ReadGroupChemistries(std::string(reinterpret_cast<const char*>(&data[off + 4]), l_text));
        off += 4 + l_text;
        const size_t n_ref = U32(&data[off]);
        off += 4;
        for (size_t r = 0; r < n_ref; ++r) {
            if (off + 4 > data.size()) throw std::runtime_error("truncated BAM header");
            const size_t l_name = U32(&data[off]);
            if (l_name > data.size() || off + 4 + l_name + 4 > data.size()) throw std::runtime_error("truncated BAM header");
            off += 4 + l_name + 4;
        }
        // pass 1a (sequential, one length word per record): where the records are.  A record that runs past the end of
        // the stream is reported after every malformed record before it, as one pass in file order would
        std::vector<size_t> bodies;
        bool truncated = false;
        while (off + 4 <= data.size()) {
            const size_t block_size = U32(&data[off]);
            const size_t body = off + 4;
            off += 4 + block_size;
            if (off > data.size() || block_size < 32) {
                truncated = true;
                break;
            }
            bodies.push_back(body);
        }
        L.n_records = bodies.size();
        // pass 1b (threads over record ranges): the record's own consistency, the filters, the reference interval
        const size_t nrec = bodies.size();
        std::vector<Kept> all(nrec);
        std::vector<uint8_t> keep(nrec, 0);
        const unsigned workers = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, nrec / 4096 + 1));
        std::vector<size_t> bad(workers, SIZE_MAX);  // first malformed record of each range
        auto scan = [&](unsigned w) {
            for (size_t i = nrec * w / workers; i < nrec * (w + 1) / workers; ++i) {
                const size_t body = bodies[i];
                const size_t block_size = U32(&data[body - 4]);
                const uint8_t* rec = &data[body];
                const size_t l_read_name = rec[8];
                const size_t n_cigar = U16(rec + 12);
                const uint16_t flag = U16(rec + 14);
                const size_t l_seq = U32(rec + 16);
                if (32 + l_read_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq > block_size) {
                    bad[w] = i;
                    return;
                }
                if (flag & 0x800) continue;  // IsSupplementaryAlignment (BamUtils.cpp:69)
                if (flag & 0x100) continue;  // !IsPrimaryAlignment (BamUtils.cpp:70)
                if (flag & 0x4) continue;    // unmapped: no reference interval to overlap
                const uint8_t* cigar = rec + 32 + l_read_name;
                int64_t ref_len = 0;
                for (size_t k = 0; k < n_cigar; ++k) {
                    const uint32_t c = U32(cigar + 4 * k);
                    const int op = c & 15;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4;
                }
                const int64_t ref_start = I32(rec + 4), ref_end = ref_start + ref_len;
                if (!(ref_start < regionEnd && ref_end > regionStart)) continue;  // BamUtils.cpp:71
                all[i] = {body, block_size, ref_start, ref_end};
                keep[i] = 1;
            }
        };
        if (workers == 1) {
            scan(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w) pool.emplace_back(scan, w);
            for (auto& th : pool) th.join();
        }
        for (unsigned w = 0; w < workers; ++w)
            if (bad[w] != SIZE_MAX) throw std::runtime_error("malformed BAM record");
        if (truncated) throw std::runtime_error("truncated BAM record");
        for (size_t i = 0; i < nrec; ++i)
            if (keep[i]) L.kept.push_back(all[i]);
        const auto t3 = Clock::now();
        L.t_read = secs(t0, t1);
        L.t_inflate = secs(t1, t2);
        L.t_index = secs(t2, t3);
        return L;
    }

    /// (@RG ID, "BINDINGKIT/SEQUENCINGKIT/BASECALLERVERSION") of every read group in the header text
    static std::vector<std::pair<std::string, std::string>> ReadGroupChemistries(const std::string& text)
    {
        std::vector<std::pair<std::string, std::string>> out;
        size_t p = 0;
        while (p < text.size()) {
            size_t e = text.find('\n', p);
            if (e == std::string::npos) e = text.size();
            if (text.compare(p, 4, "@RG\t") == 0) {
                std::string id, ds;
                size_t f = p + 4;
                while (f < e) {
                    size_t g = text.find('\t', f);
                    if (g == std::string::npos || g > e) g = e;
                    if (text.compare(f, 3, "ID:") == 0) id = text.substr(f + 3, g - f - 3);
                    if (text.compare(f, 3, "DS:") == 0) ds = text.substr(f + 3, g - f - 3);
                    f = g + 1;
                }
                std::string kit[3];
                static const char* const keys[3] = {"BINDINGKIT=", "SEQUENCINGKIT=", "BASECALLERVERSION="};
                size_t q = 0;
                while (q < ds.size()) {
                    size_t h = ds.find(';', q);
                    if (h == std::string::npos) h = ds.size();
                    for (int k = 0; k < 3; ++k)
                        if (ds.compare(q, std::strlen(keys[k]), keys[k]) == 0) kit[k] = ds.substr(q + std::strlen(keys[k]), h - q - std::strlen(keys[k]));
                    q = h + 1;
                }
                const bool any = !kit[0].empty() || !kit[1].empty() || !kit[2].empty();
                out.emplace_back(id, any ? kit[0] + "/" + kit[1] + "/" + kit[2] : std::string());
            }
            p = e + 1;
        }
        return out;
    }

    static std::string ChemistryOf(const uint8_t* rec, size_t block_size, const std::vector<std::pair<std::string, std::string>>& readGroups)
    {
        using namespace detail;
        const size_t l_read_name = rec[8], n_cigar = U16(rec + 12), l_seq = U32(rec + 16);
        const uint8_t* tags = rec + 32 + l_read_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq;
        const uint8_t* rg = nullptr;
        size_t nrg = 0;
        std::string chemistry;
        if (FindZTag(tags, rec + block_size, 'R', 'G', &rg, &nrg)) {
            const std::string id(reinterpret_cast<const char*>(rg), nrg);
            for (const auto& g : readGroups)
                if (g.first == id) chemistry = g.second;
        }
        return chemistry;
    }

    static std::shared_ptr<Data::ArrayRead> DecodeRecord(const uint8_t* rec, size_t block_size, int64_t ref_start, int64_t ref_end,
                                                         int regionStart, int regionEnd, int idx,
                                                         const std::vector<std::pair<std::string, std::string>>& readGroups,
                                                         const std::string& filePath)
    {
        const int64_t clip_lo = std::max<int64_t>(ref_start, regionStart), clip_hi = std::min<int64_t>(ref_end, regionEnd);
        std::vector<Data::ArrayBase> bases;
        bases.reserve((size_t)(clip_hi - clip_lo) + 16);
        DecodeBases(rec, block_size, ref_start, ref_end, regionStart, regionEnd, filePath, [&](const Data::ArrayBase& b) { bases.push_back(b); });
        return std::make_shared<BAMArrayRead>(idx, std::string(reinterpret_cast<const char*>(rec + 32)), (size_t)clip_lo, (size_t)clip_hi,
                                              std::move(bases), ChemistryOf(rec, block_size, readGroups));
    }

    /// Number of ArrayBases DecodeBases will emit for this record (the clipped alignment columns): run arithmetic only.
    static int64_t CountBases(const uint8_t* rec, int64_t ref_start, int64_t ref_end, int regionStart, int regionEnd)
    {
        using namespace detail;
        const size_t l_read_name = rec[8], n_cigar = U16(rec + 12);
        const uint8_t* cigar = rec + 32 + l_read_name;
        const int64_t clip_lo = std::max<int64_t>(ref_start, regionStart), clip_hi = std::min<int64_t>(ref_end, regionEnd);
        int64_t rp = ref_start, n = 0;
        for (size_t k = 0; k < n_cigar; ++k) {
            const uint32_t c = U32(cigar + 4 * k);
            const int op = c & 15;
            const int64_t len = c >> 4;
            if (op == 4 || op == 5 || op > 8) continue;
            const bool cons_ref = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
            if (cons_ref) {
                const int64_t first = std::max(rp, clip_lo), last = std::min(rp + len, clip_hi);
                if (last > first) n += last - first;
                rp += len;
            } else if (rp >= clip_lo && rp <= clip_hi) {
                n += len;
            }
        }
        return n;
    }

    /// The same walk with every base translated on the spot (Data::ReadSource::fillCompact): one byte per base through
    /// the library's table over (Cigar, Nucleotide), msq_compact_fail where a QV misses a threshold
    /// (ArrayBase::MeetQVThresholds, ArrayRead.inl:67-87: every base DecodeBases makes carries a QualQV, and the rich
    /// QVs when the record has all three tags).  Returns -1 or (index << 8 | cigar) of the first base whose cigar the
    /// reference's MSA would throw on.
    static int64_t DecodeBasesCompact(const uint8_t* rec, size_t block_size, int64_t ref_start, int64_t ref_end, int regionStart, int regionEnd,
                                      const std::string& filePath, const msq_qv_thresholds& thr, uint8_t* dst, uint32_t* presence)
    {
        using namespace detail;
        // per cigar op the byte of each of the sixteen SEQ codes ("=ACMGRSVTWYHKDBN"), and of the op's placeholder
        // base ('-' for D / N, '*' for P): the library's table, looked up once per process
        struct Rows {
            uint8_t seq[9][16], gap[9];
            Rows()
            {
                static const char kSeq[] = "=ACMGRSVTWYHKDBN";
                static const char kOps[] = "MIDNSHP=X";
                const uint8_t* table = msq_compact_table();
                for (int op = 0; op < 9; ++op) {
                    for (int v = 0; v < 16; ++v) seq[op][v] = table[(uint8_t)kOps[op] | ((unsigned)(uint8_t)kSeq[v] << 8)];
                    gap[op] = table[(uint8_t)kOps[op] | ((unsigned)(uint8_t)(op == 6 ? '*' : '-') << 8)];
                }
            }
        };
        static const Rows rows;
        static const char kOps[] = "MIDNSHP=X";
        const size_t l_read_name = rec[8];
        const size_t n_cigar = U16(rec + 12);
        const uint16_t flag = U16(rec + 14);
        const size_t l_seq = U32(rec + 16);
        const uint8_t* cigar = rec + 32 + l_read_name;
        const uint8_t* seq = cigar + 4 * n_cigar;
        const uint8_t* qual = seq + (l_seq + 1) / 2;
        const uint8_t* tags = qual + l_seq;
        const uint8_t* end = rec + block_size;
        const bool hasQual = l_seq > 0 && qual[0] != 0xFF;
        const uint8_t *sq = nullptr, *dq = nullptr, *iq = nullptr;
        size_t nsq = 0, ndq = 0, niq = 0;
        const bool rich = FindZTag(tags, end, 's', 'q', &sq, &nsq) && FindZTag(tags, end, 'd', 'q', &dq, &ndq) &&
                          FindZTag(tags, end, 'i', 'q', &iq, &niq) && nsq == l_seq && ndq == l_seq && niq == l_seq;
        const bool reverse = flag & 0x10;
        const bool anyThr = thr.qual >= 0 || thr.del >= 0 || thr.sub >= 0 || thr.ins >= 0;
        const int64_t clip_lo = std::max<int64_t>(ref_start, regionStart), clip_hi = std::min<int64_t>(ref_end, regionEnd);
        int64_t rp = ref_start, firstBad = -1;
        size_t qp = 0;
        uint8_t* out = dst;
        for (size_t k = 0; k < n_cigar; ++k) {
            const uint32_t cg = U32(cigar + 4 * k);
            const int op = cg & 15;
            const size_t len = cg >> 4;
            if (op > 8) throw std::runtime_error("bad CIGAR op in " + filePath);
            const bool cons_ref = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
            const bool cons_query = op == 0 || op == 1 || op == 4 || op == 7 || op == 8;
            if (cons_query && qp + len > l_seq) throw std::runtime_error("CIGAR longer than SEQ in " + filePath);
            if (op == 4 || op == 5) {  // clips are excised
                if (cons_query) qp += len;
                continue;
            }
            size_t skip = 0, count = len;  // the part of the run inside the clip interval, as in DecodeBases
            if (cons_ref) {
                const int64_t first = std::max<int64_t>(rp, clip_lo), last = std::min<int64_t>(rp + (int64_t)len, clip_hi);
                skip = (size_t)std::max<int64_t>(first - rp, 0);
                count = last > first ? (size_t)(last - first) : 0;
            } else if (!(rp >= clip_lo && rp <= clip_hi)) {
                count = 0;
            }
            if (count && ((cons_query ? rows.seq[op][1] : rows.gap[op]) & 0x80)) {  // a cigar the MSA throws on ('M', 'N')
                if (firstBad < 0) firstBad = ((int64_t)(out - dst) << 8) | (uint8_t)kOps[op];
                std::memset(out, 12, count);
                out += count;
            } else if (cons_query) {
                const uint8_t* row = rows.seq[op];
                const size_t q0 = qp + skip, q1 = q0 + count;
                if (!anyThr) {
                    for (size_t q = q0; q < q1; ++q) *out++ = row[(seq[q >> 1] >> ((q & 1) ? 0 : 4)) & 15];
                } else {
                    // ArrayBase::MeetQVThresholds (ArrayRead.inl:67-87) without branches; every base of a record carries a
                    // QualQV (0 without QUAL, ArrayRead.cpp:94-96), the rich QVs when all three tags are there
                    for (size_t q = q0; q < q1; ++q) {
                        const uint8_t c = row[(seq[q >> 1] >> ((q & 1) ? 0 : 4)) & 15];
                        unsigned fail = (unsigned)((int)(hasQual ? qual[q] : 0) < thr.qual);
                        if (rich) {
                            const size_t i = reverse ? l_seq - 1 - q : q;  // tags are native-oriented; SEQ / QUAL genomic
                            fail |= (unsigned)((int)(uint8_t)(dq[i] - 33) < thr.del) | (unsigned)((int)(uint8_t)(sq[i] - 33) < thr.sub) |
                                    (unsigned)((int)(uint8_t)(iq[i] - 33) < thr.ins);
                        }
                        const uint8_t masked = MSQ_COMPACT_FAIL(c);
                        *out++ = fail ? masked : c;
                    }
                }
            } else if (count) {
                // a placeholder base ('-' / '*', QualQV 0): a threshold can only turn a match symbol into 'N', and the
                // table makes none of these one
                uint8_t c = rows.gap[op];
                if (anyThr && (0 < thr.qual || (rich && (0 < thr.del || 0 < thr.sub || 0 < thr.ins)))) c = MSQ_COMPACT_FAIL(c);
                std::memset(out, c, count);
                out += count;
            }
            if (cons_query) qp += len;
            if (cons_ref) rp += (int64_t)len;
        }
        if (out != dst) *presence |= 1u | (rich ? 2u : 0u);
        return firstBad;
    }

    /// BAMArrayRead's constructor (src/ArrayRead.cpp:53-99) on the record clipped to the region: one ArrayBase per
    /// remaining alignment column, handed to `sink` in order.
    template <typename Sink>
    static void DecodeBases(const uint8_t* rec, size_t block_size, int64_t ref_start, int64_t ref_end, int regionStart, int regionEnd,
                            const std::string& filePath, Sink&& sink)
    {
        using namespace detail;
        static const char kSeq[] = "=ACMGRSVTWYHKDBN";
        static const char kOps[] = "MIDNSHP=X";
        const size_t l_read_name = rec[8];
        const size_t n_cigar = U16(rec + 12);
        const uint16_t flag = U16(rec + 14);
        const size_t l_seq = U32(rec + 16);
        const uint8_t* name = rec + 32;
        const uint8_t* cigar = name + l_read_name;
        const uint8_t* seq = cigar + 4 * n_cigar;
        const uint8_t* qual = seq + (l_seq + 1) / 2;
        const uint8_t* tags = qual + l_seq;
        const uint8_t* end = rec + block_size;
        const bool hasQual = l_seq > 0 && qual[0] != 0xFF;
        const uint8_t *sq = nullptr, *dq = nullptr, *iq = nullptr;
        size_t nsq = 0, ndq = 0, niq = 0;
        const bool rich = FindZTag(tags, end, 's', 'q', &sq, &nsq) && FindZTag(tags, end, 'd', 'q', &dq, &ndq) &&
                          FindZTag(tags, end, 'i', 'q', &iq, &niq) && nsq == l_seq && ndq == l_seq && niq == l_seq;
        const bool reverse = flag & 0x10;
        auto tagQv = [&](const uint8_t* t, size_t q) -> uint8_t {  // tags are native-oriented; SEQ / QUAL genomic
            const size_t i = reverse ? l_seq - 1 - q : q;
            return (uint8_t)(t[i] - 33);
        };
        // walk the alignment, clipped to [regionStart, regionEnd)
        const int64_t clip_lo = std::max<int64_t>(ref_start, regionStart), clip_hi = std::min<int64_t>(ref_end, regionEnd);
        int64_t rp = ref_start;
        size_t qp = 0;
        auto emit = [&](char op, char nt, bool has_query, size_t q) {
            const uint8_t qv = has_query ? (hasQual ? qual[q] : 0) : 0;
            if (rich)
                sink(Data::ArrayBase(op, nt, qv, has_query ? tagQv(sq, q) : 0, has_query ? tagQv(dq, q) : 0, has_query ? tagQv(iq, q) : 0));
            else
                sink(Data::ArrayBase(op, nt, qv));  // a record without QUAL: QualQV 0 (ArrayRead.cpp:94-96)
        };
        for (size_t k = 0; k < n_cigar; ++k) {
            const uint32_t c = U32(cigar + 4 * k);
            const int op = c & 15;
            const size_t len = c >> 4;
            if (op > 8) throw std::runtime_error("bad CIGAR op in " + filePath);
            const bool cons_ref = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;
            const bool cons_query = op == 0 || op == 1 || op == 4 || op == 7 || op == 8;
            if (cons_query && qp + len > l_seq) throw std::runtime_error("CIGAR longer than SEQ in " + filePath);
            if (op == 4 || op == 5) {  // clips are excised
                if (cons_query) qp += len;
                continue;
            }
            // the part of this run that lies inside the clip interval (a run that consumes no reference sits at rp)
            size_t skip = 0, count = len;
            if (cons_ref) {
                const int64_t first = std::max<int64_t>(rp, clip_lo), last = std::min<int64_t>(rp + (int64_t)len, clip_hi);
                skip = (size_t)std::max<int64_t>(first - rp, 0);
                count = last > first ? (size_t)(last - first) : 0;
            } else if (!(rp >= clip_lo && rp <= clip_hi)) {
                count = 0;
            }
            if (cons_query) {
                const char opc = kOps[op];
                for (size_t q = qp + skip, e = qp + skip + count; q < e; ++q) emit(opc, kSeq[(seq[q >> 1] >> ((q & 1) ? 0 : 4)) & 15], true, q);
                qp += len;
            } else {
                const char opc = kOps[op], nt = op == 6 ? '*' : '-';
                for (size_t i = 0; i < count; ++i) emit(opc, nt, false, 0);
            }
            if (cons_ref) rp += (int64_t)len;
        }
    }
};
}  // namespace IO
}  // namespace PacBio
