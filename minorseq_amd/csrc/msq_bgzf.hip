// msq_bgzf.hip -- BGZF members inflated on the GPU (the BAM front's first step).
//
// The reference reads BAM through pbbam, which inflates BGZF with zlib on the host (src/BamUtils.cpp:56-78 only sees
// records).  A BGZF file (SAMv1 4.1) is a run of independent gzip members of <= 64 KiB, each one raw DEFLATE stream
// (RFC 1951) that inflates to exactly its ISIZE -- so the members of a file are the parallel units: ONE WAVE PER
// MEMBER, 4 waves per workgroup, every member's Huffman tables in its wave's 4.7 KB of LDS.
//   * The compressed bits are wave-uniform state: a 64-bit bit buffer in scalar registers, refilled 32 bits at a
//     time from a VGPR that holds the next 256 input bytes (lane i = dword i, read with v_readlane; the chunk after
//     it is loaded a whole chunk ahead), so the decode loop never waits for global memory.
//   * Huffman decode: one LDS lookup per symbol in a 2^10 (literal/length) or 2^9 (distance) entry table whose entries
//     hold the answer (a literal's byte; a length's base and extra bits; a distance's top bits and extra bits); longer
//     codes take the canonical first-code / count compare per length.  The tables are built by the 64 lanes (counts by
//     LDS atomics, canonical order by ballots, every lane fills its 16 entries).
//   * Literals are decoded speculatively: lane k looks up the code that would start k bits into the bit buffer, the
//     wave walks the chain of real code starts through those 64 entries with v_readlane (four instructions per literal),
//     and the window's literals leave in one store.  A kernel of this shape is bound by the CU's scalar unit -- one
//     instruction per cycle and CU -- not by memory, and its time follows the count of scalar instructions one for one:
//     the hot loop is hand-written where the compiler spent them (see the window's asm statement, inflate_fail, the
//     entry formats).  A match is copied by all lanes at once (source index modulo the distance where the match overlaps
//     itself) straight in HBM: the wave reads what it has just stored, which the vector memory pipeline keeps in order.
// Byte work, bit-exact: the test is equality with zlib's output (tests/test_gpu_bgzf.py).  A member that zlib would
// refuse (invalid block type, stored length check, over-subscribed or incomplete code, repeat without a previous
// length, missing end-of-block code, invalid symbol, distance too far back, input or output exhausted) is refused
// here with a status word per member, and the call fails as the host front does ("BGZF block does not inflate to
// its ISIZE", cpp/msq_bam.hpp).  Every loop is bounded by the member's input bits or its ISIZE (<= 64 KiB, checked
// on the host), every load is clamped to the padded file image: a crafted member cannot hang or fault the wave.
#include "msq_internal.h"

#include <algorithm>
#include <chrono>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include <unistd.h>

#include "msq_bam_dev.h"

namespace {

enum : uint32_t {
    INF_OK = 0,
    INF_ERR_INPUT = 1,   // the stream runs past the member's payload
    INF_ERR_BTYPE = 2,   // block type 3
    INF_ERR_STORED = 3,  // LEN != ~NLEN
    INF_ERR_COUNTS = 4,  // more than 286 literal/length or 30 distance codes
    INF_ERR_CODES = 5,   // over-subscribed or incomplete set of code lengths
    INF_ERR_REPEAT = 6,  // repeat with no previous length, or past the last length
    INF_ERR_NOEOB = 7,   // no end-of-block code
    INF_ERR_SYMBOL = 8,  // bits that are no code, or literal/length symbol 286 / 287
    INF_ERR_DIST = 9,    // distance symbol 30 / 31, or a distance before the member's first byte
    INF_ERR_OUTPUT = 10, // more than ISIZE bytes
    INF_ERR_SHORT = 11,  // fewer than ISIZE bytes
};

constexpr int LIT_BITS = 10, DIST_BITS = 9, CL_BITS = 7;
// A primary-table entry is (payload << 4 | code length); code length 0: a longer code, or none.  Bit 15 is set on
// everything the fast path does not take as it is (one compare keeps the literal loop going):
//   code-length alphabet   payload = symbol
//   literal/length         a literal: payload = the byte.  Anything else: bit 15, and payload = (base - 3) << 3 | extra
//                          bits for a length code (RFC 1951 3.2.5: bases 3..258, 0..5 extra bits), 6 for the end-of-block
//                          code, 7 for symbols 286 / 287 -- the match path has no arithmetic on symbol numbers left
//   distance               payload = top << 4 | extra bits, distance = (top << extra) + 1 + the extra bits' value (top =
//                          the symbol below 4, else 2 + its low bit; extra = symbol / 2 - 1); symbols 30 / 31: bit 15
constexpr uint32_t kNotLiteral = 0x8000u;
enum { CODE_PLAIN = 0, CODE_LITLEN = 1, CODE_DIST = 2 };
constexpr uint32_t kLenEob = 6, kLenInvalid = 7;

__device__ inline uint32_t litlen_payload(uint32_t sym)  // of a symbol >= 256
{
    if (sym == 256) return kLenEob;
    if (sym > 285) return kLenInvalid;
    if (sym < 265) return (sym - 257u) << 3;
    if (sym == 285) return 255u << 3;
    const uint32_t xb = (sym - 261u) >> 2;
    return (((4u + ((sym - 265u) & 3u)) << xb) << 3) | xb;
}
__device__ inline uint32_t dist_payload(uint32_t sym)  // of a symbol <= 29
{
    if (sym < 4) return sym << 4;
    return ((2u + (sym & 1u)) << 4) | ((sym >> 1) - 1u);
}

struct CodeSet {  // a canonical Huffman code, per code length 1..15: first code, number of codes, offset into `sorted`
    uint16_t first[16], count[16], offs[16];
};

// timing-experiment switches (tools/inf_exp.sh builds the library with them into minorseq_amd/lib_expN): the scope of the
// fence in front of a match's copy ("wavefront": no difference measured), INF_WAVES = waves per SIMD to compile for (8: -3 %
// at the price of spills)
#ifndef INF_FENCE_SCOPE
#define INF_FENCE_SCOPE "workgroup"
#endif

struct WaveLds {
    uint16_t lit[1 << LIT_BITS];    // (symbol << 4 | length | kNotLiteral) by the next LIT_BITS bits; length 0: a longer code, or none
    uint16_t dist[1 << DIST_BITS];
    uint16_t cl[1 << CL_BITS];      // the code-length alphabet of a dynamic block's header
    uint16_t sorted_lit[288], sorted_dist[32], sorted_cl[20];  // symbols in canonical order
    CodeSet cs_lit, cs_dist, cs_cl;
    uint32_t cnt[16];
    uint8_t lens[320];              // code lengths: literal/length alphabet, then distance alphabet
    uint8_t cl_lens[20];
};

__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// One wave owns its WaveLds and the LDS executes a wave's operations in order: all that is needed between a lane's
// write and another lane's read is that the compiler keeps them in order.
#define WAVE_SYNC()                                           \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)

__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The member's compressed bits (see the header).  Everything but cur / nxt is wave-uniform.
struct BitReader {
    const uint32_t* words;  // the file image as dwords, zero-padded
    uint64_t nwords;
    uint32_t widx;          // the next dword to take (the host refuses images of 16 GiB and more)
    uint64_t buf;
    int cnt;                // valid bits in buf
    uint32_t cur, nxt;      // lane i: dword (widx & ~63) + i, and the one 64 dwords on
    int lane;

    // Clamped, not predicated (the last dwords of the image are zero padding), and hand-written with its wait inside: a
    // load the compiler tracks stays "pending" in its view of cur / nxt around the symbol loop, and every copy it makes of
    // the two registers (one per match) then gets an s_waitcnt vmcnt(0) -- which also waits for the acknowledgement of
    // every store the wave has in flight, two memory round trips per match.  Waiting here costs one round trip per 64
    // dwords of input.
    __device__ uint32_t load(uint64_t w) const
    {
        const uint32_t* p = words + (w < nwords ? w : nwords - 1);
        uint32_t v;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    __device__ void refill()  // afterwards cnt > 32
    {
        if (cnt <= 32) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(widx & 63));
            buf |= (uint64_t)w << cnt;
            cnt += 32;
            ++widx;
            if ((widx & 63) == 0) {
                cur = nxt;
                nxt = load((uint64_t)widx + 64u + (uint64_t)lane);
            }
        }
    }
    __device__ void init(uint64_t byte_addr)
    {
        widx = (uint32_t)(byte_addr >> 2);
        const uint64_t cbase = (uint64_t)(widx & ~63u);
        cur = load(cbase + (uint64_t)lane);
        nxt = load(cbase + 64 + (uint64_t)lane);
        buf = 0;
        cnt = 0;
        refill();
        const int skip = (int)(byte_addr & 3) * 8;
        buf >>= skip;
        cnt -= skip;
        refill();
    }
    __device__ uint32_t bits(int n)  // n <= 16, cnt >= n
    {
        const uint32_t v = (uint32_t)buf & ((1u << n) - 1u);
        buf >>= n;
        cnt -= n;
        return v;
    }
    __device__ void drop(int n)
    {
        buf >>= n;
        cnt -= n;
    }
    __device__ uint64_t bit_pos() const { return (uint64_t)widx * 32 - (uint64_t)cnt; }  // of the next unread bit, from the file's start
};

// Canonical code of n symbols with code lengths lens[] (0 = unused): per-length tables, symbols in canonical order and
// the primary lookup table over the next TBITS bits.  zlib's rules (inflate_table): an over-subscribed set is an
// error, an incomplete one too unless it has no code longer than one bit and the alphabet may be incomplete.
template <int TBITS, int KIND>
__device__ __noinline__ uint32_t build_code(const uint8_t* lens, int n, uint16_t* table, uint16_t* sorted, CodeSet* cs, uint32_t* cnt,
                               bool may_be_incomplete, int lane)
{
    if (lane < 16) cnt[lane] = 0;
    WAVE_SYNC();
    for (int s = lane; s < n; s += 64) {
        const uint32_t L = lens[s];
        if (L) atomicAdd(&cnt[L], 1u);
    }
    WAVE_SYNC();
    int left = 1, maxlen = 0;
    bool over = false;
    uint32_t code = 0, off = 0, my_first = 0, my_offs = 0, my_count = 0;
    for (int L = 1; L <= 15; ++L) {
        const uint32_t c = uni(cnt[L]);
        code <<= 1;  // the first code of length L
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
        if (lane == L) {
            my_first = code;
            my_offs = off;
            my_count = c;
        }
        code += c;
        off += c;
        if (c) maxlen = L;
    }
    if (over) return INF_ERR_CODES;
    if (left > 0 && !(may_be_incomplete && maxlen <= 1)) return INF_ERR_CODES;
    if (lane >= 1 && lane < 16) {
        cs->first[lane] = (uint16_t)my_first;
        cs->count[lane] = (uint16_t)my_count;
        cs->offs[lane] = (uint16_t)my_offs;
    }
    // symbols in canonical order: by length, then by symbol.  Lane L keeps the next free slot of length L.
    uint32_t next_slot = my_offs;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t myL = s < n ? (uint32_t)lens[s] : 0u;
        for (int L = 1; L <= 15; ++L) {
            const uint64_t m = __ballot(myL == (uint32_t)L);
            if (!m) continue;
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)next_slot, L);
            if (myL == (uint32_t)L) sorted[base + (uint32_t)__popcll(m & below)] = (uint16_t)s;
            if (lane == L) next_slot += (uint32_t)__popcll(m);
        }
    }
    WAVE_SYNC();
    // the primary table: lane l fills entries l, l + 64, ...
    constexpr int PER = (1 << TBITS) / 64;
    uint32_t ent[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) ent[k] = kNotLiteral;  // a longer code, or none
    for (int L = 1; L <= TBITS; ++L) {
        const uint32_t c = uni(cs->count[L]);
        if (!c) continue;
        const uint32_t f = uni(cs->first[L]), o = uni(cs->offs[L]);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t e = (uint32_t)lane + 64u * (uint32_t)k;
            const uint32_t d = (__brev(e) >> (32 - L)) - f;  // the entry's first L bits as a code (first bit = MSB)
            if (d < c) {
                const uint32_t sym = sorted[o + d];
                if (KIND == CODE_LITLEN)
                    ent[k] = (uint32_t)L | (sym < 256 ? sym << 4 : kNotLiteral | (litlen_payload(sym) << 4));
                else if (KIND == CODE_DIST)
                    ent[k] = (uint32_t)L | (sym <= 29 ? dist_payload(sym) << 4 : kNotLiteral);
                else
                    ent[k] = (uint32_t)L | (sym << 4);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) table[lane + 64 * k] = (uint16_t)ent[k];
    WAVE_SYNC();
    return INF_OK;
}

// the code at the head of `buf` is longer than the primary table's bits: the canonical compare per length.
// (symbol << 4 | length), or 0 when the bits are no code
template <int TBITS>
__device__ __noinline__ uint32_t long_sym(uint32_t head, const uint16_t* sorted, const CodeSet* cs)
{
    const uint32_t rev = __brev(head);
    for (int L = TBITS + 1; L <= 15; ++L) {
        const uint32_t c = uni(cs->count[L]);
        if (!c) continue;
        const uint32_t d = (rev >> (32 - L)) - uni(cs->first[L]);
        if (d < c) return (uint32_t)L | (uni(sorted[uni(cs->offs[L]) + d]) << 4);
    }
    return 0;
}

// (symbol << 4 | length) of the code at the head of `buf` (>= 15 bits valid), or 0 when the bits are no code
template <int TBITS>
__device__ inline uint32_t decode_sym(uint64_t buf, const uint16_t* table, const uint16_t* sorted, const CodeSet* cs)
{
    const uint32_t e = uni(table[(uint32_t)buf & ((1u << TBITS) - 1u)]) & ~kNotLiteral;
    if (e & 15u) return e;
    return uni(long_sym<TBITS>((uint32_t)buf, sorted, cs));
}

// A member that cannot be inflated ends its wave on the spot (status word, s_endpgm): with plain returns out of the nested
// loops the compiler threads an exit code through every loop level, and the kernel is bound by instruction issue.  One
// asm statement that, as far as the compiler knows, comes back: a failing branch is then a plain uniform if-then.  (As
// `if (lane == 0) *status = code; __builtin_amdgcn_endpgm();`, or declared noreturn, every failing branch is an exit
// of the symbol loop, the compiler joins all exits of a function into one block, and the paths to it run through flag
// registers that the good path has to set -- a dozen scalar moves per match.)
__device__ inline void inflate_fail(uint32_t* status, uint32_t code, int lane)
{
    uint32_t z, c;
    asm volatile(
        "v_mov_b32 %[z], 0\n\t"
        "v_mov_b32 %[c], %[code]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[lane]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "global_store_dword %[z], %[c], %[st]\n\t"
        "s_endpgm"
        : [z] "=&v"(z), [c] "=&v"(c)
        : [code] "s"(code), [lane] "v"(lane), [st] "s"(status)
        : "vcc", "memory");
}
#define INF_FAIL(code) inflate_fail(status, (code), lane)

// One window of the literal decode, in two halves (see inflate_member for why they can be apart).
//   Speculative decode: lane k looks up the code that WOULD start k bits on; the wave then walks the chain of real code
// starts through those entries with v_readlane (no memory in the chain) and all literals of the window leave in one
// store.  One LDS access per ~5 literals instead of one each.
//   Both halves are hand-written (the kernel is bound by the issue of scalar instructions, and the compiler's version
// took about twice as many).  The chain walk, four instructions per literal: every offset visited gets its bit in
// `starts`; s_min both copies the next offset into the lane select (a VALU-written SGPR as lane select needs wait
// states, a SALU-written one none) and compares it with the window's end (SCC = the first operand is the smaller one);
// unrolled six times, so that the walk of a usual window takes no branch but the one out of it.  The walk ends behind
// the window (next <= 64 - 5) or at a non-literal (next = 64, its offset the last bit set, which is then taken out again
// and its entry read into e).
struct Window {
    uint64_t starts;  // the offsets literals start at
    uint32_t mine;    // lane k: the entry of the code that would start k bits on
    uint32_t next;    // the bits the window takes: up to the next window's first code, or up to the non-literal
    uint32_t n;       // literals
    uint32_t e;       // the non-literal's entry, 0 when the window ended on its length
};

__device__ __forceinline__ void window_walk(BitReader& br, const uint16_t* lit, int lane, Window& w)
{
    br.refill();
    const uint32_t K1 = (uint32_t)br.cnt - 14u;  // a code that starts at an offset < K1 has all its bits (and 5 to spare) in buf
    w.mine = lit[(uint32_t)(br.buf >> lane) & ((1u << LIT_BITS) - 1u)];
    // where the chain goes on from offset `lane`: behind the literal that starts there, or to 64 when the code there is
    // no literal (or no code of the primary table)
    const uint32_t onward = (w.mine & kNotLiteral) ? 64u : (uint32_t)lane + (w.mine & 15u);
    asm volatile(
        "s_mov_b64 %[st], 0\n\t"
        "s_mov_b32 %[k], 0\n"
        "1:\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_bitset1_b64 %[st], %[k]\n\t"
        "v_readlane_b32 %[t], %[onward], %[k]\n\t"
        "s_min_u32 %[k], %[t], %[K1]\n\t"
        "s_cbranch_scc1 1b\n"
        "3:\n\t"
        "s_mov_b32 %[e], 0\n\t"
        "s_cmp_lt_u32 %[t], 64\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_flbit_i32_b64 %[t], %[st]\n\t"
        "s_xor_b32 %[t], %[t], 63\n\t"
        "s_bitset0_b64 %[st], %[t]\n\t"
        "v_readlane_b32 %[e], %[mine], %[t]\n"
        "2:\n\t"
        "s_bcnt1_i32_b64 %[k], %[st]"
        : [st] "=&s"(w.starts), [k] "=&s"(w.n), [t] "=&s"(w.next), [e] "=&s"(w.e)
        : [onward] "v"(onward), [mine] "v"(w.mine), [K1] "s"(K1)
        : "scc");
}

// ... and the literals leave in one store: exec = starts, a lane's rank among the starts is v_mbcnt of exec, and a byte at
// or beyond out_len is not written (pos runs on and the member fails at its next match or at its end, as it does in
// zlib); last the bits walked over are dropped.
__device__ __forceinline__ void window_emit(BitReader& br, const Window& w, uint8_t* dst, uint32_t& pos, uint32_t out_len)
{
    uint64_t save;
    uint32_t vaddr, vdata;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_mov_b64 exec, %[st]\n\t"
        "v_mbcnt_lo_u32_b32 %[va], exec_lo, 0\n\t"
        "v_mbcnt_hi_u32_b32 %[va], exec_hi, %[va]\n\t"
        "v_add_u32_e32 %[va], %[pos], %[va]\n\t"
        "v_lshrrev_b32_e32 %[vd], 4, %[mine]\n\t"
        "v_cmp_gt_u32_e32 vcc, %[outlen], %[va]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "global_store_byte %[va], %[vd], %[dst]\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_add_u32 %[pos], %[pos], %[n]\n\t"
        "s_lshr_b64 %[buf], %[buf], %[t]\n\t"
        "s_sub_u32 %[cnt], %[cnt], %[t]"
        : [save] "=&s"(save), [va] "=&v"(vaddr), [vd] "=&v"(vdata), [pos] "+s"(pos), [buf] "+s"(br.buf), [cnt] "+s"(br.cnt)
        : [st] "s"(w.starts), [n] "s"(w.n), [t] "s"(w.next), [mine] "v"(w.mine), [outlen] "s"(out_len), [dst] "s"(dst)
        : "scc", "vcc", "memory");
}

__device__ uint32_t inflate_member(const uint32_t* __restrict__ file32, uint64_t file_words, const msq_bgzf_member mem,
                                   uint8_t* __restrict__ out, WaveLds& S, int lane, uint32_t* __restrict__ status)
{
    if (mem.out_len == 0) return INF_OK;  // the end-of-file marker and other empty members: nothing to write
    const uint8_t* file8 = reinterpret_cast<const uint8_t*>(file32);
    uint8_t* dst = out + mem.out_off;
    const uint32_t out_len = mem.out_len;
    const uint64_t end_bit = (mem.in_off + (uint64_t)mem.in_len) * 8;
    BitReader br;
    br.words = file32;
    br.nwords = file_words;
    br.lane = lane;
    br.init(mem.in_off);
    uint32_t pos = 0;  // bytes written
    bool fixed_ready = false;
    for (;;) {
        if (br.bit_pos() + 3 > end_bit) INF_FAIL(INF_ERR_INPUT);
        br.refill();
        const uint32_t bfinal = br.bits(1), btype = br.bits(2);
        if (btype == 3) INF_FAIL(INF_ERR_BTYPE);
        if (btype == 0) {
            br.drop(br.cnt & 7);  // to the byte boundary
            br.refill();
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if ((len ^ 0xffffu) != nlen) INF_FAIL(INF_ERR_STORED);
            const uint64_t byte = br.bit_pos() >> 3;
            if ((byte + len) * 8 > end_bit) INF_FAIL(INF_ERR_INPUT);
            if (pos + len > out_len) INF_FAIL(INF_ERR_OUTPUT);
            for (uint32_t i = (uint32_t)lane; i < len; i += 64) dst[pos + i] = file8[byte + i];
            pos += len;
            br.init(byte + len);
        } else {
            int n_lit, n_dist;
            if (btype == 1) {
                n_lit = 288;
                n_dist = 32;
                if (!fixed_ready) {
                    for (int s = lane; s < 320; s += 64) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
                    WAVE_SYNC();
                }
            } else {
                br.refill();
                n_lit = (int)br.bits(5) + 257;
                n_dist = (int)br.bits(5) + 1;
                const int n_cl = (int)br.bits(4) + 4;
                if (n_lit > 286 || n_dist > 30) INF_FAIL(INF_ERR_COUNTS);
                if (lane < 19) S.cl_lens[lane] = 0;
                WAVE_SYNC();
                for (int i = 0; i < n_cl; ++i) {
                    br.refill();
                    const uint32_t v = br.bits(3);
                    if (lane == 0) S.cl_lens[kClOrder[i]] = (uint8_t)v;
                }
                WAVE_SYNC();
                if (const uint32_t e = uni(build_code<CL_BITS, CODE_PLAIN>(S.cl_lens, 19, S.cl, S.sorted_cl, &S.cs_cl, S.cnt, false, lane))) INF_FAIL(e);
                const int total = n_lit + n_dist;
                int i = 0;
                uint32_t prev = 0;
                while (i < total) {
                    if (br.bit_pos() > end_bit) INF_FAIL(INF_ERR_INPUT);
                    br.refill();
                    const uint32_t e = decode_sym<CL_BITS>(br.buf, S.cl, S.sorted_cl, &S.cs_cl);
                    if (!e) INF_FAIL(INF_ERR_SYMBOL);
                    br.drop((int)(e & 15u));
                    const uint32_t sym = e >> 4;
                    if (sym < 16) {
                        if (lane == 0) S.lens[i] = (uint8_t)sym;
                        prev = sym;
                        ++i;
                        continue;
                    }
                    uint32_t v = 0;
                    int rep;
                    if (sym == 16) {
                        if (i == 0) INF_FAIL(INF_ERR_REPEAT);
                        v = prev;
                        rep = 3 + (int)br.bits(2);
                    } else if (sym == 17) {
                        rep = 3 + (int)br.bits(3);
                        prev = 0;
                    } else {
                        rep = 11 + (int)br.bits(7);
                        prev = 0;
                    }
                    if (i + rep > total) INF_FAIL(INF_ERR_REPEAT);
                    for (int k = lane; k < rep; k += 64) S.lens[i + k] = (uint8_t)v;
                    i += rep;
                }
                WAVE_SYNC();
                if (uni(S.lens[256]) == 0) INF_FAIL(INF_ERR_NOEOB);
            }
            if (btype == 2 || !fixed_ready) {
                if (const uint32_t e = uni(build_code<LIT_BITS, CODE_LITLEN>(S.lens, n_lit, S.lit, S.sorted_lit, &S.cs_lit, S.cnt, true, lane))) INF_FAIL(e);
                if (const uint32_t e = uni(build_code<DIST_BITS, CODE_DIST>(S.lens + n_lit, n_dist, S.dist, S.sorted_dist, &S.cs_dist, S.cnt, true, lane))) INF_FAIL(e);
                fixed_ready = btype == 1;
            }
            // The symbol loop: windows of literals until one ends at a non-literal (w.e), then that length / distance
            // pair.  A match of up to 64 bytes (nearly all are) loads its source bytes, and while they are on their way
            // the NEXT window is looked up and walked; then the match's bytes are stored, then the window's literals
            // (behind the match: other addresses).  Alone in its SIMD a wave spends most of its time in the matches'
            // memory round trips; this puts ~60 instructions into each.  `have`: w is that next window, emitted already.
            bool have = false;
            Window w;
            for (;;) {
                if (!have) {
                    for (;;) {  // a run of literals
                        window_walk(br, S.lit, lane, w);
                        window_emit(br, w, dst, pos, out_len);
                        if (w.e) break;
                    }
                }
                have = false;
                uint32_t e = w.e;
                br.refill();  // the window may have left as few as 15 bits: the code below and its extra bits take up to 20
                uint32_t cl = e & 15u;
                if (cl == 0) {  // a code longer than the primary table's bits (or no code at all)
                    const uint32_t raw = uni(long_sym<LIT_BITS>((uint32_t)br.buf, S.sorted_lit, &S.cs_lit));
                    if (!raw) INF_FAIL(INF_ERR_SYMBOL);
                    cl = raw & 15u;
                    const uint32_t sym = raw >> 4;
                    if (sym < 256) {  // a literal with a long code
                        br.drop((int)cl);
                        if (pos >= out_len) INF_FAIL(INF_ERR_OUTPUT);
                        dst[pos] = (uint8_t)sym;  // every lane, the same byte: no divergent branch (see below)
                        ++pos;
                        continue;
                    }
                    e = litlen_payload(sym) << 4;
                }
                br.drop((int)cl);
                const int xb = (int)((e >> 4) & 7u);
                if (xb >= (int)kLenEob) {
                    if (xb == (int)kLenInvalid) INF_FAIL(INF_ERR_SYMBOL);
                    break;  // end of block
                }
                const uint32_t len = ((e >> 7) & 0xffu) + 3u + br.bits(xb);
                br.refill();
                uint32_t d = uni(S.dist[(uint32_t)br.buf & ((1u << DIST_BITS) - 1u)]);
                uint32_t dcl = d & 15u;
                if (d & kNotLiteral) {
                    if (dcl) INF_FAIL(INF_ERR_DIST);  // symbols 30 / 31
                    const uint32_t raw = uni(long_sym<DIST_BITS>((uint32_t)br.buf, S.sorted_dist, &S.cs_dist));
                    if (!raw) INF_FAIL(INF_ERR_SYMBOL);
                    if ((raw >> 4) > 29) INF_FAIL(INF_ERR_DIST);
                    dcl = raw & 15u;
                    d = dist_payload(raw >> 4) << 4;
                }
                br.drop((int)dcl);
                const int dxb = (int)((d >> 4) & 15u);
                const uint32_t dist = (((d >> 8) & 3u) << dxb) + 1u + br.bits(dxb);
                if (dist > pos) INF_FAIL(INF_ERR_DIST);
                if (pos + len > out_len) INF_FAIL(INF_ERR_OUTPUT);
                // The copy: the wave reads what it has stored, and a wave's vector memory operations are performed in order.
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, INF_FENCE_SCOPE);
                const uint8_t* src = dst + (pos - dist);
                // No branch on the lane anywhere (the lanes beyond the match repeat its last byte): with any divergent
                // branch in the symbol loop the compiler structurizes the whole loop and joins its paths through flag
                // registers, a dozen scalar moves per match.
                const uint32_t last = len - 1u;
                if (len <= 64) {
                    uint32_t i = min((uint32_t)lane, last);
                    const uint32_t at = pos + i;
                    if (dist < len) i %= dist;  // the match overlaps itself: its bytes repeat with period dist, all of them already written
                    const uint8_t byte = src[i];
                    pos += len;
                    window_walk(br, S.lit, lane, w);
                    dst[at] = byte;
                    window_emit(br, w, dst, pos, out_len);
                    have = w.e != 0;
                } else {
                    if (dist >= len) {
                        for (uint32_t b0 = 0; b0 < len; b0 += 64) {
                            const uint32_t i = min(b0 + (uint32_t)lane, last);
                            dst[pos + i] = src[i];
                        }
                    } else {
                        for (uint32_t b0 = 0; b0 < len; b0 += 64) {
                            const uint32_t i = min(b0 + (uint32_t)lane, last);
                            dst[pos + i] = src[i % dist];
                        }
                    }
                    pos += len;
                }
            }
            if (br.bit_pos() > end_bit) INF_FAIL(INF_ERR_INPUT);  // checked per block: its symbols are bounded by ISIZE either way
        }
        if (bfinal) break;
    }
    if (pos != out_len) INF_FAIL(INF_ERR_SHORT);
    return INF_OK;
}

#ifdef INF_WAVES
__attribute__((amdgpu_waves_per_eu(INF_WAVES, INF_WAVES)))
#endif
__global__ __launch_bounds__(256) void bgzf_inflate_kernel(const uint32_t* __restrict__ file32, uint64_t file_words,
                                                           const msq_bgzf_member* __restrict__ members, uint32_t n_members,
                                                           uint8_t* __restrict__ out, uint32_t* __restrict__ status)
{
    __shared__ WaveLds lds[4];
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t m = blockIdx.x * 4u + wave;
    if (m >= n_members) return;
    const uint32_t err = inflate_member(file32, file_words, members[m], out, lds[wave], lane, status + m);
    if (lane == 0) status[m] = err;
}

}  // namespace

static void free_bam_dev(msq_bam_dev* b)
{
    if (!b) return;
    for (void* p : {(void*)b->d_data, (void*)b->d_rec_off, (void*)b->d_rec, (void*)b->d_kept})
        if (p) (void)hipFree(p);
    delete b;
}

// forget a staged image: its copy thread joined, its events destroyed (msq_ctx_destroy calls it too)
static constexpr uint64_t kSlot = 8ull << 20;  // the upload ring (msq_bgzf_stage_file): pinned slots of 8 MB, eight of them
static constexpr int kSlots = 8;

void msq_bgzf_drop_staged(msq_ctx* ctx)
{
    if (ctx->stage_thread) {
        if (ctx->stage_thread->joinable()) ctx->stage_thread->join();
        delete ctx->stage_thread;
        ctx->stage_thread = nullptr;
    }
    if (ctx->staged_total && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (int k = 0; k < ctx->staged_total; ++k)
        if (ctx->staged_event[k]) (void)hipEventDestroy(ctx->staged_event[k]);
    ctx->staged_total = 0;
    ctx->staged_done.store(0);
    ctx->staged_err.store(0);
    ctx->staged_file = nullptr;
    ctx->staged_bytes = 0;
}

// The file image on its way into the context's scratch: pieces on the copy stream, an event behind each.  The copies are
// fed by threads of the library's own and this call returns at once: the caller scans the BGZF headers meanwhile, and
// msq_bgzf_inflate_device launches every piece's kernel behind that piece's event (it stages the image itself when
// nobody has).  Three pieces (4 : 3 : 2): a kernel takes ~5 ms however few members it has (one member is one wave,
// ~8000 dependent windows and 6400 matches with a memory round trip each) and 1.3 us per member once the chip's issue
// slots are full (10^5 reads: 9.2 ms), so a piece's kernel should be running while the next piece arrives; the pieces'
// kernels run side by side on three streams (tools/bam_timeline.sh; tools/bam_share_ab.sh: two to four pieces are
// within 0.5 ms of each other, one piece is 2 ms behind).
extern "C" int msq_bgzf_stage_file(msq_ctx* ctx, const uint8_t* file, uint64_t file_bytes)
{
    if (!ctx || (file_bytes && !file)) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_stage_file: NULL argument");
    MSQ_HIP(hipSetDevice(ctx->device));
    msq_bgzf_drop_staged(ctx);
    if (file_bytes == 0) return MSQ_OK;  // nothing to stage: forgets an image staged before (a caller that gave up on its file)
    if (file_bytes >= (1ull << 34) - 4096) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_stage_file: a file image of %llu bytes (the device front takes less than 16 GiB at once)", (unsigned long long)file_bytes);
    const uint64_t file_padded = ((file_bytes + 3) & ~3ull) + 1024;  // the bit reader looks two chunks ahead
    void* scratch = nullptr;
    if (int rc = msq_scratch(ctx, file_padded + 256, &scratch)) return rc;
    MSQ_HIP(hipStreamSynchronize(ctx->stream));  // the scratch's previous user is done
    if (!ctx->copy_stream) MSQ_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    char* const d_file = (char*)scratch;
    MSQ_HIP(hipMemsetAsync(d_file + (file_bytes & ~3ull), 0, file_padded - (file_bytes & ~3ull), ctx->copy_stream));
    MSQ_HIP(hipStreamSynchronize(ctx->copy_stream));  // before the last piece's copy writes the bytes beside the padding
    const int n_pieces = file_bytes >= ctx->bam_piece_min_bytes ? (int)std::min<uint64_t>(8, std::max<uint64_t>(1, ctx->bam_pieces)) : 1;
    // Pieces get smaller towards the end (two: the first share; more: sizes n + 1, n, ..., 2 in n (n + 3) / 2 parts): the
    // members of a piece are inflated behind its copy, and what remains to do once the last byte has arrived is the last
    // piece's kernel -- which, however few its members, takes as long as one wave needs for one member.
    uint64_t parts = 0, parts_all = (uint64_t)n_pieces * (uint64_t)(n_pieces + 3) / 2;
    for (int k = 0; k < n_pieces; ++k) {
        parts += (uint64_t)(n_pieces + 1 - k);
        ctx->staged_upto[k] = k + 1 == n_pieces ? file_bytes
                              : n_pieces == 2   ? file_bytes / 100 * ctx->bam_first_share_pct
                                                : file_bytes / parts_all * parts;
        ctx->staged_event[k] = nullptr;
        hipError_t e = hipEventCreateWithFlags(&ctx->staged_event[k], hipEventDisableTiming);
        ctx->staged_total = k + 1;
        if (e != hipSuccess) {
            msq_bgzf_drop_staged(ctx);
            return msq_fail(MSQ_ERR_HIP, "msq_bgzf_stage_file: %s", hipGetErrorString(e));
        }
    }
    ctx->staged_file = file;
    ctx->staged_bytes = file_bytes;
    // The upload.  hipMemcpyAsync from a pageable mapping moves ~17 GB/s here (one thread of the runtime copies into its
    // staging buffers); bam_upload_threads threads copying slices of the mapping into a ring of pinned slots, each slot
    // crossing by DMA as soon as it is full, move ~50 GB/s (tools/probe/h2d_probe.cpp: 281 MB in 5.6 instead of 15 ms).
    int workers = file_bytes >= (1u << 20) ? (int)ctx->bam_upload_threads : 0;  // a small file: not worth the threads
    if (workers > 0 && !ctx->pinned_ring) {
        if (hipHostMalloc(&ctx->pinned_ring, kSlot * kSlots, hipHostMallocDefault) == hipSuccess)
            ctx->pinned_ring_bytes = kSlot * kSlots;
        else {
            (void)hipGetLastError();
            ctx->pinned_ring = nullptr;
            workers = 0;  // no pinned memory to be had: the plain copy
        }
    }
    if (workers > 0)  // a piece ends where a slot does
        for (int k = 0; k + 1 < n_pieces; ++k) ctx->staged_upto[k] = ctx->staged_upto[k] / kSlot * kSlot;
    char* const ring = (char*)ctx->pinned_ring;
    auto upload = [ctx, file, file_bytes, d_file, n_pieces, workers, ring] {
        hipError_t e = hipSetDevice(ctx->device);
        auto give_up = [&](hipError_t err) {
            ctx->staged_err.store((int)err);
            ctx->staged_done.store(n_pieces, std::memory_order_release);  // nobody waits for a piece that will not come
        };
        if (e != hipSuccess) return give_up(e);
        if (workers <= 0) {
            uint64_t copied = 0;
            for (int k = 0; k < n_pieces; ++k) {
                const uint64_t upto = ctx->staged_upto[k];
                if (upto > copied) e = hipMemcpyAsync(d_file + copied, file + copied, upto - copied, hipMemcpyHostToDevice, ctx->copy_stream);
                if (e == hipSuccess) e = hipEventRecord(ctx->staged_event[k], ctx->copy_stream);
                if (e != hipSuccess) return give_up(e);
                copied = upto;
                ctx->staged_done.store(k + 1, std::memory_order_release);
            }
            return;
        }
        // Worker w copies the w-th part of every chunk; a chunk is ready when every worker has done its part, and may be
        // filled once the chunk kSlots before it has crossed (fill_limit).  This thread issues the DMAs and the pieces'
        // events.  `stop` ends the workers early when a HIP call has failed.
        const uint64_t n_chunks = (file_bytes + kSlot - 1) / kSlot;
        std::unique_ptr<std::atomic<int>[]> ready(new (std::nothrow) std::atomic<int>[n_chunks]);
        if (!ready) return give_up(hipErrorOutOfMemory);
        for (uint64_t c = 0; c < n_chunks; ++c) ready[c].store(0);
        std::atomic<uint64_t> fill_limit{(uint64_t)kSlots};
        std::atomic<bool> stop{false};
        hipEvent_t slot_event[kSlots] = {};
        for (hipEvent_t& ev : slot_event)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        std::vector<std::thread> pool;
        if (e == hipSuccess) {
            try {
                for (int w = 0; w < workers; ++w)
                    pool.emplace_back([&, w] {
                        for (uint64_t c = 0; c < n_chunks; ++c) {
                            while (c >= fill_limit.load(std::memory_order_acquire)) {
                                if (stop.load(std::memory_order_relaxed)) return;
                                std::this_thread::yield();
                            }
                            const uint64_t lo = c * kSlot, len = std::min(kSlot, file_bytes - lo);
                            const uint64_t a = len * (uint64_t)w / (uint64_t)workers, b = len * (uint64_t)(w + 1) / (uint64_t)workers;
                            std::memcpy(ring + (c % kSlots) * kSlot + a, file + lo + a, b - a);
                            ready[c].fetch_add(1, std::memory_order_release);
                        }
                    });
            } catch (...) {
                e = hipErrorOutOfMemory;
            }
        }
        const int started = (int)pool.size();
        uint64_t oldest = 0, n_issued = 0;
        auto poll = [&] {  // a failed copy (not "not ready yet") ends the upload: nobody may wait for a slot that will not come free
            while (oldest < n_issued) {
                const hipError_t q = hipEventQuery(slot_event[oldest % kSlots]);
                if (q != hipSuccess) {
                    if (q != hipErrorNotReady) e = q;
                    (void)hipGetLastError();
                    break;
                }
                ++oldest;
                fill_limit.store(oldest + kSlots, std::memory_order_release);
            }
        };
        int piece = 0;
        for (uint64_t c = 0; e == hipSuccess && started == workers && c < n_chunks; ++c) {
            while (e == hipSuccess && ready[c].load(std::memory_order_acquire) < workers) {
                poll();
                std::this_thread::yield();
            }
            if (e != hipSuccess) break;
            const uint64_t lo = c * kSlot, len = std::min(kSlot, file_bytes - lo);
            e = hipMemcpyAsync(d_file + lo, ring + (c % kSlots) * kSlot, len, hipMemcpyHostToDevice, ctx->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(slot_event[c % kSlots], ctx->copy_stream);
            n_issued = c + 1;
            while (e == hipSuccess && piece < n_pieces && ctx->staged_upto[piece] <= lo + len) {
                e = hipEventRecord(ctx->staged_event[piece], ctx->copy_stream);
                if (e == hipSuccess) ctx->staged_done.store(++piece, std::memory_order_release);
            }
            poll();
        }
        stop.store(true);
        for (std::thread& t : pool) t.join();
        // the slots are this call's until their copies have left them; an empty file's pieces still get their events
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
        while (e == hipSuccess && started == workers && piece < n_pieces) {
            e = hipEventRecord(ctx->staged_event[piece], ctx->copy_stream);
            if (e == hipSuccess) ctx->staged_done.store(++piece, std::memory_order_release);
        }
        for (hipEvent_t ev : slot_event)
            if (ev) (void)hipEventDestroy(ev);
        if (e != hipSuccess || started != workers) give_up(e != hipSuccess ? e : hipErrorOutOfMemory);
    };
    try {
        ctx->stage_thread = new std::thread(upload);
    } catch (...) {  // no memory for the object, or no thread to be had (std::system_error)
        ctx->stage_thread = nullptr;
    }
    if (!ctx->stage_thread) {
        msq_bgzf_drop_staged(ctx);
        return msq_fail(MSQ_ERR_NOMEM, "msq_bgzf_stage_file: cannot start the copy thread");
    }
    return MSQ_OK;
}

extern "C" int msq_bgzf_inflate_device(msq_ctx* ctx, const uint8_t* file, uint64_t file_bytes, const msq_bgzf_member* members,
                                       uint64_t n_members, msq_bam_dev** out)
{
    if (!ctx || !out) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_inflate_device: NULL argument");
    *out = nullptr;
    if ((file_bytes && !file) || (n_members && !members)) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_inflate_device: NULL array");
    if (n_members > 0xfffffff0ull) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_inflate_device: too many members");
    uint64_t total = 0;
    bool ordered = true;  // members in file order (a BGZF file's are): a piece's members are a range
    for (uint64_t m = 0; m < n_members; ++m) {
        const msq_bgzf_member& k = members[m];
        if (k.in_off > file_bytes || k.in_len > file_bytes - k.in_off) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_inflate_device: member %llu lies outside the file", (unsigned long long)m);
        // SAMv1 4.1: a BGZF block inflates to at most 64 KiB; this bounds every loop of the kernel
        if (k.out_len > 65536u) return msq_fail(MSQ_ERR_DATA, "BGZF block %llu claims %u bytes (more than 64 KiB)", (unsigned long long)m, k.out_len);
        if (k.out_off != total) return msq_fail(MSQ_ERR_ARG, "msq_bgzf_inflate_device: member %llu does not follow its predecessor in the output", (unsigned long long)m);
        total += k.out_len;
        if (m && (k.in_off < members[m - 1].in_off || k.in_off + k.in_len < members[m - 1].in_off + members[m - 1].in_len)) ordered = false;
    }
    MSQ_HIP(hipSetDevice(ctx->device));
    const bool trace = getenv("MSQ_BAM_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&t0]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const bool was_staged = ctx->staged_total > 0 && ctx->staged_file == file && ctx->staged_bytes == file_bytes;
    if (!was_staged)
        if (int rc = msq_bgzf_stage_file(ctx, file, file_bytes)) return rc;
    if (!ctx->copy_stream) MSQ_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (hipStream_t& st : ctx->side_stream)
        if (!st) MSQ_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (hipEvent_t& ev : ctx->side_event)
        if (!ev) MSQ_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const double t_staged = ms_since();
    msq_bam_dev* b = new (std::nothrow) msq_bam_dev;
    if (!b) {
        msq_bgzf_drop_staged(ctx);
        return msq_fail(MSQ_ERR_NOMEM, "msq_bgzf_inflate_device: out of memory");
    }
    auto fail = [&](int code, const char* what, hipError_t e) {
        for (hipStream_t st : ctx->side_stream) (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(ctx->stream);
        msq_bgzf_drop_staged(ctx);
        free_bam_dev(b);
        return msq_fail(code, "msq_bgzf_inflate_device: %s: %s", what, hipGetErrorString(e));
    };
    // the file image sits in the context's scratch (msq_bgzf_stage_file); the member table and the status words in a
    // buffer of the context's own that grows on demand; the inflated stream is the caller's to keep
    const uint64_t file_padded = ((file_bytes + 3) & ~3ull) + 1024;
    const uint64_t table_bytes = (std::max<uint64_t>(n_members, 1) * sizeof(msq_bgzf_member) + 255) & ~255ull;
    const uint64_t status_bytes = (std::max<uint64_t>(n_members, 1) * 4 + 255) & ~255ull;
    hipError_t e = hipSuccess;
    if (ctx->bam_tables_bytes < table_bytes + status_bytes) {
        if (ctx->d_bam_tables) (void)hipFree(ctx->d_bam_tables);
        ctx->d_bam_tables = nullptr;
        ctx->bam_tables_bytes = 0;
        const size_t want = (size_t)(table_bytes + status_bytes) * 2;
        e = hipMalloc(&ctx->d_bam_tables, want);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "hipMalloc", e);
        ctx->bam_tables_bytes = want;
    }
    char* const d_file = (char*)ctx->d_scratch;
    char* const d_members = (char*)ctx->d_bam_tables;
    char* const d_status = d_members + table_bytes;
    e = hipMalloc((void**)&b->d_data, total + MSQ_BAM_DATA_PAD);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MSQ_ERR_NOMEM : MSQ_ERR_HIP, "hipMalloc", e);
    b->data_bytes = total;
    std::vector<uint32_t> status;
    try {
        status.resize(n_members);
    } catch (const std::bad_alloc&) {
        return fail(MSQ_ERR_NOMEM, "status", hipErrorOutOfMemory);
    }
    e = hipMemsetAsync(b->d_data + total, 0, MSQ_BAM_DATA_PAD, ctx->stream);
    if (e == hipSuccess && n_members) e = hipMemcpyAsync(d_members, members, n_members * sizeof(msq_bgzf_member), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0xFF, status_bytes, ctx->stream);
    // every piece's members behind its copy, the pieces' kernels in turn on the context's stream and the two side streams.
    // (A wave may look up to two chunks past its member into bytes that have not arrived: bits no valid stream consumes.)
    if (e == hipSuccess) e = hipEventRecord(ctx->side_event[0], ctx->stream);  // the tables are in place
    for (hipStream_t st : ctx->side_stream)
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->side_event[0], 0);
    bool side_used[2] = {false, false};
    uint64_t m0 = 0;
    for (int k = 0; e == hipSuccess && k < ctx->staged_total; ++k) {
        const bool last = k + 1 == ctx->staged_total;
        uint64_t m1 = m0;
        if (last || !ordered)
            m1 = last ? n_members : m0;
        else
            while (m1 < n_members && members[m1].in_off + members[m1].in_len <= ctx->staged_upto[k]) ++m1;
        while (ctx->staged_done.load(std::memory_order_acquire) <= k) usleep(20);  // the copy thread has recorded this piece's event
        if (ctx->staged_err.load()) {
            e = (hipError_t)ctx->staged_err.load();
            break;
        }
        hipStream_t on = k % 3 ? ctx->side_stream[k % 3 - 1] : ctx->stream;
        e = hipStreamWaitEvent(on, ctx->staged_event[k], 0);
        if (e == hipSuccess && m1 > m0) {
            if (k % 3) side_used[k % 3 - 1] = true;
            hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((unsigned)((m1 - m0 + 3) / 4)), dim3(256), 0, on, (const uint32_t*)d_file, file_padded / 4,
                               (const msq_bgzf_member*)d_members + m0, (uint32_t)(m1 - m0), b->d_data, (uint32_t*)d_status + m0);
            e = hipGetLastError();
        }
        m0 = m1;
    }
    for (int i = 0; i < 2; ++i) {
        if (e == hipSuccess && side_used[i]) e = hipEventRecord(ctx->side_event[1 + i], ctx->side_stream[i]);
        if (e == hipSuccess && side_used[i]) e = hipStreamWaitEvent(ctx->stream, ctx->side_event[1 + i], 0);
    }
    const double t_issued = ms_since();
    if (e == hipSuccess && n_members) e = hipMemcpyAsync(status.data(), d_status, n_members * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(MSQ_ERR_HIP, "inflate kernel", e);
    msq_bgzf_drop_staged(ctx);
    if (trace)
        std::fprintf(stderr, "msq_bgzf_inflate_device: %llu members, %.1f MB -> %.1f MB: %s at %.2f ms, launches issued at %.2f ms, done at %.2f ms\n",
                     (unsigned long long)n_members, file_bytes / 1e6, total / 1e6, was_staged ? "image staged by the caller, taken" : "image staged", t_staged, t_issued, ms_since());
    for (uint64_t m = 0; m < n_members; ++m)
        if (status[m] != INF_OK) {
            free_bam_dev(b);
            // std::runtime_error("BGZF block does not inflate to its ISIZE") of the host front (cpp/msq_bam.hpp)
            return msq_fail(MSQ_ERR_DATA, "BGZF block does not inflate to its ISIZE (member %llu, code %u)", (unsigned long long)m, status[m]);
        }
    *out = b;
    return MSQ_OK;
}

extern "C" uint64_t msq_bam_dev_bytes(const msq_bam_dev* b) { return b ? b->data_bytes : 0; }

extern "C" int msq_bam_dev_read(msq_ctx* ctx, const msq_bam_dev* b, uint64_t off, uint64_t len, void* host_dst)
{
    if (!ctx || !b || (len && !host_dst)) return msq_fail(MSQ_ERR_ARG, "msq_bam_dev_read: NULL argument");
    if (off > b->data_bytes || len > b->data_bytes - off) return msq_fail(MSQ_ERR_ARG, "msq_bam_dev_read: range outside the stream");
    MSQ_HIP(hipSetDevice(ctx->device));
    if (len) MSQ_HIP(hipMemcpyAsync(host_dst, b->d_data + off, len, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_bam_dev_free(msq_ctx* ctx, msq_bam_dev* b)
{
    if (!b) return MSQ_OK;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    free_bam_dev(b);
    return MSQ_OK;
}
