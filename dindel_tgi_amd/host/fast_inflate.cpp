// fast_inflate.cpp — see fast_inflate.hpp.
#include "fast_inflate.hpp"
#include <cstring>

namespace dindel {

namespace {

const int kLitBits = 10, kDistBits = 8;            // primary table widths; longer codes go through a second lookup
const int kMaxCodeLen = 15;

// One table entry: bits 0-3 = code length consumed by this lookup (0 = invalid), bits 4-5 = kind, bits 8-... = value.
//   kind 0: literal (value = byte) / distance symbol (value = symbol)      kind 1: length symbol (value = symbol - 257)
//   kind 2: end of block                                                   kind 3: go to sub-table (value = offset, length field = its index bits)
typedef uint32_t Entry;
inline Entry makeEntry(unsigned len, unsigned kind, unsigned value) { return len | (kind << 4) | (value << 8); }

struct Tables {
    Entry lit[(1 << kLitBits) + 640];              // primary + sub-tables (a code set that needs more is left to zlib)
    Entry dist[(1 << kDistBits) + 1024];
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline unsigned reverseBits(unsigned code, int len)
{
    unsigned r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Canonical Huffman code (RFC 1951 3.2.2) of `n` symbols with the given lengths -> lookup table indexed by the next input bits
// (LSB first).  false: over-subscribed, or incomplete in a way the format does not allow.
enum CodeKind { LITLEN, DISTANCE, CODELEN };
bool buildTable(const uint8_t *lens, int n, int primaryBits, Entry *table, int tableCap, CodeKind kind)
{
    const bool isLit = kind == LITLEN;
    int count[kMaxCodeLen + 1] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    if (count[0] == n) {                            // no code at all: legal for the distance code of a block without matches
        for (int i = 0; i < (1 << primaryBits); i++) table[i] = 0;
        return kind == DISTANCE;
    }
    int left = 1;
    for (int l = 1; l <= kMaxCodeLen; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
    if (left > 0) {                                 // incomplete: only the distance tree of ONE code of ONE bit is allowed (3.2.7; zlib:
        if (kind != DISTANCE || n - count[0] != 1 || count[1] != 1) return false;     // `left > 0 && max != 1` is an error) — anything else goes to zlib
    }
    unsigned next[kMaxCodeLen + 2];
    unsigned code = 0;
    count[0] = 0;
    for (int l = 1; l <= kMaxCodeLen; l++) { code = (code + unsigned(count[l - 1])) << 1; next[l] = code; }
    const int primarySize = 1 << primaryBits;
    for (int i = 0; i < primarySize; i++) table[i] = 0;
    // sub-tables: one per distinct primary prefix of the long codes, sized for the longest code under that prefix
    int subBits[1 << 10];
    for (int i = 0; i < primarySize; i++) subBits[i] = 0;
    {
        unsigned nx[kMaxCodeLen + 2];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (l <= primaryBits) { if (l) nx[l]++; continue; }
            const unsigned rev = reverseBits(nx[l]++, l);
            const int prefix = int(rev & unsigned(primarySize - 1));
            if (l - primaryBits > subBits[prefix]) subBits[prefix] = l - primaryBits;
        }
    }
    int used = primarySize;
    for (int i = 0; i < primarySize; i++) if (subBits[i]) {
        if (used + (1 << subBits[i]) > tableCap) return false;
        table[i] = makeEntry(unsigned(subBits[i]), 3, unsigned(used));
        for (int k = 0; k < (1 << subBits[i]); k++) table[used + k] = 0;
        used += 1 << subBits[i];
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned rev = reverseBits(next[l]++, l);
        Entry e;
        if (!isLit) e = makeEntry(0, 0, unsigned(s));
        else if (s < 256) e = makeEntry(0, 0, unsigned(s));
        else if (s == 256) e = makeEntry(0, 2, 0);
        else if (s > 285) continue;                 // 286 / 287 take part in the fixed code but never occur: their slots stay invalid
        else e = makeEntry(0, 1, unsigned(s - 257));
        if (l <= primaryBits) {
            e |= unsigned(l);
            for (unsigned idx = rev; idx < unsigned(primarySize); idx += 1u << l) table[idx] = e;
        } else {
            const int prefix = int(rev & unsigned(primarySize - 1));
            const int sb = subBits[prefix], base = int(table[prefix] >> 8);
            e |= unsigned(l - primaryBits);
            for (unsigned idx = rev >> primaryBits; idx < (1u << sb); idx += 1u << (l - primaryBits)) table[base + int(idx)] = e;
        }
    }
    return true;
}

struct BitReader {
    const uint8_t *p, *end;
    uint64_t buf;
    int cnt;                                        // valid bits in buf
    bool overrun;
    BitReader(const uint8_t *in, size_t n) : p(in), end(in + n), buf(0), cnt(0), overrun(false) {}
    inline void refill()                            // afterwards cnt >= 56 unless the input is exhausted
    {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);                       // little-endian host (x86-64)
            buf |= w << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= uint64_t(*p++) << cnt; cnt += 8; }
        }
    }
    inline unsigned peek(int n) const { return unsigned(buf & ((uint64_t(1) << n) - 1)); }
    inline void drop(int n) { if (n > cnt) { overrun = true; cnt = 0; buf = 0; return; } buf >>= n; cnt -= n; }
    inline unsigned take(int n) { const unsigned v = peek(n); drop(n); return v; }
};

} // namespace

bool fastInflate(const uint8_t *in, size_t inLen, uint8_t *out, size_t outLen)
{
    BitReader br(in, inLen);
    uint8_t *o = out, *const oend = out + outLen;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Tables T;
    bool fixedBuilt = false;
    Tables F;                                        // the fixed code's tables (3.2.6), built on first use
    for (;;) {
        br.refill();
        const unsigned final = br.take(1), type = br.take(2);
        if (br.overrun) return false;
        if (type == 0) {                             // stored block
            br.drop(br.cnt & 7);                     // to the next byte boundary
            // give back the whole bytes still in the bit buffer
            const uint8_t *q = br.p - (br.cnt >> 3);
            if (br.end - q < 4) return false;
            const unsigned len = unsigned(q[0]) | (unsigned(q[1]) << 8), nlen = unsigned(q[2]) | (unsigned(q[3]) << 8);
            if ((len ^ 0xFFFFu) != nlen) return false;
            q += 4;
            if (size_t(br.end - q) < len || size_t(oend - o) < len) return false;
            memcpy(o, q, len);
            o += len;
            br.p = q + len; br.buf = 0; br.cnt = 0;
        } else if (type == 1 || type == 2) {
            const Tables *tab;
            if (type == 1) {
                if (!fixedBuilt) {
                    uint8_t l[288];
                    for (int i = 0; i < 144; i++) l[i] = 8;
                    for (int i = 144; i < 256; i++) l[i] = 9;
                    for (int i = 256; i < 280; i++) l[i] = 7;
                    for (int i = 280; i < 288; i++) l[i] = 8;
                    uint8_t d[32];                                  // 32 five-bit codes; 30 and 31 never occur (3.2.6) and are rejected below
                    for (int i = 0; i < 32; i++) d[i] = 5;
                    if (!buildTable(l, 288, kLitBits, F.lit, int(sizeof(F.lit) / sizeof(Entry)), LITLEN)) return false;
                    if (!buildTable(d, 32, kDistBits, F.dist, int(sizeof(F.dist) / sizeof(Entry)), DISTANCE)) return false;
                    fixedBuilt = true;
                }
                tab = &F;
            } else {
                const unsigned hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
                if (br.overrun || hlit > 286 || hdist > 30) return false;
                uint8_t cl[19];
                memset(cl, 0, sizeof(cl));
                for (unsigned i = 0; i < hclen; i++) { br.refill(); cl[order[i]] = uint8_t(br.take(3)); }
                if (br.overrun) return false;
                Entry clt[(1 << 7) + 8];
                if (!buildTable(cl, 19, 7, clt, int(sizeof(clt) / sizeof(Entry)), CODELEN)) return false;
                uint8_t lens[286 + 30 + 138];
                unsigned n = 0;
                while (n < hlit + hdist) {
                    br.refill();
                    const Entry e = clt[br.peek(7)];
                    if (!(e & 15u)) return false;
                    br.drop(int(e & 15u));
                    const unsigned sym = e >> 8;
                    if (sym < 16) lens[n++] = uint8_t(sym);
                    else {
                        unsigned rep, val = 0;
                        if (sym == 16) { if (!n) return false; val = lens[n - 1]; rep = 3 + br.take(2); }
                        else if (sym == 17) rep = 3 + br.take(3);
                        else rep = 11 + br.take(7);
                        if (n + rep > hlit + hdist) return false;
                        while (rep--) lens[n++] = uint8_t(val);
                    }
                    if (br.overrun) return false;
                }
                if (lens[256] == 0) return false;     // no end-of-block code
                if (!buildTable(lens, int(hlit), kLitBits, T.lit, int(sizeof(T.lit) / sizeof(Entry)), LITLEN)) return false;
                if (!buildTable(lens + hlit, int(hdist), kDistBits, T.dist, int(sizeof(T.dist) / sizeof(Entry)), DISTANCE)) return false;
                tab = &T;
            }
            const Entry *lit = tab->lit, *dst = tab->dist;
            for (;;) {
                br.refill();
                Entry e = lit[br.peek(kLitBits)];
                if ((e >> 4 & 3u) == 3u) {           // long code: second lookup
                    br.drop(kLitBits);
                    e = lit[(e >> 8) + br.peek(int(e & 15u))];
                }
                if (!(e & 15u)) return false;
                br.drop(int(e & 15u));
                const unsigned kind = (e >> 4) & 3u;
                if (kind == 0) {
                    if (o >= oend) return false;
                    *o++ = uint8_t(e >> 8);
                    continue;
                }
                if (kind == 2) break;                // end of block
                const unsigned ls = e >> 8;
                if (ls >= 29) return false;
                unsigned len = kLenBase[ls] + br.take(kLenExtra[ls]);
                if (br.cnt < 32) br.refill();
                Entry d = dst[br.peek(kDistBits)];
                if ((d >> 4 & 3u) == 3u) {
                    br.drop(kDistBits);
                    d = dst[(d >> 8) + br.peek(int(d & 15u))];
                }
                if (!(d & 15u)) return false;
                br.drop(int(d & 15u));
                const unsigned ds = d >> 8;
                if (ds >= 30) return false;
                const unsigned dist = kDistBase[ds] + br.take(kDistExtra[ds]);
                if (br.overrun || dist > size_t(o - out) || len > size_t(oend - o)) return false;
                const uint8_t *s = o - dist;
                if (dist >= 8) {                     // eight bytes at a time (may write up to 7 bytes past the match: the slack)
                    uint8_t *const stop = o + len;
                    do { uint64_t w; memcpy(&w, s, 8); memcpy(o, &w, 8); s += 8; o += 8; } while (o < stop);
                    o = stop;
                } else {
                    while (len--) *o++ = *s++;       // overlapping run
                }
            }
            if (br.overrun) return false;
        } else return false;
        if (final) break;
    }
    return o == oend && !br.overrun;
}

// ---------------- CRC-32 ----------------
namespace {

uint32_t g_crcTable[256];
bool g_crcTableReady = false;
void makeCrcTable()
{
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crcTable[n] = c;
    }
    g_crcTableReady = true;
}
struct CrcTableInit { CrcTableInit() { makeCrcTable(); } } g_crcTableInit;

inline uint32_t crcBytes(uint32_t c, const uint8_t *p, size_t n)      // c is the running register (already complemented)
{
    while (n--) c = g_crcTable[(c ^ *p++) & 0xFFu] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
} // namespace
} // namespace dindel
#include <immintrin.h>
namespace dindel {
namespace {
// Folding constants of the reflected CRC-32 polynomial 0x1DB710641 (x^(512+64), x^512, x^(128+64), x^128, x^96 mod P; P'; mu)
__attribute__((target("pclmul,sse4.1")))
uint32_t crcFold(uint32_t c, const uint8_t *buf, size_t len)          // len >= 64 and a multiple of 16; c: running register
{
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x00));
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x10));
    x3 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x20));
    x4 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128(int(c)));
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k1k2));
    buf += 64; len -= 64;
    while (len >= 64) {                                               // four 128-bit lanes folded 512 bits ahead
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x00)); y6 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x10));
        y7 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x20)); y8 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(k3k4));     // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                               // single folds
        x2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(buf));
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                          // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i *>(k5k0));
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128(reinterpret_cast<const __m128i *>(poly));     // Barrett reduction 64 -> 32 bits
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return uint32_t(_mm_extract_epi32(x1, 1));
}
// decided on first use (a static initialiser may run before the CPU model has been probed: __builtin_cpu_init first, GCC manual 6.60.x)
bool hasClmul()
{
    static const bool yes = (__builtin_cpu_init(), __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"));
    return yes;
}
#else
bool hasClmul() { return false; }
uint32_t crcFold(uint32_t c, const uint8_t *, size_t) { return c; }
#endif

} // namespace

uint32_t fastCrc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    if (!g_crcTableReady) makeCrcTable();
    uint32_t c = ~crc;
    if (hasClmul() && len >= 64) {
        const size_t body = len & ~size_t(15);
        c = crcFold(c, buf, body);
        buf += body; len -= body;
    }
    return ~crcBytes(c, buf, len);
}

} // namespace dindel
