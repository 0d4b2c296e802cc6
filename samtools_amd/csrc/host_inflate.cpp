// host_inflate.cpp -- see host_inflate.h.  RFC 1951 section numbers in the comments.
#include "host_inflate.h"
#include <cstring>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#include <initializer_list>
#endif

namespace sta {

namespace {

// one table entry: `bits` code bits to drop, and what the code means
//   op == 0            literal, val = the byte
//   op & OP_BASE       length or distance: val = base, op & 15 = number of extra bits
//   op & OP_EOB        end of block
//   op & OP_LINK       go to the second-level table at val, indexed by the next `bits` bits (the primary index bits are dropped first)
//   op & OP_BAD        no code ends here (incomplete code): the stream is damaged
struct Entry { uint8_t bits; uint8_t op; uint16_t val; };       // as one little-endian word: bits | op << 8 | val << 16 (the fast loop shifts by the word itself)
enum { OP_BASE = 16, OP_EOB = 32, OP_LINK = 64, OP_BAD = 128 };
static_assert(sizeof(Entry) == 4, "the fast loop reads an entry as one 32-bit word");

enum { LIT_PB = 10, DIST_PB = 8, LIT_CAP = (1 << LIT_PB) + 288 * 32, DIST_CAP = (1 << DIST_PB) + 32 * 128 };

const uint16_t LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

enum Kind { K_LITLEN, K_DIST, K_CODELEN };

inline Entry meaning(Kind k, int sym, int bits)
{
    Entry e; e.bits = (uint8_t)bits; e.val = 0; e.op = OP_BAD;
    if (k == K_LITLEN) {
        if (sym < 256) { e.val = (uint16_t)sym; e.op = 0; }
        else if (sym == 256) e.op = OP_EOB;
        else if (sym < 286) { e.val = LEN_BASE[sym - 257]; e.op = (uint8_t)(OP_BASE | LEN_EXTRA[sym - 257]); }
    } else if (k == K_DIST) {
        if (sym < 30) { e.val = DIST_BASE[sym]; e.op = (uint8_t)(OP_BASE | DIST_EXTRA[sym]); }
    } else { e.val = (uint16_t)sym; e.op = 0; }
    return e;
}

inline unsigned bit_reverse(unsigned c, int n)
{
    unsigned r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

// Canonical Huffman code (3.2.2) of n symbols with the given lengths -> look-up table indexed by the next `pb` stream bits (codes are
// packed starting from the least significant bit, so the table index is the bit-reversed code).  Codes longer than pb bits hang off
// second-level tables.  -1: over-subscribed lengths or the table does not fit.  Incomplete codes are accepted; their holes are OP_BAD.
int build_table(const uint8_t *lens, int n, Kind kind, int pb, Entry *tab, int cap)
{
    unsigned count[16] = { 0 }, next[16];
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= (int)count[l]; if (left < 0) return -1; }
    unsigned code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    const int psize = 1 << pb;
    Entry bad; bad.val = 0; bad.bits = 0; bad.op = OP_BAD;
    for (int i = 0; i < psize; ++i) tab[i] = bad;
    uint8_t maxlen[1 << LIT_PB];
    uint16_t rev[320];
    bool any_long = false;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const unsigned r = bit_reverse(next[l]++, l);
        rev[s] = (uint16_t)r;
        if (l <= pb) {
            const Entry e = meaning(kind, s, l);
            for (int i = (int)r; i < psize; i += 1 << l) tab[i] = e;
        } else {
            if (!any_long) { memset(maxlen, 0, (size_t)psize); any_long = true; }
            const unsigned prefix = r & (unsigned)(psize - 1);
            if (l > maxlen[prefix]) maxlen[prefix] = (uint8_t)l;
        }
    }
    if (!any_long) return 0;
    int next_free = psize;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= pb) continue;
        const unsigned prefix = rev[s] & (unsigned)(psize - 1);
        if (!(tab[prefix].op & OP_LINK)) {
            const int sb = maxlen[prefix] - pb;
            if (next_free + (1 << sb) > cap) return -1;
            tab[prefix].val = (uint16_t)next_free; tab[prefix].bits = (uint8_t)sb; tab[prefix].op = OP_LINK;
            for (int j = 0; j < (1 << sb); ++j) tab[next_free + j] = bad;
            next_free += 1 << sb;
        }
        const int sb = tab[prefix].bits, l2 = l - pb;
        const Entry e = meaning(kind, s, l2);
        for (int j = (int)(rev[s] >> pb); j < (1 << sb); j += 1 << l2) tab[tab[prefix].val + j] = e;
    }
    return 0;
}

struct Tables { Entry lit[LIT_CAP]; Entry dist[DIST_CAP]; };

struct FixedTables {
    Tables t;
    FixedTables()
    {
        uint8_t l[288];                           // 3.2.6
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, K_LITLEN, LIT_PB, t.lit, LIT_CAP);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, K_DIST, DIST_PB, t.dist, DIST_CAP);
    }
};

struct Bits {
    uint64_t buf = 0; int cnt = 0;
    const uint8_t *next, *end, *lim;              // end = in + in_len; lim = end + 8: readable
    // at least 56 valid bits afterwards (zeros behind the readable region)
    inline void refill()
    {
        if (next + 8 <= lim) {
            uint64_t w; memcpy(&w, next, 8);
            buf |= w << cnt;
            next += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56) { buf |= (uint64_t)(next < lim ? *next : 0) << cnt; ++next; cnt += 8; }
        }
    }
    inline unsigned peek(int n) const { return (unsigned)(buf & ((1ull << n) - 1)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline unsigned take(int n) { const unsigned v = peek(n); drop(n); return v; }
    // bytes of the input the decoder has really used (bits still in the buffer are handed back)
    inline ptrdiff_t used(const uint8_t *in) const { return (next - in) - (cnt >> 3); }
};

inline Entry lookup(Bits &b, const Entry *tab, int pb)
{
    Entry e = tab[b.peek(pb)];
    if (e.op & OP_LINK) { b.drop(pb); e = tab[e.val + b.peek(e.bits)]; }
    b.drop(e.bits);
    return e;
}

// 3.2.7: the code lengths of a dynamic block -> its two tables
int read_dynamic(Bits &b, Tables &t)
{
    static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    b.refill();
    const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    uint8_t cl[19] = { 0 };
    for (int i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); cl[order[i]] = (uint8_t)b.take(3); }
    Entry ct[1 << 7];
    if (build_table(cl, 19, K_CODELEN, 7, ct, 1 << 7) != 0) return -1;
    uint8_t lens[320];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        b.refill();
        const Entry e = lookup(b, ct, 7);
        if (e.op & OP_BAD) return -1;
        const int sym = e.val;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (i == 0) return -1; val = lens[i - 1]; rep = 3 + (int)b.take(2); }
        else if (sym == 17) rep = 3 + (int)b.take(3);
        else rep = 11 + (int)b.take(7);
        if (i + rep > total) return -1;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return -1;                 // no end-of-block code
    if (build_table(lens, hlit, K_LITLEN, LIT_PB, t.lit, LIT_CAP) != 0) return -1;
    if (build_table(lens + hlit, hdist, K_DIST, DIST_PB, t.dist, DIST_CAP) != 0) return -1;
    return 0;
}

}  // namespace

int fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len)
{
    static const FixedTables fixed;
    static thread_local Tables dyn;
    Bits b; b.next = in; b.end = in + in_len; b.lim = b.end + 8;
    uint8_t *o = out, *const oend = out + out_cap;
    for (;;) {
        b.refill();
        const unsigned final_block = b.take(1), type = b.take(2);
        if (type == 0) {
            // 3.2.4 stored: to the next byte boundary, LEN, NLEN, LEN bytes
            b.drop(b.cnt & 7);
            const uint8_t *p = in + b.used(in);
            b.buf = 0; b.cnt = 0;
            if (p < in || p + 4 > b.end) return 1;
            const unsigned len = p[0] | (unsigned)p[1] << 8, nlen = p[2] | (unsigned)p[3] << 8;
            if ((len ^ 0xffffu) != nlen) return 1;
            p += 4;
            if ((size_t)(b.end - p) < len || (size_t)(oend - o) < len) return 1;
            memcpy(o, p, len);
            o += len; b.next = p + len;
        } else if (type == 1 || type == 2) {
            const Tables *t = &fixed.t;
            if (type == 2) { if (read_dynamic(b, dyn) != 0) return 1; t = &dyn; }
            // ---- fast loop: while 16 input bytes and a longest match + the slack of the wide copies are certainly there, nothing is
            // bounds-checked per symbol.  One refill (>= 56 bits) covers up to three literal codes, or two and a length code with its
            // extra bits (15 + 15 + 15 + 5); a second refill covers the distance code and its extra bits (15 + 13). ----
            const Entry *const lt = t->lit, *const dt = t->dist;
            bool done = false;
            {
                // entries as 32-bit words: the code length sits in the low byte, so `buf >> (e & 63)` needs no field extraction on the
                // path from one look-up to the next (a shift count is taken modulo 64 anyway); literal <=> bits 8..15 clear
                uint64_t buf = b.buf; int cnt = b.cnt; const uint8_t *nx = b.next;
                const auto ld = [](const Entry *t2, uint64_t idx) { uint32_t w; memcpy(&w, t2 + idx, 4); return w; };
                while ((ptrdiff_t)(b.end - nx) >= 16 && (ptrdiff_t)(oend - o) >= 258 + 16) {
                    { uint64_t w; memcpy(&w, nx, 8); buf |= w << cnt; nx += (63 - cnt) >> 3; cnt |= 56; }
                    uint32_t e = ld(lt, buf & ((1u << LIT_PB) - 1));
                    if (!(e & 0xff00u)) {
                        buf >>= (e & 63); cnt -= (int)(e & 63); *o++ = (uint8_t)(e >> 16);
                        e = ld(lt, buf & ((1u << LIT_PB) - 1));
                        if (!(e & 0xff00u)) {
                            buf >>= (e & 63); cnt -= (int)(e & 63); *o++ = (uint8_t)(e >> 16);
                            e = ld(lt, buf & ((1u << LIT_PB) - 1));
                            if (!(e & 0xff00u)) { buf >>= (e & 63); cnt -= (int)(e & 63); *o++ = (uint8_t)(e >> 16); continue; }
                        }
                    }
                    if (e & ((uint32_t)OP_LINK << 8)) { buf >>= LIT_PB; cnt -= LIT_PB; e = ld(lt, (e >> 16) + (buf & ((1u << (e & 63)) - 1))); }
                    buf >>= (e & 63); cnt -= (int)(e & 63);
                    if (!(e & 0xff00u)) { *o++ = (uint8_t)(e >> 16); continue; }                       // a literal with a long code
                    if (!(e & ((uint32_t)OP_BASE << 8))) { if (e & ((uint32_t)OP_EOB << 8)) { done = true; break; } return 1; }
                    const int xl = (int)((e >> 8) & 15);
                    const unsigned len = (e >> 16) + (unsigned)(buf & ((1u << xl) - 1));
                    buf >>= xl; cnt -= xl;
                    { uint64_t w; memcpy(&w, nx, 8); buf |= w << cnt; nx += (63 - cnt) >> 3; cnt |= 56; }
                    uint32_t d = ld(dt, buf & ((1u << DIST_PB) - 1));
                    if (d & ((uint32_t)OP_LINK << 8)) { buf >>= DIST_PB; cnt -= DIST_PB; d = ld(dt, (d >> 16) + (buf & ((1u << (d & 63)) - 1))); }
                    buf >>= (d & 63); cnt -= (int)(d & 63);
                    if (!(d & ((uint32_t)OP_BASE << 8))) return 1;
                    const int xd = (int)((d >> 8) & 15);
                    const unsigned dist = (d >> 16) + (unsigned)(buf & ((1u << xd) - 1));
                    buf >>= xd; cnt -= xd;
                    if (dist > (size_t)(o - out)) return 1;
                    const uint8_t *src = o - dist;
                    uint8_t *dst = o;
                    o += len;
                    if (dist >= 16) {
                        memcpy(dst, src, 16);                                           // most matches are short: one 16-byte move
                        if (len > 16) { dst += 16; src += 16; do { memcpy(dst, src, 16); dst += 16; src += 16; } while (dst < o); }
                    } else if (dist >= 8) { do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < o); }
                    else if (dist == 1) { uint64_t v = 0x0101010101010101ull * *src; do { memcpy(dst, &v, 8); dst += 8; } while (dst < o); }
                    else { while (dst < o) *dst++ = *src++; }
                }
                b.buf = buf; b.cnt = cnt; b.next = nx;
            }
            // ---- careful loop: the last bytes of the block, every access checked ----
            while (!done) {
                if (b.next > b.lim + 8) return 1;          // ran off the input long ago: damaged stream
                b.refill();                                // >= 56 bits: two literal codes and a length code with its extra bits fit
                Entry e = lookup(b, t->lit, LIT_PB);
                if (e.op == 0) {
                    if (o >= oend) return 1;
                    *o++ = (uint8_t)e.val;
                    e = lookup(b, t->lit, LIT_PB);
                    if (e.op == 0) { if (o >= oend) return 1; *o++ = (uint8_t)e.val; continue; }
                }
                if (e.op & OP_BASE) {
                    const unsigned len = e.val + b.take(e.op & 15);
                    b.refill();                            // a distance code and its extra bits: at most 28
                    const Entry d = lookup(b, t->dist, DIST_PB);
                    if (!(d.op & OP_BASE)) return 1;
                    const unsigned dist = d.val + b.take(d.op & 15);
                    if (dist > (size_t)(o - out) || len > (size_t)(oend - o)) return 1;
                    const uint8_t *src = o - dist;
                    uint8_t *dst = o;
                    o += len;
                    if (dist >= 8) {                                                                        // (exact: a caller may have live data right behind out_cap)
                        while (o - dst >= 8) { memcpy(dst, src, 8); dst += 8; src += 8; }
                        while (dst < o) *dst++ = *src++;
                    }
                    else if (dist == 1) memset(dst, *src, len);
                    else { while (dst < o) *dst++ = *src++; }
                } else if (e.op & OP_EOB) break;
                else return 1;
            }
        } else return 1;
        if (final_block) break;
    }
    if (b.used(in) > (ptrdiff_t)in_len) return 1;         // the stream needed bytes beyond its end
    *out_len = (size_t)(o - out);
    return 0;
}

// slicing-by-8 (eight table look-ups per eight input bytes), from the running state c (no pre / post inversion)
static uint32_t crc32_tables(const uint8_t *p, size_t n, uint32_t c)
{
    struct Tab {
        uint32_t t[8][256];
        Tab()
        {
            for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
            for (uint32_t i = 0; i < 256; ++i) for (int s = 1; s < 8; ++s) t[s][i] = t[0][t[s - 1][i] & 0xff] ^ (t[s - 1][i] >> 8);
        }
    };
    static const Tab T;
    while (n && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint32_t a, b2; memcpy(&a, p, 4); memcpy(&b2, p + 4, 4);
        a ^= c;
        c = T.t[7][a & 0xff] ^ T.t[6][(a >> 8) & 0xff] ^ T.t[5][(a >> 16) & 0xff] ^ T.t[4][a >> 24]
          ^ T.t[3][b2 & 0xff] ^ T.t[2][(b2 >> 8) & 0xff] ^ T.t[1][(b2 >> 16) & 0xff] ^ T.t[0][b2 >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// Carry-less-multiply folding (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009;
// the bit-reflected constants for the polynomial 0xEDB88320 are the paper's): four 128-bit lanes folded 64 bytes at a time, then to
// one lane, to 64 bits, Barrett reduction.  n >= 64 and a multiple of 16; running state in and out, like crc32_tables.  ~8x the table
// walk on the hosts measured -- the decode threads' CRC check was a sixth of their work.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(const uint8_t *p, size_t n, uint32_t c)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    const __m128i *v = (const __m128i *)p;
    __m128i x1 = _mm_xor_si128(_mm_loadu_si128(v), _mm_cvtsi32_si128((int)c)), x2 = _mm_loadu_si128(v + 1), x3 = _mm_loadu_si128(v + 2), x4 = _mm_loadu_si128(v + 3);
    v += 4; n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), a1), _mm_loadu_si128(v));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), a2), _mm_loadu_si128(v + 1));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), a3), _mm_loadu_si128(v + 2));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), a4), _mm_loadu_si128(v + 3));
        v += 4; n -= 64;
    }
#define STA_CRC_FOLD(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128((acc), k3k4, 0x11), _mm_clmulepi64_si128((acc), k3k4, 0x00)), (next))
    x1 = STA_CRC_FOLD(x1, x2); x1 = STA_CRC_FOLD(x1, x3); x1 = STA_CRC_FOLD(x1, x4);
    while (n >= 16) { x1 = STA_CRC_FOLD(x1, _mm_loadu_si128(v)); ++v; n -= 16; }
#undef STA_CRC_FOLD
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    // Barrett reduction to 32 bits
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(t, poly, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

// usable on this CPU, and agreeing with the table walk on a probe (checked once)
static bool clmul_ok()
{
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        if (const char *e = getenv("STA_CRC")) if (!strcmp(e, "tables")) return false;
        uint8_t probe[64 * 5 + 48];
        uint32_t s = 12345;
        for (uint8_t &b : probe) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        for (size_t n : { (size_t)64, (size_t)80, (size_t)128, sizeof probe }) if (crc32_clmul(probe, n, 0x2468ace1u) != crc32_tables(probe, n, 0x2468ace1u)) return false;
        return true;
    }();
    return ok;
}
#endif

uint32_t fast_crc32(const uint8_t *p, size_t n)
{
    uint32_t c = ~0u;
#if defined(__x86_64__)
    if (n >= 64 && clmul_ok()) {
        const size_t m = n & ~(size_t)15;
        c = crc32_clmul(p, m, c);
        p += m; n -= m;
    }
#endif
    return ~crc32_tables(p, n, c);
}

}  // namespace sta
