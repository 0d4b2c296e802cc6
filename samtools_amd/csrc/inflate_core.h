// inflate_core.h -- raw DEFLATE (RFC 1951) decoding of ONE BGZF block by ONE thread, and the block's CRC-32 (RFC 1952).
// SURVEY.md 8(f)-2 "GPU inflate later": what HTSlib's bgzf.c does per block (inflate_block -> zlib inflate, then the
// crc32 check of bgzf_read_block; HTSlib is absent from the reference tree, the formats are the published RFCs and the SAM
// specification 4.1).  A BGZF block is self-contained (no dictionary crosses blocks, at most 64 KiB either side), so a file is
// thousands of independent streams: k_bgzf_inflate (kernels_inflate.hip) gives every lane its own block.  The decoder is
// written for that setting -- canonical-code decoding from (count per length, symbols in code order), 38 + 320 16-bit entries
// per lane that the kernel keeps in LDS (stride = lanes per wave, so that lanes touching the same entry hit different banks),
// no look-up tables to build per block, bit-serial code walk.  Plain functions shared by the kernel and by the CPU harness
// (tests/cpu/inflate_emul.cpp) that checks them against zlib on every block of the test inputs.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define INF_HD __host__ __device__ __forceinline__
#else
#define INF_HD inline
#endif

namespace sta_inflate {

enum { OK = 0, ERR_INPUT_END = 1, ERR_BLOCK_TYPE = 2, ERR_STORED_LEN = 3, ERR_CODE_LENGTHS = 4, ERR_BAD_CODE = 5, ERR_DISTANCE = 6,
       ERR_OUTPUT_FULL = 7, ERR_SIZE = 8, ERR_CRC = 9 };

// bits, least significant first, one byte at a time: never touches a byte the stream does not need
struct BitIn {
    const uint8_t *p; uint32_t n, pos; uint64_t buf; int cnt; bool over;
};
INF_HD uint32_t take(BitIn &b, int need)      // need <= 16
{
    while (b.cnt < need) {
        uint32_t v = 0;
        if (b.pos < b.n) v = b.p[b.pos]; else b.over = true;
        ++b.pos;
        b.buf |= (uint64_t)v << b.cnt; b.cnt += 8;
    }
    const uint32_t r = (uint32_t)b.buf & ((1u << need) - 1u);
    b.buf >>= need; b.cnt -= need;
    return r;
}

// a canonical Huffman code: count[len] codes of every length 1..15, symbols ordered by (length, value); entries `stride` apart.
// cnt[] is a copy of the counts that the decode loop reads with constant indices (fully unrolled): registers on the device, so
// that a code bit costs arithmetic only and the one table access of a symbol is the final look-up
struct Huff { uint16_t *count; uint16_t *symbol; int stride; uint16_t cnt[16]; };

INF_HD int decode(BitIn &b, const Huff &h)
{
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        code |= (int)take(b, 1);
        const int count = h.cnt[len];
        if (code - count < first) return h.symbol[(index + (code - first)) * h.stride];
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return -1;
}

// builds h from n code lengths (0 = unused); returns 0 for a complete code, < 0 over-subscribed, > 0 incomplete
template <class LenAt>
INF_HD int build(Huff &h, const LenAt &len_at, int n)
{
    for (int l = 0; l <= 15; ++l) h.count[l * h.stride] = 0;
    for (int s = 0; s < n; ++s) { const int l = len_at(s); h.count[l * h.stride] = (uint16_t)(h.count[l * h.stride] + 1); }
#pragma unroll
    for (int l = 0; l <= 15; ++l) h.cnt[l] = h.count[l * h.stride];
    if (h.count[0] == n) return 0;                       // no codes at all: complete, but decoding anything fails
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= h.count[l * h.stride]; if (left < 0) return left; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l * h.stride]);
    for (int s = 0; s < n; ++s) { const int l = len_at(s); if (l) { h.symbol[offs[l] * h.stride] = (uint16_t)s; ++offs[l]; } }
    return left;
}

// CRC-32 (reflected 0xEDB88320), one byte per step; tab = the usual 256 words
INF_HD void crc_table_entry(uint32_t i, uint32_t *out) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; *out = c; }

struct Out { uint8_t *p; uint32_t cap, pos; uint32_t crc; const uint32_t *tab; };
INF_HD void put(Out &o, uint32_t byte)
{
    o.p[o.pos++] = (uint8_t)byte;
    o.crc = o.tab[(o.crc ^ byte) & 255u] ^ (o.crc >> 8);
}

// literal / length and distance codes -> bytes, until the end-of-block symbol
INF_HD int codes(BitIn &b, Out &o, const Huff &lit, const Huff &dist)
{
    for (;;) {
        int sym = decode(b, lit);
        if (sym < 0 || b.over) return b.over ? ERR_INPUT_END : ERR_BAD_CODE;
        if (sym < 256) {
            if (o.pos >= o.cap) return ERR_OUTPUT_FULL;
            put(o, (uint32_t)sym);
        } else if (sym == 256) return OK;
        else {
            sym -= 257;
            if (sym >= 29) return ERR_BAD_CODE;
            // length: 3..10 plain, then 4 codes per extra-bit count, 258 for the last symbol
            int len;
            if (sym < 8) len = 3 + sym;
            else if (sym == 28) len = 258;
            else { const int e = (sym >> 2) - 1; len = 3 + ((4 + (sym & 3)) << e) + (int)take(b, e); }
            int ds = decode(b, dist);
            if (ds < 0 || ds >= 30) return b.over ? ERR_INPUT_END : ERR_BAD_CODE;
            uint32_t d;
            if (ds < 4) d = 1u + (uint32_t)ds;
            else { const int e = (ds >> 1) - 1; d = 1u + ((2u + (uint32_t)(ds & 1)) << e) + take(b, e); }
            if (b.over) return ERR_INPUT_END;
            if (d > o.pos) return ERR_DISTANCE;
            if ((uint32_t)len > o.cap - o.pos) return ERR_OUTPUT_FULL;
            for (int i = 0; i < len; ++i) put(o, o.p[o.pos - d]);
        }
    }
}

// Scratch a thread needs: the two codes (count 16 + symbols 288 / 32, 16-bit, `stride` apart) and 320 code lengths (bytes)
struct Work { Huff lit, dist; uint8_t *lens; int lens_stride; };

// One whole DEFLATE stream (all its blocks) of `in_len` bytes into out[0 .. cap); *out_len = bytes produced, *crc = their CRC-32
INF_HD int inflate_stream(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t cap, const uint32_t *crc_tab, Work &w, uint32_t *out_len, uint32_t *crc)
{
    BitIn b; b.p = in; b.n = in_len; b.pos = 0; b.buf = 0; b.cnt = 0; b.over = false;
    Out o; o.p = out; o.cap = cap; o.pos = 0; o.crc = 0xffffffffu; o.tab = crc_tab;
    int err = OK, last;
    do {
        last = (int)take(b, 1);
        const int type = (int)take(b, 2);
        if (b.over) { err = ERR_INPUT_END; break; }
        if (type == 0) {
            // stored: to the byte boundary, LEN, ~LEN, bytes
            b.buf = 0; b.cnt = 0;
            if (b.pos + 4 > b.n) { err = ERR_INPUT_END; break; }
            const uint32_t len = b.p[b.pos] | (uint32_t)b.p[b.pos + 1] << 8, nlen = b.p[b.pos + 2] | (uint32_t)b.p[b.pos + 3] << 8;
            b.pos += 4;
            if (len != (~nlen & 0xffffu)) { err = ERR_STORED_LEN; break; }
            if (b.pos + len > b.n) { err = ERR_INPUT_END; break; }
            if (len > o.cap - o.pos) { err = ERR_OUTPUT_FULL; break; }
            for (uint32_t i = 0; i < len; ++i) put(o, b.p[b.pos + i]);
            b.pos += len;
        } else if (type == 1) {
            // fixed code: lengths 8 / 9 / 7 / 8 for the literal-length alphabet, 5 bits for every distance
            const int ls = w.lens_stride; uint8_t *ln = w.lens;
            build(w.lit, [](int s) { return s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8; }, 288);
            build(w.dist, [](int) { return 5; }, 30);
            (void)ls; (void)ln;
            err = codes(b, o, w.lit, w.dist);
        } else if (type == 2) {
            const int nlen = (int)take(b, 5) + 257, ndist = (int)take(b, 5) + 1, ncode = (int)take(b, 4) + 4;
            if (b.over) { err = ERR_INPUT_END; break; }
            if (nlen > 286 || ndist > 30) { err = ERR_CODE_LENGTHS; break; }
            const int ls = w.lens_stride; uint8_t *ln = w.lens;
            // the code-length code, in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (5 bits per entry below)
            const uint64_t order_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
            const uint64_t order_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
            for (int i = 0; i < 19; ++i) ln[i * ls] = 0;
            for (int i = 0; i < ncode; ++i) {
                const int s = (int)((i < 12 ? order_lo >> (5 * i) : order_hi >> (5 * (i - 12))) & 31u);
                ln[s * ls] = (uint8_t)take(b, 3);
            }
            if (b.over) { err = ERR_INPUT_END; break; }
            // (the code-length code borrows the distance code's storage: 16 counts + 19 symbols fit in 16 + 32)
            if (build(w.dist, [ln, ls](int s) { return (int)ln[s * ls]; }, 19) != 0) { err = ERR_CODE_LENGTHS; break; }
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(b, w.dist);
                if (sym < 0 || b.over) { err = b.over ? ERR_INPUT_END : ERR_BAD_CODE; break; }
                if (sym < 16) ln[(idx++) * ls] = (uint8_t)sym;
                else {
                    int len = 0, rep;
                    if (sym == 16) { if (idx == 0) { err = ERR_CODE_LENGTHS; break; } len = ln[(idx - 1) * ls]; rep = 3 + (int)take(b, 2); }
                    else if (sym == 17) rep = 3 + (int)take(b, 3);
                    else rep = 11 + (int)take(b, 7);
                    if (idx + rep > nlen + ndist) { err = ERR_CODE_LENGTHS; break; }
                    while (rep--) ln[(idx++) * ls] = (uint8_t)len;
                }
            }
            if (err) break;
            if (b.over) { err = ERR_INPUT_END; break; }
            if (ln[256 * ls] == 0) { err = ERR_CODE_LENGTHS; break; }                 // no end-of-block code
            int left = build(w.lit, [ln, ls](int s) { return (int)ln[s * ls]; }, nlen);
            if (left != 0 && (left < 0 || nlen != w.lit.count[0] + w.lit.count[1 * w.lit.stride])) { err = ERR_CODE_LENGTHS; break; }
            left = build(w.dist, [ln, ls, nlen](int s) { return (int)ln[(nlen + s) * ls]; }, ndist);
            if (left != 0 && (left < 0 || ndist != w.dist.count[0] + w.dist.count[1 * w.dist.stride])) { err = ERR_CODE_LENGTHS; break; }
            err = codes(b, o, w.lit, w.dist);
        } else err = ERR_BLOCK_TYPE;
    } while (!err && !last);
    *out_len = o.pos; *crc = o.crc ^ 0xffffffffu;
    return err;
}

}  // namespace sta_inflate
