// bgzf_inflate_dev.h -- raw DEFLATE (RFC 1951) decoding of one BGZF block by ONE WAVE (gfx950), as plain functions that
// tests/cpu/inflate_emul.cpp also runs on the CPU (one "lane", the cooperative loops degenerate) against zlib.
//
// Stands where HTSlib's bgzf.c inflate_block() (absent from the reference tree; the call chain is sam_read1 -> bgzf_read ->
// bgzf_read_block, reached from mplp_func, bam_plcmd.c:409, and fastdepth_core, bam2depth.c:541-543) stands on the host: a BGZF
// block is a self-contained deflate stream of at most 64 KiB either side (SAM spec 4.1), so a file is tens of thousands of
// independent streams -- one wave each.
//
// Symbol decoding is serial by nature; the wave runs it as UNIFORM code (every lane computes the same bit-reader state, which the
// compiler keeps in scalar registers) and uses its 64 lanes where a step has width:
//   * the compressed bytes are held 8 per lane (512 bytes of input in one coalesced load); the bit reader takes its next 64 bits
//     from two lanes with v_readlane -- no memory round trip on the symbol path;
//   * the last 16 KiB of output live in an LDS ring: a literal is one ds_write_b8, a match is copied by up to 64 lanes at once
//     (periodic source for overlapping copies); every finished 4 KiB page of the ring leaves LDS as 16-byte stores, and the few
//     matches that reach further back than the ring (3 % on BAM data) read the bytes already written to the destination;
//   * Huffman tables (two-level, 10 / 8 root bits, the layout of host_inflate.cpp) are built in LDS with lane-parallel fills.
// Anything unexpected -- a damaged stream, a table that does not fit -- ends the block with a non-zero status; the host inflates
// such a block again with zlib, whose verdict counts (host_bgzf.cpp bgzf_inflate_block).  The CRC-32 of every block is checked on
// the host by the thread that parses it.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define BZ_HD __device__ __forceinline__
#define BZ_LANE ((int)(threadIdx.x & 63))
#define BZ_NL 64
#define BZ_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define BZ_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define BZ_MEM_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")
#define BZ_BALLOT(c) ((uint64_t)__ballot(c))
#define BZ_COUNT(c) ((unsigned)__popcll(__ballot(c)))
#define BZ_POPC(m) ((unsigned)__popcll((unsigned long long)(m)))
#define BZ_LT_MASK ((1ull << (threadIdx.x & 63)) - 1ull)
#define BZ_FFS(m) (__ffsll((long long)(m)) - 1)
#define BZ_READLANE(v, l) __builtin_amdgcn_readlane((int)(v), (l))
#else
#define BZ_HD static inline
#define BZ_LANE 0
#define BZ_NL 1
#define BZ_UNI(x) ((uint32_t)(x))
#define BZ_LDS_FENCE() ((void)0)
#define BZ_MEM_FENCE() ((void)0)
#define BZ_BALLOT(c) ((uint64_t)((c) ? 1 : 0))
#define BZ_COUNT(c) ((unsigned)((c) ? 1 : 0))
#define BZ_POPC(m) ((unsigned)__builtin_popcountll((unsigned long long)(m)))
#define BZ_LT_MASK 0ull
#define BZ_FFS(m) (__builtin_ffsll((long long)(m)) - 1)
#define BZ_READLANE(v, l) ((void)(l), (v))
#endif

namespace bgzi {

// (table capacities: zlib's ENOUGH bounds for these root sizes are 1332 / ~400 entries)
enum { LIT_PB = 10, DIST_PB = 8, LIT_CAP = (1 << LIT_PB) + 320, DIST_CAP = (1 << DIST_PB) + 256, RING = 16384, RING_MASK = RING - 1, PAGE_SHIFT = 12 };
enum { OP_BASE = 16, OP_EOB = 32, OP_LINK = 64, OP_BAD = 128 };
// a table entry as one word: bits | op << 8 | val << 16 (host_inflate.cpp's Entry)
BZ_HD uint32_t mk_entry(int bits, int op, int val) { return (uint32_t)bits | (uint32_t)op << 8 | (uint32_t)val << 16; }

enum { ST_OK = 0, ST_BAD_STREAM = 1, ST_TABLE = 2, ST_SIZE = 3, ST_INPUT = 4 };

// per-wave LDS
struct Lds {
    uint32_t lit[LIT_CAP];
    uint32_t dist[DIST_CAP];
    uint32_t ct[128];               // the code-length code's table
    uint16_t rev[320];
    uint8_t lens[320];
    alignas(16) uint8_t ring[RING];             // the last RING output bytes: output position x lives at ring[(align + x) & RING_MASK], align = the block's destination address mod 16
};

struct Consts {                     // RFC 1951 3.2.5 tables (+ 2^20 / d rounded up, for the periodic copies), in device memory
    uint16_t len_base[29]; uint8_t len_extra[29];
    uint16_t dist_base[30]; uint8_t dist_extra[30];
    uint8_t order[19];
    uint32_t inv20[64];
};

BZ_HD void fill_consts(Consts &c)
{
    const uint16_t lb[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
    const uint8_t le[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
    const uint16_t db[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
    const uint8_t de[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
    const uint8_t od[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    for (int i = 0; i < 29; ++i) { c.len_base[i] = lb[i]; c.len_extra[i] = le[i]; }
    for (int i = 0; i < 30; ++i) { c.dist_base[i] = db[i]; c.dist_extra[i] = de[i]; }
    for (int i = 0; i < 19; ++i) c.order[i] = od[i];
    c.inv20[0] = 0;
    for (uint32_t d = 1; d < 64; ++d) c.inv20[d] = ((1u << 20) + d - 1) / d;
}

enum Kind { K_LITLEN, K_DIST, K_CODELEN };

BZ_HD uint32_t meaning(const Consts &C, int k, int sym, int bits)
{
    if (k == K_LITLEN) {
        if (sym < 256) return mk_entry(bits, 0, sym);
        if (sym == 256) return mk_entry(bits, OP_EOB, 0);
        if (sym < 286) return mk_entry(bits, OP_BASE | C.len_extra[sym - 257], C.len_base[sym - 257]);
        return mk_entry(bits, OP_BAD, 0);
    }
    if (k == K_DIST) return sym < 30 ? mk_entry(bits, OP_BASE | C.dist_extra[sym], C.dist_base[sym]) : mk_entry(bits, OP_BAD, 0);
    return mk_entry(bits, 0, sym);
}

BZ_HD unsigned bit_reverse(unsigned c, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(c) >> (32 - n);
#else
    unsigned r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
#endif
}

// Canonical Huffman code (RFC 1951 3.2.2) of n symbols -> look-up table indexed by the next pb stream bits, second-level tables for
// longer codes (the layout of host_inflate.cpp build_table).  A lane owns the symbols lane, lane + 64, ...: code lengths are counted
// with ballots, a symbol's code is its length's first code plus the number of equally long symbols in front of it (ballot + lane
// mask), and every lane replicates its own symbols' entries; only the (few) codes longer than pb bits are placed one at a time.
// 0 = fine.
BZ_HD int build_table(Lds &L, const Consts &C, const uint8_t *lens, int n, int kind, int pb, uint32_t *tab, int cap)
{
    const int lane = BZ_LANE;
    unsigned count[16], next[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int base = 0; base < n; base += BZ_NL) {
        const int ls = base + lane < n ? (int)lens[base + lane] : 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) count[l] += BZ_COUNT(ls == l);
    }
    int left = 1;
#pragma unroll
    for (int l = 1; l <= 15; ++l) { left <<= 1; left -= (int)count[l]; if (left < 0) return ST_TABLE; }
    unsigned code = 0;
    count[0] = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    const int psize = 1 << pb;
    const uint32_t bad = mk_entry(0, OP_BAD, 0);
    for (int i = lane; i < psize; i += BZ_NL) tab[i] = bad;
    BZ_LDS_FENCE();
    bool any_long = false;
    for (int base = 0; base < n; base += BZ_NL) {
        const int s = base + lane;
        const int ls = s < n ? (int)lens[s] : 0;
        unsigned cd = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            const uint64_t m = BZ_BALLOT(ls == l);
            if (ls == l) cd = next[l] + BZ_POPC(m & BZ_LT_MASK);
            next[l] += BZ_POPC(m);
        }
        const unsigned r = ls ? bit_reverse(cd, ls) : 0u;
        if (s < n) L.rev[s] = (uint16_t)r;
        if (ls && ls <= pb) {
            const uint32_t e = meaning(C, kind, s, ls);
            for (int i = (int)r; i < psize; i += 1 << ls) tab[i] = e;
        }
        uint64_t lm = BZ_BALLOT(ls > pb);
        while (lm) {
            const int j = BZ_FFS(lm); lm &= lm - 1;
            const int lj = (int)BZ_READLANE(ls, j);
            const unsigned prefix = (unsigned)BZ_READLANE(r, j) & (unsigned)(psize - 1);
            // (the longest code under a primary prefix is kept in that prefix's own -- still empty -- entry until the sub-table is placed)
            if (lane == 0 && lj > (int)(tab[prefix] & 0xff)) tab[prefix] = mk_entry(lj, OP_BAD, 0);
            any_long = true;
        }
    }
    BZ_LDS_FENCE();
    if (!any_long) return 0;
    int next_free = psize;
    for (int base = 0; base < n; base += BZ_NL) {
        const int s0 = base + lane;
        const int ls = s0 < n ? (int)lens[s0] : 0;
        const unsigned rs = s0 < n ? (unsigned)L.rev[s0] : 0u;
        uint64_t lm = BZ_BALLOT(ls > pb);
        while (lm) {
            const int j = BZ_FFS(lm); lm &= lm - 1;
            const int s = base + j, l = (int)BZ_READLANE(ls, j);
            const unsigned rv = (unsigned)BZ_READLANE(rs, j);
            const unsigned prefix = rv & (unsigned)(psize - 1);
            uint32_t pe = BZ_UNI(tab[prefix]);
            if (!((pe >> 8) & OP_LINK)) {
                const int sb = (int)(pe & 0xff) - pb;
                if (next_free + (1 << sb) > cap) return ST_TABLE;
                pe = mk_entry(sb, OP_LINK, next_free);
                if (lane == 0) tab[prefix] = pe;
                for (int k = lane; k < (1 << sb); k += BZ_NL) tab[next_free + k] = bad;
                next_free += 1 << sb;
                BZ_LDS_FENCE();
            }
            const int sb = (int)(pe & 0xff), l2 = l - pb, tb = (int)(pe >> 16);
            const uint32_t e = meaning(C, kind, s, l2);
            for (int k = (int)(rv >> pb) + (lane << l2); k < (1 << sb); k += BZ_NL << l2) tab[tb + k] = e;
            BZ_LDS_FENCE();
        }
    }
    BZ_LDS_FENCE();
    return 0;
}

// ---- the bit reader: 64-bit buffer, refilled from the wave's input window ----
struct InWin {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo, hi;                 // this lane's 8 bytes: in[base + 8 lane .. + 8)
#else
    uint64_t w[64];
#endif
    int64_t base;                    // byte offset (from the start of the deflate data) of lane 0's bytes; < 0: nothing loaded
};

struct Bits {
    uint64_t buf; int cnt;
    int64_t next;                    // offset of the next byte that has not entered the buffer
    int64_t in_len;                  // deflate bytes; the input is readable 8 bytes beyond and zero-extended by the loader
    const uint8_t *in;
};

BZ_HD void win_load(InWin &W, const Bits &b, int64_t at)
{
    W.base = at;
#if defined(__HIP_DEVICE_COMPILE__)
    // (bytes behind the block belong to the next block or to the buffer's padding: loaded, never used for output)
    const uint8_t *p = b.in + at + 8 * BZ_LANE;
    uint32_t v[2];
    __builtin_memcpy(v, p, 8);
    W.lo = v[0]; W.hi = v[1];
#else
    for (int l = 0; l < 64; ++l) { uint64_t v = 0; memcpy(&v, b.in + at + 8 * l, 8); W.w[l] = v; }
#endif
}

BZ_HD uint64_t win_get64(InWin &W, const Bits &b, int64_t at)
{
    if (W.base < 0 || at < W.base || at - W.base > 8 * 62) win_load(W, b, at);
    const int rel = (int)(at - W.base), l = rel >> 3, sh = (rel & 7) * 8;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t a = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)W.lo, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)W.hi, l) << 32;
    const uint64_t c = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)W.lo, l + 1) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)W.hi, l + 1) << 32;
#else
    const uint64_t a = W.w[l], c = W.w[l + 1];
#endif
    return sh ? (a >> sh) | (c << (64 - sh)) : a;
}

// at least 56 valid bits afterwards
BZ_HD void refill(Bits &b, InWin &W)
{
    const uint64_t w = win_get64(W, b, b.next);
    b.buf |= w << b.cnt;
    b.next += (63 - b.cnt) >> 3;
    b.cnt |= 56;
}
BZ_HD unsigned peek(const Bits &b, int n) { return (unsigned)(b.buf & ((1ull << n) - 1)); }
BZ_HD void drop(Bits &b, int n) { b.buf >>= n; b.cnt -= n; }
BZ_HD unsigned take(Bits &b, int n) { const unsigned v = peek(b, n); drop(b, n); return v; }

BZ_HD uint32_t lookup(Bits &b, const uint32_t *tab, int pb)
{
    uint32_t e = BZ_UNI(tab[peek(b, pb)]);
    if ((e >> 8) & OP_LINK) { drop(b, pb); e = BZ_UNI(tab[(e >> 16) + peek(b, (int)(e & 0xff))]); }
    drop(b, (int)(e & 0xff));
    return e;
}

// RFC 1951 3.2.7: the code lengths of a dynamic block -> the two tables
BZ_HD int read_dynamic(Lds &L, const Consts &C, Bits &b, InWin &W)
{
    const int lane = BZ_LANE;
    refill(b, W);
    const int hlit = (int)take(b, 5) + 257, hdist = (int)take(b, 5) + 1, hclen = (int)take(b, 4) + 4;
    if (hlit > 286 || hdist > 30) return ST_BAD_STREAM;
    for (int i = lane; i < 19; i += BZ_NL) L.lens[i] = 0;
    BZ_LDS_FENCE();
    for (int i = 0; i < hclen; ++i) {
        if (b.cnt < 3) { if (b.next > b.in_len + 8) return ST_INPUT; refill(b, W); }      // (a truncated header must not walk the window beyond the loader's pad)
        const unsigned v = take(b, 3); if (lane == 0) L.lens[C.order[i]] = (uint8_t)v;
    }
    BZ_LDS_FENCE();
    {
        // the 19 lengths move out of the way of the 320 they describe
        uint8_t cl[19];
        for (int i = 0; i < 19; ++i) cl[i] = (uint8_t)BZ_UNI(L.lens[i]);
        BZ_LDS_FENCE();
        for (int i = lane; i < 19; i += BZ_NL) L.lens[300 + i] = cl[i];
        BZ_LDS_FENCE();
    }
    if (build_table(L, C, L.lens + 300, 19, K_CODELEN, 7, L.ct, 128) != 0) return ST_BAD_STREAM;
    int i = 0;
    const int total = hlit + hdist;
    int prev = 0;
    while (i < total) {
        // up to 316 refills in this loop: without the test a crafted or truncated header in the last block of a batch reads ~100 bytes
        // beyond the 1 KiB zero pad behind the compressed bytes
        if (b.next > b.in_len + 8) return ST_INPUT;
        refill(b, W);
        const uint32_t e = lookup(b, L.ct, 7);
        if ((e >> 8) & OP_BAD) return ST_BAD_STREAM;
        const int sym = (int)(e >> 16);
        if (sym < 16) { if (lane == 0) L.lens[i] = (uint8_t)sym; prev = sym; ++i; continue; }
        int rep, val = 0;
        if (sym == 16) { if (i == 0) return ST_BAD_STREAM; val = prev; rep = 3 + (int)take(b, 2); }
        else if (sym == 17) rep = 3 + (int)take(b, 3);
        else rep = 11 + (int)take(b, 7);
        if (i + rep > total) return ST_BAD_STREAM;
        for (int j = lane; j < rep; j += BZ_NL) L.lens[i + j] = (uint8_t)val;
        i += rep; prev = val;
    }
    BZ_LDS_FENCE();
    if (BZ_UNI(L.lens[256]) == 0) return ST_BAD_STREAM;             // no end-of-block code
    // (the literal table's build reuses `rev`; `lens` itself is not touched)
    int rc = build_table(L, C, L.lens, hlit, K_LITLEN, LIT_PB, L.lit, LIT_CAP);
    if (rc) return rc;
    return build_table(L, C, L.lens + hlit, hdist, K_DIST, DIST_PB, L.dist, DIST_CAP);
}

BZ_HD int fixed_tables(Lds &L, const Consts &C)
{
    const int lane = BZ_LANE;
    for (int i = lane; i < 288; i += BZ_NL) L.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
    for (int i = lane; i < 32; i += BZ_NL) L.lens[288 + i] = 5;
    BZ_LDS_FENCE();
    int rc = build_table(L, C, L.lens, 288, K_LITLEN, LIT_PB, L.lit, LIT_CAP);
    if (rc) return rc;
    return build_table(L, C, L.lens + 288, 32, K_DIST, DIST_PB, L.dist, DIST_CAP);
}

// output bytes [x0, x1) of the block: ring -> dst (ring offset == destination address mod 16: 16-byte body, byte stores for ragged ends)
BZ_HD void ring_flush(const Lds &L, uint32_t align, uint8_t *dst, uint32_t x0, uint32_t x1)
{
    const int lane = BZ_LANE;
    if (x1 <= x0) return;
    BZ_LDS_FENCE();
    const uint32_t n = x1 - x0;
    const uint32_t mis = (align + x0) & 15u;
    uint32_t head = mis ? 16u - mis : 0u; if (head > n) head = n;
    for (uint32_t i = (uint32_t)lane; i < head; i += BZ_NL) dst[x0 + i] = L.ring[(align + x0 + i) & RING_MASK];
    const uint32_t body = (n - head) >> 4;
    for (uint32_t i = (uint32_t)lane; i < body; i += BZ_NL) {
        const uint32_t x = x0 + head + 16u * i;
        uint32_t v[4];
        __builtin_memcpy(v, &L.ring[(align + x) & RING_MASK], 16);       // (a 16-byte piece never wraps: ring size and pieces are 16-aligned)
        __builtin_memcpy(dst + x, v, 16);
    }
    const uint32_t done = head + (body << 4);
    for (uint32_t i = done + (uint32_t)lane; i < n; i += BZ_NL) dst[x0 + i] = L.ring[(align + x0 + i) & RING_MASK];
}

// One BGZF block: in[0, in_len) deflate data (readable to in + in_len + 8 + 520: the input window runs ahead) -> dst[0, isize), where
// align = dst's address mod 16.  Returns ST_OK when the final block ended exactly at isize bytes (everything is then in dst).
BZ_HD int inflate_block(Lds &L, const Consts &C, const uint8_t *in, int64_t in_len, uint32_t isize, uint32_t align, uint8_t *dst)
{
    const int lane = BZ_LANE;
    Bits b; b.buf = 0; b.cnt = 0; b.next = 0; b.in = in; b.in_len = in_len;
    InWin W; W.base = -1;
    uint32_t o = 0;                  // output bytes so far
    uint32_t flushed = 0;            // of them, already in dst: everything below the last finished ring page
    // pages of the ring are finished when the output passes their end: (align + x) >> PAGE_SHIFT changes
#define BZ_FLUSH_PAGES() do { const uint32_t pg_ = ((align + o) >> PAGE_SHIFT) << PAGE_SHIFT; if (pg_ > align + flushed) { ring_flush(L, align, dst, flushed, pg_ - align); flushed = pg_ - align; } } while (0)
    for (;;) {
        if (b.next > in_len + 8) return ST_INPUT;
        refill(b, W);
        const unsigned final_block = take(b, 1), type = take(b, 2);
        if (type == 0) {
            // 3.2.4 stored: skip to the byte boundary, LEN, NLEN, the bytes
            drop(b, b.cnt & 7);
            refill(b, W);
            const unsigned len = take(b, 16), nlen = take(b, 16);
            if ((len ^ nlen) != 0xffffu) return ST_BAD_STREAM;
            // whole bytes still in the buffer go back to the input
            int64_t p = b.next - (b.cnt >> 3);
            b.buf = 0; b.cnt = 0;
            if (p + (int64_t)len > in_len || o + len > isize) return ST_BAD_STREAM;
            for (unsigned done = 0; done < len;) {          // a page at a time: the ring is smaller than a stored block may be
                unsigned n = len - done; if (n > (1u << PAGE_SHIFT)) n = 1u << PAGE_SHIFT;
                for (unsigned i = (unsigned)lane; i < n; i += BZ_NL) L.ring[(align + o + i) & RING_MASK] = in[p + done + i];
                o += n; done += n;
                BZ_FLUSH_PAGES();
            }
            b.next = p + len;
        } else if (type == 1 || type == 2) {
            int rc = type == 1 ? fixed_tables(L, C) : read_dynamic(L, C, b, W);
            if (rc) return rc;
            for (;;) {
                if (b.cnt < 48) { if (b.next > in_len + 8) return ST_INPUT; refill(b, W); }
                uint32_t e = lookup(b, L.lit, LIT_PB);
                const unsigned op = (e >> 8) & 0xff;
                if (op == 0) {
                    if (o >= isize) return ST_SIZE;
                    if (lane == 0) L.ring[(align + o) & RING_MASK] = (uint8_t)(e >> 16);
                    ++o;
                    if (((align + o) & ((1u << PAGE_SHIFT) - 1)) == 0) BZ_FLUSH_PAGES();
                    continue;
                }
                if (op & OP_EOB) break;
                if (!(op & OP_BASE)) return ST_BAD_STREAM;
                const unsigned len = (e >> 16) + take(b, (int)(op & 15));
                const uint32_t d = lookup(b, L.dist, DIST_PB);
                const unsigned dop = (d >> 8) & 0xff;
                if (!(dop & OP_BASE)) return ST_BAD_STREAM;
                const unsigned dist = (d >> 16) + take(b, (int)(dop & 15));
                if (dist > o || len > isize - o) return ST_BAD_STREAM;
                BZ_LDS_FENCE();
                const uint32_t s0 = align + o - dist, d0 = align + o;       // ring positions before masking
                if (dist + len > (unsigned)RING) {
                    // further back than the ring holds: those bytes are in dst already (dist + len > RING puts the whole source below the
                    // last finished page; it cannot overlap the copy).  Same wave, program order: the flush stores are visible to these loads
                    BZ_MEM_FENCE();
                    const uint8_t *g = dst + (o - dist);
                    for (unsigned i = (unsigned)lane; i < len; i += BZ_NL) { const uint8_t v = g[i]; L.ring[(d0 + i) & RING_MASK] = v; }
                } else if (dist >= len) {
                    // no overlap
                    for (unsigned i = (unsigned)lane; i < len; i += BZ_NL) { const uint8_t v = L.ring[(s0 + i) & RING_MASK]; L.ring[(d0 + i) & RING_MASK] = v; }
                } else if (dist >= (unsigned)BZ_NL) {
                    // lanes of one pass never read what the same pass writes; a later pass reads what an earlier one wrote, so the passes
                    // are separated (the wave runs them in order anyway; the fence says so to the compiler and to the CPU emulation of
                    // tests/cpu/hipemu, where the lanes of a wave only meet at wave-level operations)
                    for (unsigned i0 = 0; i0 < len; i0 += BZ_NL) {
                        const unsigned i = i0 + (unsigned)lane;
                        if (i < len) { const uint8_t v = L.ring[(s0 + i) & RING_MASK]; L.ring[(d0 + i) & RING_MASK] = v; }
                        BZ_LDS_FENCE();
                    }
                } else if (dist == 1) {
                    const uint8_t v = L.ring[s0 & RING_MASK];
                    for (unsigned i = (unsigned)lane; i < len; i += BZ_NL) L.ring[(d0 + i) & RING_MASK] = v;
                } else {
                    // an overlapping copy repeats the dist bytes in front of it
                    const uint32_t inv = C.inv20[dist];
                    for (unsigned i = (unsigned)lane; i < len; i += BZ_NL) { const unsigned q = (i * inv) >> 20; const uint8_t v = L.ring[(s0 + i - q * dist) & RING_MASK]; L.ring[(d0 + i) & RING_MASK] = v; }
                }
                const uint32_t page_before = (align + o) >> PAGE_SHIFT;
                o += len;
                BZ_LDS_FENCE();
                if (((align + o) >> PAGE_SHIFT) != page_before) BZ_FLUSH_PAGES();
            }
        } else return ST_BAD_STREAM;
        if (final_block) break;
    }
#undef BZ_FLUSH_PAGES
    if (b.next - (b.cnt >> 3) > in_len) return ST_INPUT;         // the stream needed bytes beyond its end
    if (o != isize) return ST_SIZE;
    ring_flush(L, align, dst, flushed, o);
    return ST_OK;
}

}  // namespace bgzi
