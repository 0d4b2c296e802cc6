// plp_swar_test.cpp -- the packed-byte helpers of samtools_amd/csrc/plp_tile.h against plain per-byte loops: every byte value for
// the compare / quality-character helpers, random reads for tile_convert16 (every first covered column, both strands, with and
// without a reference, '^' / '$' flags).  Test infrastructure.
#define PLP_WAVE_ANY(x) (true)
#include "../../samtools_amd/csrc/plp_tile.h"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

thread_local char *plp_host_lds = nullptr;

int main()
{
    long checked = 0;
    // swar_ge_u8 / swar_qual_chars / swar_expand80 / swar_byte_range: all byte values in every byte position beside random neighbours
    std::mt19937_64 rng(99);
    for (int m = 0; m <= 127; ++m)
        for (int v = 0; v < 256; ++v)
            for (int pos = 0; pos < 4; ++pos) {
                uint32_t x = (uint32_t)rng();
                x = (x & ~(0xffu << (8 * pos))) | ((uint32_t)v << (8 * pos));
                const uint32_t f = swar_ge_u8(x, (uint32_t)m * 0x01010101u), e = swar_expand80(f), qc = swar_qual_chars(x);
                for (int i = 0; i < 4; ++i) {
                    const int b = (x >> (8 * i)) & 255;
                    const bool ge = b >= m;
                    if ((((f >> (8 * i)) & 255) == 0x80) != ge || (((e >> (8 * i)) & 255) == 0xff) != ge) { printf("swar_ge_u8 m=%d byte=%d\n", m, b); return 1; }
                    const int want = b + 33 < 126 ? b + 33 : 126;
                    if ((int)((qc >> (8 * i)) & 255) != want) { printf("swar_qual_chars byte=%d got %d want %d\n", b, (int)((qc >> (8 * i)) & 255), want); return 1; }
                    ++checked;
                }
            }
    for (int lo = -3; lo <= 6; ++lo)
        for (int hi = -3; hi <= 6; ++hi) {
            const uint32_t r = swar_byte_range(lo, hi);
            for (int i = 0; i < 4; ++i) if ((((r >> (8 * i)) & 255) == 0xff) != (i >= lo && i < hi)) { printf("swar_byte_range %d %d\n", lo, hi); return 1; }
        }
    for (int c = 0; c < 16; ++c)
        for (int rev = 0; rev < 2; ++rev) {
            const uint32_t r = swar_base_chars((uint32_t)c * 0x01010101u, rev != 0);
            const char want = rev ? ",acmgrsvtwyhkdbn"[c] : ".ACMGRSVTWYHKDBN"[c];
            for (int i = 0; i < 4; ++i) if ((char)((r >> (8 * i)) & 255) != want) { printf("swar_base_chars %d %d\n", c, rev); return 1; }
        }
    // tile_convert16
    for (int it = 0; it < 300000; ++it) {
        const int lq = 20 + (int)(rng() % 200);
        std::vector<uint8_t> qual((size_t)lq + 64, 0), seq((size_t)(lq + 64) / 2 + 16, 0);
        for (int i = 0; i < lq; ++i) qual[(size_t)i] = (uint8_t)((rng() & 15) == 0 ? rng() % 256 : rng() % 50);
        for (size_t i = 0; i < seq.size(); ++i) seq[i] = (uint8_t)(rng() & 0xff);
        const int minq = (int)(rng() % 3 == 0 ? rng() % 128 : 13);
        const bool rev = rng() & 1, has_ref = (rng() & 3) != 0;
        // the read occupies chunk columns [d0, d0 + ncov); column d0 shows query index qb
        const int d0 = (int)(rng() % 16);
        const int qb = d0 > 0 ? 0 : (int)(rng() % (uint64_t)lq);            // a read that starts inside the chunk starts with query index 0
        int ncov = 16 - d0; if (ncov > lq - qb) ncov = lq - qb;
        if (rng() & 1) ncov = 1 + (int)(rng() % (uint64_t)ncov);
        const int head_col = (qb == 0 && (rng() & 1)) ? d0 : -1;
        const int tail_col = (qb + ncov == lq && (rng() & 1)) ? d0 + ncov - 1 : -1;
        uint32_t q4[4], s4[3];
        memcpy(q4, &qual[(size_t)qb], 16); memcpy(s4, &seq[(size_t)(qb >> 1)], 12);
        uint64_t rbpack = 0; int rb[16];
        for (int k = 0; k < 16; ++k) {
            const int qk = qb + (k - d0);
            int code = (int)(rng() % 16);
            if (k >= d0 && k < d0 + ncov && (rng() & 1)) code = (seq[(size_t)(qk >> 1)] >> ((~qk & 1) << 2)) & 15;
            rb[k] = code; rbpack |= (uint64_t)code << (4 * k);
        }
        uint32_t tb[4], tq[4];
        tile_convert16(q4, s4, qb, d0, ncov, rbpack, has_ref, (uint32_t)minq * 0x01010101u, rev, head_col, tail_col, tb, tq);
        for (int k = 0; k < 16; ++k) {
            int want_b = 0, want_q = 0;
            if (k >= d0 && k < d0 + ncov) {
                const int qk = qb + (k - d0), q = qual[(size_t)qk];
                if (q >= minq) {
                    int c = (seq[(size_t)(qk >> 1)] >> ((~qk & 1) << 2)) & 15;
                    if (has_ref && c == rb[k]) c = 0;
                    want_b = (unsigned char)(rev ? ",acmgrsvtwyhkdbn"[c] : ".ACMGRSVTWYHKDBN"[c]);
                    want_q = q + 33 < 126 ? q + 33 : 126;
                    if (k == head_col) want_q |= 0x80;
                    if (k == tail_col) want_b |= 0x80;
                }
            }
            const int got_b = (int)((tb[k >> 2] >> (8 * (k & 3))) & 255u), got_q = (int)((tq[k >> 2] >> (8 * (k & 3))) & 255u);
            if (got_b != want_b || got_q != want_q) { printf("tile_convert16 it=%d d0=%d qb=%d ncov=%d k=%d: base %d/%d qual %d/%d\n", it, d0, qb, ncov, k, got_b, want_b, got_q, want_q); return 1; }
            ++checked;
        }
    }
    printf("plp_swar_test OK: %ld bytes\n", checked);
    return 0;
}
