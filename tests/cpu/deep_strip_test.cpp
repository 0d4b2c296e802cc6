// deep_strip_test.cpp -- samtools_amd/csrc/deep_strip.h against the straightforward indexing (what k_mplp_emit_deep did per column
// before the per-block shift): random reads, every (first covered column d, first query index parity), reference codes with and
// without matches.  Test infrastructure.
#include "../../samtools_amd/csrc/deep_strip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

int main()
{
    std::mt19937_64 rng(12345);
    long checked = 0;
    for (int it = 0; it < 200000; ++it) {
        const int lq = 40 + (int)(rng() % 200);
        std::vector<uint8_t> qual((size_t)lq + 64, 0), seq((size_t)(lq + 64) / 2 + 16, 0);
        for (int i = 0; i < lq; ++i) qual[(size_t)i] = (uint8_t)(rng() % 94);
        for (size_t i = 0; i < seq.size(); ++i) seq[i] = (uint8_t)(rng() & 0xff);
        const int d = (int)(rng() % 16);
        // d > 0: the read starts inside the strip, qb is the query index of its first aligned base in the op (any value);
        // d == 0: the read covers column 0 with query index qb
        const int qb = (int)(rng() % (uint64_t)(lq - 17 > 1 ? lq - 17 : 1));
        uint32_t q4[4], s4[3];
        memcpy(q4, &qual[(size_t)qb], 16);
        memcpy(s4, &seq[(size_t)(qb >> 1)], 12);
        uint64_t rbpack = 0; int rb[16];
        const bool has_ref = (rng() & 3) != 0;
        for (int k = 0; k < 16; ++k) {
            const int qk = qb + (k - d);
            int code = (int)(rng() % 16);
            if (k >= d && (rng() & 1)) code = (seq[(size_t)(qk >> 1)] >> ((~qk & 1) << 2)) & 15;      // force matches half the time
            rb[k] = code; rbpack |= (uint64_t)code << (4 * k);
        }
        uint32_t qs[4];
        deep_shift_quals(q4, d, qs);
        const uint64_t nib = deep_shift_bases(s4, qb, d, rbpack, has_ref);
        for (int k = d; k < 16; ++k) {
            const int qk = qb + (k - d);
            const int want_q = qual[(size_t)qk];
            int want_b = (seq[(size_t)(qk >> 1)] >> ((~qk & 1) << 2)) & 15;
            if (has_ref && want_b == rb[k]) want_b = 0;
            const int got_q = (int)((qs[k >> 2] >> (8 * (k & 3))) & 255u), got_b = (int)((nib >> (4 * k)) & 15u);
            if (got_q != want_q || got_b != want_b) {
                printf("MISMATCH it=%d d=%d qb=%d k=%d: qual %d/%d base %d/%d\n", it, d, qb, k, got_q, want_q, got_b, want_b);
                return 1;
            }
            ++checked;
        }
    }
    printf("deep_strip_test OK: %ld (read, column) pairs\n", checked);
    return 0;
}
