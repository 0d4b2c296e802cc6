// tests/cpu/baq_emul.cpp -- TEST INFRASTRUCTURE: the lane functions of samtools_amd/csrc/baq_band7s.h (what every lane of the
// device kernel k_baq7s executes) run on the CPU, one read at a time, against the oracle's restatement of HTSlib's
// sam_prob_realn() (oracle/o_baq.c), on generated class-S reads: random lengths, soft clips, substitutions, ambiguous bases in
// the read and in the reference, the whole quality range, extended and per-base mode.  The product has no CPU path: this
// harness is linked only by tests/test_baq_emul.py.
//
//   clang++ -O1 -std=c++17 -ffp-contract=off -I samtools_amd/csrc tests/cpu/baq_emul.cpp o_baq.o o_io.o -lz -lm -o baq_emul
//   baq_emul <n_reads> <seed> [force_edge] [mode]
// prints "reads N changed C mismatching_reads X" and exits 1 when X > 0.  mode: the kernel's STA_BAQ7S_MODE (16: the MAP quality from the
// threshold table, the default build; 0: from the formula).
//   baq_emul logtab <n_random> <seed>
// checks the threshold table on its own: clean steps over 2 x 10^5 doubles either side of every threshold, table == formula on random
// posteriors (uniform, near 1, near the thresholds) and on the special values.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

extern "C" {
#include "../../oracle/o_common.h"
int o_prob_realn(orec_t *b, const char *ref, hpos_t ref_len, int flag);
}

static bool g_force_edge = false;
#define BQS_TEST_FORCE_EDGE g_force_edge
#include "baq_band7s.h"

static int check_logtab(long n_random, unsigned seed)
{
    baq7s::LogTab T;
    if (!baq7s::make_log_thresholds(T, 200000)) { printf("logtab: NOT a clean step function around a threshold\n"); return 1; }
    std::mt19937_64 rng(seed);
    long bad = 0, n = 0;
    auto one = [&](double zs, double sum) {
        const int a = baq7s::map_quality_formula(zs, sum), b = baq7s::map_quality(zs, sum, (const double *)T.t);
        ++n;
        if (a != b) { if (bad < 10) fprintf(stderr, "logtab mismatch zs %.17g sum %.17g formula %d table %d\n", zs, sum, a, b); ++bad; }
    };
    // special values: posterior 1 (x = 0), 0, NaN (0 / 0), tiny, the largest posterior below 1
    one(1., 1.); one(0., 1.); one(0., 0.); one(1e-300, 1.); one(1. - 0x1p-53, 1.); one(0.5, 1.); one(3., 3.); one(1e-310, 1e-310);
    // every threshold and its neighbours, as x = 1 - mx cannot be set directly: mx = 1 - x is exact for x >= 2^-53 multiples; walk mx
    for (int k = 1; k <= 101; ++k) {
        const double x = T.t[k];
        for (int d = -300; d <= 300; ++d) {
            double mx = 1. - x;                 // rounds; then walk the neighbouring posteriors
            uint64_t u; memcpy(&u, &mx, 8); u += (uint64_t)(int64_t)d; memcpy(&mx, &u, 8);
            if (mx >= 0. && mx <= 1.) one(mx, 1.);
        }
    }
    std::uniform_real_distribution<double> U(0., 1.);
    for (long i = 0; i < n_random; ++i) {
        const double sum = ldexp(U(rng) + .5, (int)(rng() % 40) - 20);
        double mx;
        switch (i % 4) {
        case 0: mx = U(rng); break;
        case 1: mx = 1. - ldexp(U(rng), -(int)(rng() % 54)); break;          // near 1: the whole range of x
        case 2: { const int k = 1 + (int)(rng() % 101); mx = 1. - T.t[k] * (1. + (U(rng) - .5) * 1e-9); break; }
        default: mx = ldexp(U(rng), -(int)(rng() % 30)); break;
        }
        if (mx < 0.) mx = 0.; if (mx > 1.) mx = 1.;
        one(mx * sum, sum);
    }
    printf("logtab checked %ld mismatches %ld\n", n, bad);
    return bad ? 1 : 0;
}

template <int MODE>
static void run_lanes(const baq7s::Par &par, int lq, int l_ref, bool amb, std::vector<uint32_t> &IN, std::vector<baq7s::d2> &F2, std::vector<double> &S,
                      const float *q2p, baq7s::BwdCtx &ctx)
{
    baq7s::d2 Ln[baq7s::NB];                      // what is LDS on the device: the normalised middle row of the group in work
    baq7s::fwd_lane<1, MODE>(par, lq, amb, IN.data(), F2.data(), S.data(), 0, q2p);
    baq7s::bwd_lane<1, MODE>(par, lq, l_ref, amb, IN.data(), F2.data(), S.data(), 0, q2p, &Ln[0], ctx);
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "logtab")) return check_logtab(argc > 2 ? atol(argv[2]) : 1000000, argc > 3 ? (unsigned)atoi(argv[3]) : 1);
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    baq7s::LogTab logtab;
    if (!baq7s::make_log_thresholds(logtab)) { fprintf(stderr, "log thresholds: not a clean step function\n"); return 2; }
    const int n_reads = argc > 1 ? atoi(argv[1]) : 2000;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    g_force_edge = argc > 3 && atoi(argv[3]) != 0;
    std::mt19937 rng(seed);
    auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };

    const int L = 20000;
    std::string ref(L, 'A');
    for (int i = 0; i < L; ++i) ref[i] = "ACGT"[rnd(4)];
    for (int k = 0; k < 12; ++k) { int p = rnd(L - 40), n = 1 + rnd(3); for (int i = 0; i < n; ++i) ref[p + i] = "NnRYM"[rnd(5)]; }   // ambiguous reference bases
    for (int i = 0; i < L; i += 7) if (rnd(5) == 0) ref[i] = (char)tolower(ref[i]);

    float q2p[256]; uint8_t refc[256];
    for (int i = 0; i < 256; ++i) { q2p[i] = (float)pow(10, -i / 10.); refc[i] = (uint8_t)nt16_int[nt16_table[i]]; }

    long n_changed = 0, n_bad = 0, n_class = 0, n_amb = 0;
    for (int it = 0; it < n_reads; ++it) {
        const int lq = (it % 7 == 0) ? 16 + rnd(241) : (it % 3 == 0 ? 150 : (rnd(2) ? 151 : 100 + rnd(60)));
        int s5 = rnd(4) == 0 ? rnd(lq / 3) : 0, s3 = rnd(4) == 0 ? rnd(lq / 3) : 0;
        if (lq - s5 - s3 < 1) { s5 = s3 = 0; }
        const int mlen = lq - s5 - s3;
        const long long pos = (it % 50 == 0) ? rnd(12) : (it % 51 == 0 ? L - mlen - rnd(12) : 20 + rnd(L - lq - 60));
        std::vector<uint32_t> cigar;
        if (rnd(8) == 0) cigar.push_back((uint32_t)(3 << 4) | 5);
        if (s5) cigar.push_back((uint32_t)(s5 << 4) | 4);
        cigar.push_back((uint32_t)(mlen << 4) | (rnd(10) == 0 ? 7 : 0));
        if (s3) cigar.push_back((uint32_t)(s3 << 4) | 4);
        if (rnd(8) == 0) cigar.push_back((uint32_t)(2 << 4) | 5);
        // the read: the reference under the M operation with substitutions (sometimes a stretch of them, sometimes a shifted copy:
        // both make the MAP path leave the diagonal), random bases in the clips, a few N
        std::vector<uint8_t> bases(lq), qual(lq);
        const int shift = rnd(6) == 0 ? rnd(7) - 3 : 0;
        for (int i = 0; i < lq; ++i) {
            long long rp = pos + (i - s5) + (i > lq / 2 ? shift : 0);
            char c = (i >= s5 && i < s5 + mlen && rp >= 0 && rp < L) ? (char)toupper(ref[(size_t)rp]) : "ACGT"[rnd(4)];
            if (rnd(100) < ((it % 5 == 0) ? 15 : 1)) c = "ACGT"[rnd(4)];
            if (rnd(300) == 0) c = 'N';
            int code = c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15;
            bases[i] = (uint8_t)code;
            const int kind = rnd(20);
            qual[i] = (uint8_t)(kind == 0 ? rnd(94) : kind == 1 ? 0 : kind == 2 ? 93 : (int[]){ 2, 11, 25, 37, 37, 37, 37, 40 }[rnd(8)]);
        }
        if (qual[0] == 0xff) qual[0] = 30;
        std::vector<uint8_t> seq((lq + 1) / 2 + 8, 0);
        for (int i = 0; i < lq; ++i) seq[i >> 1] |= (uint8_t)(bases[i] << ((~i & 1) << 2));
        const bool plain = it % 4 == 3;

        baq7s::Shape sh = baq7s::classify(cigar.data(), (int)cigar.size(), pos, lq, L);
        if (!sh.ok) continue;                      // (clipped windows, too short: the general kernels' business)
        ++n_class;

        // oracle
        std::vector<uint8_t> oq(qual);
        orec_t rec; memset(&rec, 0, sizeof(rec));
        rec.pos = pos; rec.flag = 0; rec.l_qseq = lq; rec.n_cigar = (uint32_t)cigar.size(); rec.cigar = cigar.data();
        rec.seq = seq.data(); rec.qual = oq.data(); rec.aux = NULL; rec.l_aux = 0;
        o_prob_realn(&rec, ref.c_str(), L, plain ? 1 : 3);

        // lane functions, one lane (LS = 1)
        const int l_ref = lq + 6;
        std::vector<uint32_t> IN(lq + 2, 0);
        std::vector<baq7s::d2> F2((size_t)((lq + 2) / 3) * baq7s::NB);
        std::vector<double> S(lq + 2, 0.);
        std::vector<uint8_t> mq(qual); mq.resize((size_t)(lq + 7) & ~(size_t)7, 0xee);      // the device pools pad every read to 8 bases
        const baq7s::Par par = baq7s::make_par(lq, l_ref);
        const bool amb = baq7s::pack_lane<1>(lq, l_ref, mq.data(), seq.data(), ref.c_str() + sh.xb, refc, IN.data(), 0);
        if (amb) ++n_amb;
        baq7s::BwdCtx ctx; ctx.ys = sh.ys; ctx.mlen = sh.mlen; ctx.run_r = 0; ctx.plain_mask = plain ? -1 : 0; ctx.LT = logtab.t;
        switch (mode) {
        case 0: run_lanes<0>(par, lq, l_ref, amb, IN, F2, S, q2p, ctx); break;
        case 16: run_lanes<16>(par, lq, l_ref, amb, IN, F2, S, q2p, ctx); break;
        default: fprintf(stderr, "mode %d is not built into the harness\n", mode); return 2;
        }
        baq7s::final_lane<1>(lq, IN.data(), 0, ctx, mq.data());

        if (memcmp(oq.data(), qual.data(), (size_t)lq) != 0) ++n_changed;
        for (size_t i = (size_t)lq; i < mq.size(); ++i) if (mq[i] != 0xee) { fprintf(stderr, "read %d: padding byte %zu was written\n", it, i); ++n_bad; break; }
        if (memcmp(oq.data(), mq.data(), (size_t)lq) != 0) {
            if (n_bad < 5) {
                fprintf(stderr, "MISMATCH read %d lq %d pos %lld s5 %d s3 %d plain %d\n", it, lq, pos, s5, s3, (int)plain);
                for (int i = 0; i < lq; ++i) if (oq[i] != mq[i]) fprintf(stderr, "  q[%d]: in %d oracle %d lanes %d\n", i, qual[i], oq[i], mq[i]);
            }
            ++n_bad;
        }
    }
    printf("reads %ld changed %ld ambiguous_windows %ld mismatching_reads %ld\n", n_class, n_changed, n_amb, n_bad);
    return n_bad ? 1 : 0;
}
