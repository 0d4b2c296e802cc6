// cons_core.h -- per-read and per-column arithmetic of the consensus path, shared by the HIP kernels of
// kernels_cons.hip (device) and by the CPU harness under tests/cpu (host, test infrastructure: it lets the not-gpu suite diff
// this logic against the oracle and the reference's goldens without a device; the product never runs it on the host).
//
// What it replaces (reference file:line):
//   get_next_base                 consensus_pileup.c:69-286    Cursor::step  (one read, one (position, nth) column)
//   pileup_loop's column sequence consensus_pileup.c:373-451   read_shape() + the insertion-column counts of kernels_cons.hip
//   nm_init, homopoly_qual_fix    bam_consensus.c:943-973, 1012-1206   read_prepare()
//   nm_local, poly_len            bam_consensus.c:978-1000     nm_word()
//   calculate_consensus_simple    bam_consensus.c:1907-2014    SimpleAcc
//   calculate_consensus_gap5      bam_consensus.c:1258-1793    Gap5Acc (K2 / DO_* blocks are compiled out in the reference)
//   calculate_consensus_gap5m     bam_consensus.c:1799-1880    mix_calls()
//   consensus_base                bam_consensus.c:2139-2183    final_call()
//   consensus_init + tab.h        bam_consensus.c:740-883, bam_consensus_tab.h   build_tables() (host only, libm)
//
// Column model: a column is (reference position, nth) with nth > 0 for the inserted bases after that position; a position
// has as many extra columns as the longest insertion/pad run any read alive there carries.  A read is alive from its start
// to the column holding its last CIGAR-consumed base.  Per-column sums run over the alive reads in file order, one read
// after the other, so the fp64 additions happen in the reference's order.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CONS_HD __host__ __device__ __forceinline__
#else
#define CONS_HD inline
#endif

namespace cons {

enum { MODE_SIMPLE = 0, MODE_BAYES_116 = 1, MODE_RECALL = 2, MODE_PRECISE = 3, MODE_MIXED = 4 };

// plain-data parameters the kernels consume (subset of consensus_opts, bam_consensus.c:211-260)
struct Par {
    int32_t mode, use_qual, min_qual, adj_qual, use_mqual, nm_adjust, nm_halo, sc_cost, low_mqual, high_mqual, min_depth;
    int32_t cons_cutoff, ambig, default_qual, excl_flags, incl_flags, min_mqual, homopoly_on;
    double scale_mqual, call_fract, het_fract, homopoly_fix;
};

struct Probs {
    double lprior15[15];
    double pMM[101], pxx[101], pxM[101], pox[101], poM[101], poo[101], puu[101], pum[101], pmm[101];
    double poly_mul;
};
struct Tables {
    double e_tab[1001];          // exp(i), i = -500 .. 500
    double e_tab2[1001];         // exp(i / 10)
    double q2p[101], mqual_pow_1m[256], ph2err[256];
    Probs recall, precise;
};

// ---- entry words: what one read shows in one column ----
#define CONS_E_BASE4(e) ((int)((e) & 31u))
#define CONS_E_QUAL(e) ((int)(((e) >> 5) & 255u))
#define CONS_E_REFSKIP 0x2000u       // pileup_t.ref_skip (also set on the bases either side of a reference skip)
#define CONS_E_REV 0x4000u
#define CONS_E_SKIPCOL 0x8000u       // the column lies inside an N operation: pileup_t.base == '.'
#define CONS_E_PAD 0x10000u          // pileup_t.padding: the entry is a pad opposite another read's insertion
#define CONS_NM_FLAT 0x80000000u     // nm word: index fell off the read, nm_local() does not divide by 10 and poly_len() is 0

struct ReadView {
    int32_t start;               // leftmost coordinate relative to the window origin
    int32_t l_qseq, n_cigar;
    const uint32_t *cigar;
    const uint8_t *seq;          // 4-bit packed, first base of this read at seq[0] high nibble
    const uint8_t *qual;
};

CONS_HD int seqi(const uint8_t *s, int i) { return (s[i >> 1] >> ((~i & 1) << 2)) & 0xf; }

// ---- shape of a read in column space ----
struct Shape {
    int32_t last;                // last reference position it consumes (start - 1 if it consumes none)
    int32_t tail_run;            // inserted / pad columns after `last` that belong to it
    int32_t bad_op;              // an op outside MIDNSHP=X was seen
};
// Walks the CIGAR once; every inserted / pad run that follows a reference-consuming base is reported through
// on_run(position, run length).  Leading insertions are part of the clip (consensus_pileup.c:56-59); a soft clip ends a run
// and hides any insertion behind it until the next reference base (the pos loop of get_next_base skips I and S alike).
template <class F> CONS_HD Shape read_shape(const ReadView &r, F on_run)
{
    Shape s; s.bad_op = 0;
    int32_t ref = r.start, run = 0, tail = 0;
    bool seen_ref = false, blocked = false;
    for (int k = 0; k < r.n_cigar; ++k) {
        const int op = (int)(r.cigar[k] & 15u);
        const int32_t len = (int32_t)(r.cigar[k] >> 4);
        if (len == 0) continue;
        if (op == 0 || op == 7 || op == 8 || op == 2 || op == 3) {
            if (run > 0) { on_run(ref - 1, run); run = 0; }
            ref += len; seen_ref = true; blocked = false; tail = 0;
        } else if (op == 1 || op == 6) {
            if (seen_ref && !blocked) run += len;
        } else if (op == 4) {
            if (run > 0) { on_run(ref - 1, run); tail = run; run = 0; }
            blocked = true;
        } else if (op != 5) { s.bad_op = 1; break; }
    }
    if (run > 0) { on_run(ref - 1, run); tail = run; }
    s.last = ref - 1; s.tail_run = tail;
    return s;
}

// ---- the column cursor of one read ----
struct Cursor {
    int32_t pos;                 // reference position of the last consumed column (relative)
    int32_t nth, seq_off, cig_ind, cig_op, cig_len, eof, qual, base4;
    bool ref_skip, skipcol, pad;

    CONS_HD void init(int32_t start)
    {
        pos = start - 1; nth = 0; seq_off = -1; cig_ind = 0; cig_op = -1; cig_len = 0; eof = 0; qual = 0; base4 = 0;
        ref_skip = false; skipcol = false; pad = false;
    }
    CONS_HD bool take(const ReadView &r)
    {
        if (cig_ind >= r.n_cigar) return false;
        cig_op = (int)(r.cigar[cig_ind] & 15u); cig_len = (int32_t)(r.cigar[cig_ind] >> 4); cig_ind++;
        return true;
    }
    // qualities are read one past the last base in two corners of the reference (b_qual[seq_offset+1]); the staged pool has
    // no aux block behind it, so that byte reads as 0 here (DESIGN.md lists the corner)
    CONS_HD int qat(const ReadView &r, int i) const { return i >= 0 && i < r.l_qseq ? r.qual[i] : 0; }

    // state at column (p, n); ins = inserted bases the read still has at this position.  1 fetched, 0 ran off, -1 bad op
    CONS_HD int step(const ReadView &r, int32_t p, int32_t n, int32_t &ins)
    {
        int op = cig_op;
        ins = 0;
        while (pos < p) {
            nth = 0;
            if (cig_len == 0) { if (!take(r)) { eof = 1; return 0; } op = cig_op; }
            const bool al = op == 0 || op == 7 || op == 8;
            if (al && cig_len <= p - pos) { seq_off += cig_len; pos += cig_len; cig_len = 0; }
            else if (al) { seq_off++; pos++; cig_len--; }
            else if (op == 2 || op == 3) { pos++; cig_len--; }
            else if (op == 1 || op == 4) { seq_off += cig_len; cig_len = 0; }
            else if (op == 6 || op == 5) cig_len = 0;
            else return -1;
        }
        while (nth < n) {
            if (cig_len == 0) { if (!take(r)) { eof = 1; return 0; } op = cig_op; }
            if (op == 1) { seq_off++; cig_len--; nth++; }
            else if (op == 6) { cig_len--; nth++; }
            else if (op == 5) cig_len = 0;
            else if (op <= 8) break;
            else return -1;
        }
        ref_skip = false; skipcol = false; pad = false;
        if (nth < n && op != 1) {                        // pad opposite somebody else's insertion
            base4 = 16; pad = true;
            if (seq_off < r.l_qseq) { const int q = qat(r, seq_off + 1); if (q < qual) qual = q; }
            else qual = 0;
        } else if (op == 2 || op == 6) {
            base4 = 16;
            const int q = seq_off + 1 < r.l_qseq ? qat(r, seq_off + 1) : qat(r, seq_off);
            if (q < qual) qual = q;
        } else if (op == 3) {
            base4 = 0; qual = 0; skipcol = true; ref_skip = true;
            eof = eof ? 2 : 3;
        } else if (seq_off < r.l_qseq) {
            qual = r.qual[seq_off];
            base4 = seqi(r.seq, seq_off);
        } else { base4 = 15; qual = 0xff; }
        if (eof && !skipcol) { ref_skip = true; eof = 0; }
        if (cig_len == 0) {
            if (take(r)) { op = cig_op; if (op == 3) { eof = 3; ref_skip = true; } }
            else eof = 1;
        }
        if (op == 6 || op == 1) ins = cig_len;
        else if (op == 4)
            eof = (cig_ind == r.n_cigar || (cig_ind + 1 == r.n_cigar && (int)(r.cigar[cig_ind] & 15u) == 5)) ? 1 : 0;
        else if (op == 5) eof = 1;
        return 1;
    }
    CONS_HD uint32_t entry(bool rev) const
    {
        return (uint32_t)base4 | ((uint32_t)(qual & 255) << 5) | (ref_skip ? CONS_E_REFSKIP : 0u) | (rev ? CONS_E_REV : 0u) | (skipcol ? CONS_E_SKIPCOL : 0u) | (pad ? CONS_E_PAD : 0u);
    }
};

// nm_local() / poly_len() argument is always pos + seq_offset + 1, i.e. query index seq_off + 1 (bam_consensus.c:1383,1417)
CONS_HD uint32_t nm_word(const int32_t *nm, int l_qseq, int seq_off)
{
    const int qi = seq_off + 1;
    if (qi >= l_qseq) return ((uint32_t)nm[l_qseq - 1] & 0xffffffu) | CONS_NM_FLAT;
    return (uint32_t)nm[qi];
}

CONS_HD double fast_log2(double val)
{
    union { double d; uint64_t x; } u; u.d = val;
    const int E = (int)((u.x >> 52) & 2047) - 1024;
    u.x &= ~(2047ULL << 52);
    u.x += 1023ULL << 52;
    val = ((-1 / 3.) * u.d + 2) * u.d - 2 / 3.;
    return E + val;
}
CONS_HD double ph_log(double x) { return -3.0103 * fast_log2(x); }
CONS_HD double fast_exp(const Tables &t, double y)
{
    if (y >= -50 && y <= 50) return t.e_tab2[500 + (int)(y * 10)];
    if (y < -500) y = -500;
    if (y > 500) y = 500;
    return t.e_tab[500 + (int)y];
}

struct Par; struct ReadView;
CONS_HD void read_prepare_md(const Par &o, const ReadView &r, const char *md, int md_len, int32_t *nm);

// ---- per-read preparation: nm[i] = homopolymer run << 24 | local edit cost; may rewrite the working qualities ----
// md / md_len: the MD:Z text (md_len = 0 or text not starting with a digit: no tag).  Returns 0 if the read leaves the pileup.
CONS_HD int read_prepare(const Par &o, const Tables &t, const ReadView &r, uint8_t *qual /* writable working copy */, const char *md, int md_len, int32_t *nm)
{
    const int qlen = r.l_qseq;
    if (qlen <= 0) return 0;
    for (int i = 0; i < qlen; ++i) nm[i] = 0;
    const double poly_adj = o.homopoly_on ? o.homopoly_fix : 1;
    const uint8_t *seq = r.seq;
    int i;
    if (o.adj_qual) {
        const int qhalo = 8, qhalop = 2;
        int qmin = qual[0], qminp = qual[0];
        int base = seqi(seq, 0), polyl = 0, polyr = 0;
        for (i = 1; i < qlen; i++) {
            if (seqi(seq, i) != base) break;
            if (i < qhalop && qminp > qual[i]) qminp = qual[i];
        }
        for (i = 0; i < qlen && i < qhalo; i++) if (qmin > qual[i]) qmin = qual[i];
        for (; i < qlen - qhalo; i++) {
            if (o.homopoly_on && seqi(seq, i) != base) {
                polyl = i; base = seqi(seq, i); qminp = qual[i];
                int j;
                for (j = i + 1; j < qlen; j++) {
                    if (seqi(seq, j) != base) break;
                    if (i < qhalop && qminp > qual[j]) qminp = qual[j];
                }
                polyr = j - 1;
            } else polyr = polyl;
            const int pl = polyr - polyl;
            const int tq = o.mode == MODE_BAYES_116 ? (qual[i] + 5 * qmin) / 4 : (int)(qual[i] / 3 + (qminp - pl * 2) * poly_adj);
            nm[i] += tq < qual[i] ? qual[i] - tq : 0;
            qminp = qual[i];
            const int k0 = polyl > i - qhalop ? polyl : i - qhalop, k1 = polyr < i + qhalop ? polyr : i + qhalop;
            for (int k = k0; k <= k1; k++) if (qminp > qual[k]) qminp = qual[k];
            if (qmin > qual[i + qhalo]) qmin = qual[i + qhalo];
            else if (qmin <= qual[i - qhalo]) {
                qmin = 99;
                for (int j = i - qhalo + 1; j <= i + qhalo; j++) if (qmin > qual[j]) qmin = qual[j];
            }
        }
        for (; i < qlen; i++) {
            const int tq = o.mode == MODE_BAYES_116 ? (qual[i] + 5 * qmin) / 4 : (int)(qual[i] / 3 + qminp * poly_adj);
            nm[i] += tq < qual[i] ? qual[i] - tq : 0;
        }
    }
    if (o.homopoly_on) {                                 // homopoly_qual_fix: average the outer pairs of every run
        for (i = 0; i < qlen; i++) {
            const int s = i, base = seqi(seq, i);
            while (i + 1 < qlen && seqi(seq, i + 1) == base) i++;
            for (int j = s, k = i; j < k; j++, k--) {
                const double e = t.ph2err[qual[j]] + t.ph2err[qual[k]];
                qual[j] = qual[k] = (uint8_t)(-fast_log2(e / 2) * 3.0104 + .49);
            }
        }
    }
    for (i = 0; i < qlen; i++) {
        const int base = seqi(seq, i);
        int j;
        for (j = i + 1; j < qlen; j++) if (seqi(seq, j) != base) break;
        int poly = j - i - 1; if (poly > 100) poly = 100;
        for (int k = i; k < j; k++) { const int cur = nm[k] >> 24; nm[k] = ((poly > cur ? poly : cur) << 24) | (nm[k] & 0xffffff); }
        i = j - 1;
    }
    read_prepare_md(o, r, md, md_len, nm);
    return 1;
}

// ---- the same preparation, one base at a time ----
// With homopolymer fixing off and outside the samtools-1.16 mode (the defaults) the sequential loops of nm_init collapse:
// polyl / polyr never move, so the "minimum quality of the run" a base is compared with is simply the previous base's
// quality (the initial value for the first tested base, the last main-loop value for the final 8 bases), and the window
// minimum qmin is not used at all.  That lets the device compute every base independently (lane per base, coalesced) instead
// of one lane walking one read.  The MD / soft-clip costs are added afterwards by read_prepare_md().
CONS_HD bool prepare_is_per_base(const Par &o) { return !o.homopoly_on && o.mode != MODE_BAYES_116; }

// nm words of the 8 bases i0 .. i0 + 7 (i0 a multiple of 8; rows of the staged pools are padded to 8 bases and start on 8-byte /
// 4-byte boundaries).  Everything the 8 bases need is loaded up front with independent loads -- their qualities as one 64-bit
// word, the packed sequence of bases i0 - 16 .. i0 + 23 as five 32-bit words -- and the per-base work runs on registers: the
// first 12 neighbours either side of a base are compared without data-dependent control flow, only a longer homopolymer falls
// into the loops, which search no further than the cap of 100 needs.
CONS_HD void prepare_granule(const Par &o, const ReadView &r, int i0, int32_t out[8])
{
    const int qlen = r.l_qseq;
    const uint8_t *qual = r.qual, *seq = r.seq;
    uint64_t q8; __builtin_memcpy(&q8, qual + i0, 8);
    const int qprev = i0 > 0 ? qual[i0 - 1] : 0;
    const int q0 = qual[0], q1 = qlen > 1 ? qual[1] : 0, s01 = seq[0];
    const int qtail = qlen > 16 ? qual[qlen - 9] : 0;
    const int n_words = (qlen + 7) >> 3, wc = i0 >> 3;
    uint32_t wd[5];
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int k = 0; k < 5; ++k) {
        int wi = wc - 2 + k; if (wi < 0) wi = 0; if (wi > n_words - 1) wi = n_words - 1;
        uint32_t x; __builtin_memcpy(&x, seq + (size_t)wi * 4, 4); wd[k] = x;
    }
    int qminp0 = q0;
    if (qlen > 1 && (s01 >> 4) == (s01 & 15) && qminp0 > q1) qminp0 = q1;
    // the window as three little-endian numbers with base t of the window in bits 4t .. 4t+3 (the packed sequence holds the
    // first base of a byte in its HIGH nibble: swap the nibbles of every byte)
#define CONS_SWAPN(x) ((((x) & 0x0f0f0f0fu) << 4) | (((x) >> 4) & 0x0f0f0f0fu))
    const uint64_t n0 = (uint64_t)CONS_SWAPN(wd[0]) | ((uint64_t)CONS_SWAPN(wd[1]) << 32);       // window bases 0 .. 15
    const uint64_t n1 = (uint64_t)CONS_SWAPN(wd[2]) | ((uint64_t)CONS_SWAPN(wd[3]) << 32);       // 16 .. 31
    const uint64_t n2 = (uint64_t)CONS_SWAPN(wd[4]);                                              // 32 .. 39
#undef CONS_SWAPN
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 8; ++b) {
        const int i = i0 + b;
        if (i >= qlen) { out[b] = 0; continue; }
        const int qi = (int)((q8 >> (8 * b)) & 255u), qim1 = b ? (int)((q8 >> (8 * (b - 1))) & 255u) : qprev;
        int adj = 0;
        if (o.adj_qual && i >= 8) {
            int qminp;
            if (qlen > 16) qminp = i < qlen - 8 ? (i == 8 ? qminp0 : qim1) : qtail;
            else qminp = qminp0;
            const int tq = qi / 3 + qminp;
            adj = tq < qi ? qi - tq : 0;
        }
        // run of equal codes around the base: XOR the window with the base in every nibble, mark the nibbles that differ, and
        // count the clear ones next to the base with clz / ctz (12 either side; the window holds bases i0 - 16 .. i0 + 23)
        const int rel = 16 + b;                                     // the base's nibble inside the window
        const int base = (int)((rel < 16 ? n0 >> (4 * rel) : n1 >> (4 * (rel - 16))) & 15u);
        const uint64_t rep = 0x1111111111111111ull * (uint64_t)base;
        const uint64_t x0 = n0 ^ rep, x1 = n1 ^ rep, x2 = n2 ^ (uint32_t)rep;
        const uint64_t d0 = (x0 | x0 >> 1 | x0 >> 2 | x0 >> 3) & 0x1111111111111111ull;      // bit 4k set: nibble k differs
        const uint64_t d1 = (x1 | x1 >> 1 | x1 >> 2 | x1 >> 3) & 0x1111111111111111ull;
        const uint64_t d2 = (x2 | x2 >> 1 | x2 >> 2 | x2 >> 3) & 0x11111111ull;
        // left: nibbles rel-12 .. rel-1 (inside nibbles 4 .. 22 of the window), the nearest one in the top nibble of a 48-bit field
        const int ls = 4 * (rel - 12);                              // 16 .. 44
        const uint64_t lf = ((d0 >> ls) | (d1 << (64 - ls))) & 0xffffffffffffull;
        int left = lf ? (int)(__builtin_clzll(lf << 16) >> 2) : 12;
        // right: nibbles rel+1 .. rel+12 (inside 17 .. 35): bits 4(rel+1) .. of d1:d2 counted from nibble 16
        const int rs = 4 * (rel + 1 - 16);                          // 4 .. 32
        const uint64_t rf = ((d1 >> rs) | (rs ? d2 << (64 - rs) : 0ull)) & 0xffffffffffffull;
        int right = rf ? (int)(__builtin_ctzll(rf) >> 2) : 12;
        if (left > i) left = i;                                    // nibbles outside the read are not part of it
        if (right > qlen - 1 - i) right = qlen - 1 - i;
        int lo = i - left, hi = i + right;
        if (left == 12) while (lo > 0 && i - lo < 101 && seqi(seq, lo - 1) == base) --lo;
        if (right == 12) while (hi + 1 < qlen && hi - lo < 101 && seqi(seq, hi + 1) == base) ++hi;
        int poly = hi - lo; if (poly > 100) poly = 100;
        out[b] = (poly << 24) | adj;
    }
}

// soft-clip and MD mismatch costs on top of nm[] (the tail of nm_init, bam_consensus.c:1138-1203)
CONS_HD void read_prepare_md(const Par &o, const ReadView &r, const char *md, int md_len, int32_t *nm)
{
    const int qlen = r.l_qseq;
    if (qlen <= 0 || md_len <= 0 || md[0] < '0' || md[0] > '9') return;
    const int halo = o.nm_halo;
    int i;
    const int op0 = (int)(r.cigar[0] & 15u), opn = (int)(r.cigar[r.n_cigar - 1] & 15u);
    if (op0 == 4 || (op0 == 5 && r.n_cigar > 1 && (int)(r.cigar[1] & 15u) == 4)) {
        for (i = 0; i < halo && i < qlen; i++) nm[i] += o.sc_cost;
        for (; i < halo * 2 && i < qlen; i++) nm[i] += o.sc_cost >> 1;
    }
    if (opn == 4 || (opn == 5 && r.n_cigar > 1 && (int)(r.cigar[r.n_cigar - 2] & 15u) == 4)) {
        for (i = qlen - 1; i >= qlen - halo && i >= 0; i--) nm[i] += o.sc_cost;
        for (; i >= qlen - halo * 2 && i >= 0; i--) nm[i] += o.sc_cost >> 1;
    }
    int pos = 0, m = 0;                                  // matched bases only: a substitution does not advance pos
    while (m < md_len && md[m]) {
        const char ch = md[m];
        if (ch >= '0' && ch <= '9') {
            long v = 0;
            while (m < md_len && md[m] >= '0' && md[m] <= '9') { v = v * 10 + (md[m] - '0'); m++; }
            pos += (int)v;
            continue;
        }
        if (ch == '^') { m++; while (m < md_len && md[m] && !(md[m] >= '0' && md[m] <= '9')) m++; continue; }
        for (i = pos - halo * 2 >= 0 ? pos - halo * 2 : 0; i < pos - halo && i < qlen; i++) nm[i] += 5;
        for (; i < pos + halo && i < qlen; i++) nm[i] += 10;
        for (; i < pos + halo * 2 && i < qlen; i++) nm[i] += 5;
        m++;
    }
}

// ---- frequency caller ----
struct SimpleAcc {
    uint64_t score[5];           // A C G T *
    int32_t tot_depth;
    CONS_HD void init() { for (int i = 0; i < 5; ++i) score[i] = 0; tot_depth = 0; }
    CONS_HD void add(const Par &o, uint32_t e)
    {
        const int q = CONS_E_QUAL(e);
        if (q < o.min_qual) return;
        const uint64_t w = o.use_qual ? (uint64_t)q : 1u;
        const int b = CONS_E_BASE4(e);
        if (b < 16) {
            // weight of nt16 code b on A, C, G, T: one hex digit per code, code 0 in the lowest digit (bam_consensus.c:1916-1919)
            const int sh = b * 4;
            score[0] += ((0x1020204020404080ull >> sh) & 15u) * w; score[1] += ((0x1200240024004800ull >> sh) & 15u) * w;
            score[2] += ((0x1224000014480000ull >> sh) & 15u) * w; score[3] += ((0x1228244800000000ull >> sh) & 15u) * w;
        } else score[4] += 8 * w;
        tot_depth++;
    }
    CONS_HD int finish(const Par &o, int32_t &qual) const
    {
        uint64_t tscore = 0, s1 = 0, s2 = 0;
        int call1 = 15, call2 = 15;
        for (int i = 0; i < 5; ++i) tscore += score[i];
        for (int i = 0; i < 5; ++i) {
            const int c = 1 << i;
            if (s1 < score[i]) { s2 = s1; call2 = call1; s1 = score[i]; call1 = c; }
            else if (s2 < score[i]) { s2 = score[i]; call2 = c; }
        }
        uint64_t used = s1;
        int ub = call1;
        if ((double)s2 >= o.het_fract * (double)s1 && o.ambig) { ub |= call2; used += s2; }
        if (tot_depth < o.min_depth || (double)used < o.call_fract * (double)tscore) ub = call1 == 16 ? 16 : 0;
        // 100.0 * 0 / 0 is NaN and the reference's (int) of it is INT_MIN on x86-64 (cvttsd2si): all reads inside a ref skip
        qual = ub ? (tscore ? (int32_t)(100.0 * (double)used / (double)tscore) : INT32_MIN) : 0;
        return "NACMGRSVTWYHKDBN*ac?g???t???????"[ub];
    }
};

// ---- Bayesian caller ----
struct Call { int32_t call, het_call, het_logodd, phred, depth; };

struct Gap5Acc {
    double S[15];
    int32_t n_N, depth;
    CONS_HD void init() { for (int j = 0; j < 15; ++j) S[j] = 0; n_N = 0; depth = 0; }
    // e: entry; nmw: nm word (0 when the read has none); mapq: mapping quality; td: number of reads alive in the column
    // q2p / mqpow: Tables::q2p and ::mqual_pow_1m (the device keeps them and cp in LDS); q0_absent: the read's first quality
    // byte is 0xff
    CONS_HD void add(const Par &o, const double *q2p, const double *mqpow, const Probs &cp, uint32_t e, uint32_t nmw, int mapq, bool q0_absent, int td)
    {
        const int pq = CONS_E_QUAL(e);
        if (pq < o.min_qual) return;
        if (e & CONS_E_REFSKIP) return;
        int qual = pq;
        if (qual == 255 || (qual == 0 && q0_absent)) qual = o.default_qual & 255;
        const int b4 = CONS_E_BASE4(e);
        // =ACM GRSV TWYH KDBN * -> A C G T * N
        const int base = b4 >= 16 ? 4 : (b4 == 1 ? 0 : b4 == 2 ? 1 : b4 == 4 ? 2 : b4 == 8 ? 3 : 5);
        if (o.use_mqual) {
            double mqual = mapq;
            if (o.nm_adjust) {
                const double nml = (nmw & CONS_NM_FLAT) ? (double)(nmw & 0xffffffu) : (double)(nmw & 0xffffffu) / 10.0;
                mqual /= (nml + 1);
                mqual *= 1 + 2 * (0.5 - (td > 30 ? 30 : td) / 60.0);
            }
            mqual *= o.scale_mqual;
            if (mqual < o.low_mqual) mqual = o.low_mqual;
            if (mqual > o.high_mqual) mqual = o.high_mqual;
            const double P = q2p[qual > 100 ? 100 : qual], M = mqpow[(int)mqual & 255];
            qual = (int)ph_log(P + .75 * M - P * M) & 255;
        }
        if (qual < 1) qual = 1;
        if (qual > 100) qual = 100;
        const double poly = (nmw & CONS_NM_FLAT) ? 0.0 : (double)((nmw >> 24) & 127u);
        const double q2d = qual - (poly - 2) * cp.poly_mul;
        int qual2 = (int)(1 > q2d ? 1 : q2d);
        if (qual2 > 100) qual2 = 100;
        const double xx = cp.pxx[qual];
        const double MM = cp.pMM[qual] - xx, xM = cp.pxM[qual] - xx;
        const double oo = cp.poo[qual2] - xx, oM = cp.poM[qual2] - xx, ox = cp.pox[qual2] - xx;
        const double uu = cp.puu[qual2] - xx, um = cp.pum[qual2] - xx, mm = cp.pmm[qual2] - xx;
        if (base == 5) n_N++;
        // genotype j = (a, c) over ACGT*: 0 AA 1 AC 2 AG 3 AT 4 A* 5 CC 6 CG 7 CT 8 C* 9 GG 10 GT 11 G* 12 TT 13 T* 14 **.
        // Every genotype gets one addend per read; where the reference adds nothing (a called base against a genotype that
        // does not hold it) the addend is +0.0, which leaves the sum bit-identical and keeps S[] in registers (no indexing).
        const bool called = base < 4, star = base == 4;
        const double other = called ? 0.0 : star ? uu : MM;              // genotypes of two bases, seen from a pad / an N
        const double hA = base == 0 ? MM : other, hC = base == 1 ? MM : other, hG = base == 2 ? MM : other, hT = base == 3 ? MM : other;
        const double xA = base == 0 ? xM : other, xC = base == 1 ? xM : other, xG = base == 2 ? xM : other, xT = base == 3 ? xM : other;
        const double sx = called ? ox : star ? um : oM;                  // (base, *) genotypes that do not hold the called base
        const double sA = base == 0 ? oM : sx, sC = base == 1 ? oM : sx, sG = base == 2 ? oM : sx, sT = base == 3 ? oM : sx;
        S[0] += hA;
        S[1] += base == 1 ? xC : xA;  S[2] += base == 2 ? xG : xA;  S[3] += base == 3 ? xT : xA;  S[4] += sA;
        S[5] += hC;
        S[6] += base == 2 ? xG : xC;  S[7] += base == 3 ? xT : xC;  S[8] += sC;
        S[9] += hG;
        S[10] += base == 3 ? xT : xG; S[11] += sG;
        S[12] += hT; S[13] += sT;
        S[14] += star ? mm : oo;
        depth++;
    }
    CONS_HD void finish(const Tables &t, const Probs &cp, Call &cons)
    {
        const double min_e_exp = -1021 * 0.693147180559945309417232121458 + 1;      // DBL_MIN_EXP * log(2) + 1
        const double DMAX = 1.7976931348623157e308, DMIN = 2.2250738585072014e-308;
        double shift = -DMAX, mx = -DMAX, mx_het = -DMAX;
        int call = 0, het_call = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 15; ++j) {
            S[j] += cp.lprior15[j];
            if (shift < S[j]) shift = S[j];
            const bool pure = j == 0 || j == 5 || j == 9 || j == 12 || j == 14;
            if (!pure) { if (mx_het < S[j]) { mx_het = S[j]; het_call = j; } continue; }
            if (mx < S[j]) { mx = S[j]; call = j; }
        }
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 15; ++j) {
            S[j] -= shift;
            const double e = fast_exp(t, S[j]);
            S[j] = S[j] > min_e_exp ? e : DMIN;
        }
        // norm[m] = (sum of S before m, added up from 0) + (sum of S after m, added up from 14 down): the reference builds both
        // running sums in one loop (bam_consensus.c:1739-1744); only the two called genotypes are needed
        double tot1 = 0, tot2 = 0, Sc = 0, Sh = 0, P_c = 0, P_h = 0, Q_c = 0, Q_h = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 15; ++j) {
            if (j == call) { P_c = tot1; Sc = S[j]; }
            if (j == het_call) { P_h = tot1; Sh = S[j]; }
            if (14 - j == call) Q_c = tot2;
            if (14 - j == het_call) Q_h = tot2;
            tot1 += S[j]; tot2 += S[14 - j];
        }
        double norm_c = P_c + Q_c, norm_h = P_h + Q_h;
        if (!depth || depth == n_N) { cons.call = 4; cons.het_call = 0; cons.het_logodd = 0; cons.phred = 0; cons.depth = 0; return; }
        cons.depth = depth;
        if (norm_c == 0) norm_c = DMIN;
        int ph;
        if (Sc == 1 && norm_c < .01) ph = (int)(ph_log(norm_c) + .5);
        else ph = (int)(ph_log(1 - Sc / (norm_c + Sc)) + .5);
        cons.call = call == 0 ? 0 : call == 5 ? 1 : call == 9 ? 2 : call == 12 ? 3 : 4;          // pure genotypes only
        cons.phred = ph < 0 ? 0 : ph;
        if (norm_h == 0) norm_h = DMIN;
        ph = (int)(3.0103 * (fast_log2(Sh) - fast_log2(norm_h)) + .5);
        // index in the 5x5 matrix of the pair: rows 0 5 9 12 14 of the triangle start at 0 6 12 18 24
        cons.het_call = het_call < 5 ? het_call : het_call < 9 ? het_call + 1 : het_call < 12 ? het_call + 3 : het_call < 14 ? het_call + 6 : 24;
        cons.het_logodd = ph;
    }
};

CONS_HD int imin(int a, int b) { return a < b ? a : b; }
CONS_HD int imax(int a, int b) { return a > b ? a : b; }

// MODE_MIXED: blend of the precision- and recall-oriented calls
CONS_HD Call mix_calls(const Call &P, const Call &R0)
{
    Call R = R0, c = P;
    if (P.phred > 0 && R.phred > 0 && P.call == R.call) c.phred += imin(20, R.phred);
    else if (P.het_logodd >= 0 && R.het_logodd >= 0 && P.het_call == R.het_call) c.het_logodd += imin(20, R.het_logodd);
    else if (P.het_logodd >= 0) { const int q2 = imax(R.phred, R.het_logodd); c.het_logodd = imax(1, c.het_logodd - q2 / 2); }
    else if (R.het_logodd >= 70) {
        const int q1 = P.phred, q2 = R.het_logodd;
        c = R;
        const double a = (q2 - q1 * 2) / 2, b = 1 + q2 / (q1 + 1.0), m = a > b ? a : b;
        c.het_logodd = (int)(15 < m ? 15 : m);
    } else if (R.het_logodd >= 0) {
        const int q1 = P.phred, q2 = R.het_logodd;
        c = R;
        const double v = q2 - 0.3 * q1;
        c.het_logodd = (int)((1 > v ? 1 : v) + 5 * (P.het_call == R.het_call));
        c.phred = 0;
    } else {
        R.phred = R.phred / 2;
        if (R.phred > P.phred) c = R;
        c.phred = imax(10, c.phred);
    }
    return c;
}

CONS_HD int final_call(const Par &o, const Call &cons, int32_t &qual)
{
    int cb, cq;
    if (cons.depth < o.min_depth && cons.call != 4) { cb = 'N'; cq = 0; }
    else if (cons.het_logodd > 0 && o.ambig) { cb = "AMRWaMCSYcRSGKgWYKTtacgt*"[cons.het_call]; cq = cons.het_logodd; }
    else { cb = "ACGT*"[cons.call]; cq = cons.phred; }
    if (cq < o.cons_cutoff && cb != '*' && cons.het_call % 5 != 4 && cons.het_call / 5 != 4) { cb = 'N'; cq = 0; }
    qual = cq;
    return cb;
}

}  // namespace cons
