// baq_band7s.h -- BAQ for the common read shape, written once for the device and for the CPU harness.
//
// What it replaces: HTSlib realn.c sam_prob_realn() + probaln.c probaln_glocal() (absent from the reference tree; call
// site bam_plcmd.c:451), for reads of class S:
//     CIGAR = [H] [S] one M/=/X operation [S] [H], band width 7, the reference window [xb, xe) not clipped by a contig end
//     (so l_ref = l_query + 6), 16 <= l_query <= 256, and the same l_query for every read a wave takes.
// For such a read the read's own diagonal is band cell JS = 10 in every row, the only rows with band cells outside the
// reference are rows 1..7 and row l_query, and the transition parameters are the same for all 64 lanes -- which is what this
// file exploits: the MAP step needs no arg-max bookkeeping (only "is the M state of cell 10 the first maximum of the row"),
// interior rows need no outside-the-window tests, emissions are picked with a per-row match-bit word, and the parameters live
// in scalar registers.  Every floating-point operation is the reference's, on the reference's operands, in the reference's
// order (build with -ffp-contract=off): results are bit-identical to the general kernels' and to
// the CPU restatement (tests/cpu/baq_emul.cpp, test infrastructure, runs these functions on the host beside it).
//
// One lane per read.  Per wave ("slot") scratch in HBM, [row][lane] so that a wave access is one contiguous run:
//     IN  uint32 [lq_cap + 2][64]   packed per-row inputs (see pack_lane)
//     F2  (M, I) pairs of doubles [pairs][15][64]: the RAW forward cells of the ODD rows only (row 1: normalised)
//     S   double [lq_cap + 2][64]   the forward row sums s[0 .. lq + 1]
// The forward pass stores odd rows only; the backward pass re-normalises them, re-runs their D chain and re-evaluates the even
// row above with the forward pass's own expressions (same scheme as k_baq_bwd<7, 2> in kernels_baq.hip).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define BQS_HD __host__ __device__ __forceinline__
#else
#define BQS_HD inline
#endif

#ifndef BQS_TEST_FORCE_EDGE
#define BQS_TEST_FORCE_EDGE false          // the CPU harness can send every row through the all-tests (EDGE) code
#endif

namespace baq7s {

constexpr int BW = 7, NB = 15, JS = 10;
constexpr double kEI = .25, kEM = .33333333333;
constexpr uint64_t AMB_MASK = 0444444444444444ull;     // bit 2 of every 3-bit field: code >= 4 (ambiguous 4, outside the window 7)
constexpr uint64_t ONE_MASK = 0111111111111111ull;     // bit 0 of every field
constexpr uint64_t WORD_MASK = (1ull << 45) - 1;

typedef double d2 __attribute__((ext_vector_type(2)));

struct Par { double m0, m1, m2, m3, m4, m6, m8, sM, sI, bM, bI, eim1, eim4; };

// probaln_glocal's transition parameters (probaln_par_t { d = 0.001f, e = 0.1f })
BQS_HD Par make_par(int lq, int l_ref)
{
    Par p;
    const float cd = 0.001f, ce = 0.1f;
    p.sM = p.sI = 1. / (2 * lq + 2);
    p.m0 = (1 - cd - cd) * (1 - p.sM); p.m1 = p.m2 = cd * (1 - p.sM);
    p.m3 = (1 - ce) * (1 - p.sI); p.m4 = ce * (1 - p.sI);
    p.m6 = 1 - ce; p.m8 = ce;
    p.bM = (1 - cd) / l_ref; p.bI = cd / l_ref;
    p.eim1 = kEI * p.m1; p.eim4 = kEI * p.m4;
    return p;
}

// ---- the things that differ between the device and the CPU harness ----
#if defined(__HIP_DEVICE_COMPILE__)
BQS_HD bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
// MODE: 0 = non-temporal row stream, 1 = plain loads / stores; diagnostics with wrong results: 2 = no row stream at all (the
// arithmetic and the small per-row inputs), 3 = 2 with the small inputs taken from a handful of cache-hot rows; 4, 5, 6: scheduling
// experiments with right results (4: the backward step's scaling fenced behind its chain; 5: 4 + the pair's phases fenced apart; 6: the
// forward pass fetches its row words four rows ahead; 7: plain stores + non-temporal loads; 8: non-temporal stores + plain loads)
template <int MODE> BQS_HD d2 ld_d2(const d2 *p) { if (MODE == 2 || MODE == 3) { d2 v = { 1e-3, 1e-3 }; return v; } return (MODE == 1 || MODE == 8) ? *p : __builtin_nontemporal_load(p); }
template <int MODE> BQS_HD void st_d2(d2 *p, d2 v) { if (MODE == 2 || MODE == 3) return; if (MODE == 1 || MODE == 7) *p = v; else __builtin_nontemporal_store(v, p); }
template <int MODE> BQS_HD int hot_row(int i) { return MODE == 3 ? 1 + (i & 7) : i; }
BQS_HD double fmax_(double a, double b) { return __builtin_fmax(a, b); }
BQS_HD void sched_fence() { __builtin_amdgcn_sched_barrier(0); }       // nothing is scheduled across this point
#else
BQS_HD bool wave_any(bool c) { return c; }
template <int MODE> BQS_HD d2 ld_d2(const d2 *p) { return *p; }
template <int MODE> BQS_HD void st_d2(d2 *p, d2 v) { *p = v; }
template <int MODE> BQS_HD int hot_row(int i) { return i; }
BQS_HD double fmax_(double a, double b) { return fmax(a, b); }
BQS_HD void sched_fence() {}
#endif

BQS_HD uint64_t d_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
BQS_HD double bits_d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

// element (row, lane) of a [row][lane] array of a slot: the slot pointers are wave-uniform, the index is 32 bits (one VGPR instead of a
// 64-bit address per array)
template <int LS> BQS_HD uint32_t at(int row, int ln) { return (uint32_t)row * (uint32_t)LS + (uint32_t)ln; }

#define BQS_FLD(w, j) ((int)((uint32_t)((w) >> (3 * (j))) & 7u))

// packed input word of row r (1-based; query index r - 1):
//   bits  0.. 7  base quality (as staged; the emissions always use this byte)
//   bits  8..10  query code 0..3, 4 = anything else
//   bits 11..13  reference code entering the band at its upper end in row r  = code(r + BW - 1)   (forward pass)
//   bits 14..16  reference code entering the band at its lower end in row r  = code(r - BW - 1)   (backward pass)
//   bits 24..31  the quality being worked on: the backward pass lowers it to the right-hand running maximum, the final
//                pass to the left-hand one
// code(idx): 0..3 = A C G T, 4 = ambiguous, 7 = idx outside [0, l_ref)
BQS_HD int rcode(const char *ref, int l_ref, int idx, const uint8_t *refc) { return (idx >= 0 && idx < l_ref) ? (int)refc[(unsigned char)ref[idx]] : 7; }
BQS_HD int qcode(int nib) { return nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : 4; }

// returns true when the window holds an ambiguous reference base (such a group takes the all-tests code in every row)
template <int LS>
BQS_HD bool pack_lane(int lq, int l_ref, const uint8_t *qual, const uint8_t *seq, const char *ref, const uint8_t *refc, uint32_t *IN, int ln)
{
    bool amb = false;
    for (int r = 1; r <= lq; ++r) {
        const int i0 = r - 1;
        const uint32_t q = qual[i0];
        const int nib = (seq[i0 >> 1] >> ((~i0 & 1) << 2)) & 0xf;
        const int fc = rcode(ref, l_ref, r + BW - 1, refc), bc = rcode(ref, l_ref, r - BW - 1, refc);
        amb |= fc == 4 || bc == 4;
        const uint32_t w = q | (uint32_t)qcode(nib) << 8 | (uint32_t)fc << 11 | (uint32_t)bc << 14 | q << 24;
        IN[at<LS>(r, ln)] = w;
    }
    return amb;
}

// emission of one band cell.  EDGE rows test everything; interior rows (all 15 cells inside the window, no ambiguous
// reference base) pick between the row's two values with the match bit of the cell.
struct Emis { double ematch, e_lo; uint64_t nm; int qyc; };
BQS_HD Emis make_emis(uint32_t w, uint64_t rw, const float *q2p)
{
    Emis e;
    const double qli = q2p[w & 255];
    const int qy = (int)((w >> 8) & 7);
    e.ematch = 1. - qli; e.e_lo = qy > 3 ? 1. : qli * kEM;
    e.qyc = qy > 3 ? 9 : qy;
    // bit 3j of nm: field j of the band word equals the query code (fields are <= 3 where this is used)
    const uint64_t x = rw ^ ((uint64_t)(qy & 3) * ONE_MASK);
    e.nm = ~(x | (x >> 1) | (x >> 2)) & ONE_MASK;
    if (qy > 3) e.ematch = 1.;          // an ambiguous query base: emission 1 whatever the reference says
    return e;
}
template <bool EDGE>
BQS_HD double emis_cell(const Emis &e, uint64_t rw, int j)
{
    if (EDGE) {
        const int rc = BQS_FLD(rw, j);
        const double v = (rc == e.qyc) ? e.ematch : e.e_lo;
        const double hi = (rc == 7) ? 0. : 1.;
        return rc > 3 ? hi : v;
    }
    // a bit-wise blend, not `bit ? ematch : e_lo`: the compiler turned that select into a two-entry table in scratch memory
    const uint64_t m = 0 - ((e.nm >> (3 * j)) & 1);
    return bits_d((d_bits(e.ematch) & m) | (d_bits(e.e_lo) & ~m));
}

// ------------------------------------------------------------------------------------------------------------------
// forward pass of one read
template <bool EDGE>
BQS_HD double fwd_row(const Par &p, const Emis &em, uint64_t rw, double (&M)[NB], double (&I)[NB], double (&D)[NB])
{
    double sum = 0., pm = 0., pd = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double e = emis_cell<EDGE>(em, rw, j);
        const double fm = e * (p.m0 * M[j] + p.m3 * I[j] + p.m6 * D[j]);
        const double fi = (j + 1 < NB) ? kEI * (p.m1 * M[j + 1] + p.m4 * I[j + 1]) : 0.;
        double fd = p.m2 * pm + p.m8 * pd;
        if (EDGE) fd = BQS_FLD(rw, j) == 7 ? 0. : fd;
        M[j] = fm; I[j] = fi; D[j] = fd;
        sum += fm + fi + fd;
        pm = fm; pd = fd;
    }
    return sum;
}

struct FwdState { double M[NB], I[NB], D[NB]; uint64_t rw; uint32_t w_next, w_next2, w_next3, w_next4; };

// one row i >= 2: inputs, the row, its sum, the raw store of an odd row, the normalisation
template <int LS, bool EDGE, int MODE>
BQS_HD void fwd_step(const Par &p, int lq, int i, const uint32_t *IN, d2 *F2, double *S, int ln, const float *q2p, FwdState &f)
{
    const uint32_t w = f.w_next;
    f.w_next = f.w_next2;
    if (MODE == 6) {                          // experiment: four rows ahead
        f.w_next2 = f.w_next3; f.w_next3 = f.w_next4;
        if (i + 4 <= lq) f.w_next4 = IN[at<LS>(i + 4, ln)];
    } else if (i + 2 <= lq) f.w_next2 = IN[at<LS>(hot_row<MODE>(i + 2), ln)];
    f.rw = (f.rw >> 3) | ((uint64_t)((w >> 11) & 7u) << (3 * (NB - 1)));
    const Emis em = make_emis(w, f.rw, q2p);
    const double sum = fwd_row<EDGE>(p, em, f.rw, f.M, f.I, f.D);
    if (i & 1) {                          // raw (M, I) of an odd row; even rows are not stored
        const int t = ((i - 1) >> 1) * NB;
#pragma unroll
        for (int j = 0; j < NB; ++j) { d2 v = { f.M[j], f.I[j] }; st_d2<MODE>(&F2[at<LS>(t + j, ln)], v); }
    }
    const double inv = 1. / sum;
    S[at<LS>(i, ln)] = i < lq ? inv : sum;      // rows below the top: 1 / s[i], the value the backward pass multiplies by (no division there)
#pragma unroll
    for (int j = 0; j < NB; ++j) { f.M[j] *= inv; f.I[j] *= inv; f.D[j] *= inv; }
}

// all_edge: the group's windows hold an ambiguous reference base somewhere: every row takes the all-tests code.  Otherwise rows
// 8 .. lq - 1 (all 15 cells inside the window) take the interior code.  Three loops, not a branch per row: a row body that exists
// in two variants inside one loop doubles the live state at the join (measured: +130 spilled registers).
template <int LS, int MODE = 0>
BQS_HD void fwd_lane(const Par &p, int lq, bool all_edge, const uint32_t *IN, d2 *F2, double *S, int ln, const float *q2p)
{
    FwdState f;
    // band word of row 1: field j = code(j - BW): outside the window below cell BW, code(0..7) above = the lower-end codes of rows 8..15
    f.rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) f.rw |= (uint64_t)(j < BW ? 7u : ((IN[at<LS>(j + 1, ln)] >> 14) & 7u)) << (3 * j);
    S[at<LS>(0, ln)] = 1.;
    {   // row 1 (no D state; the only row normalised by a division)
        const uint32_t w = IN[at<LS>(1, ln)];
        const Emis em = make_emis(w, f.rw, q2p);
        const double eibi = kEI * p.bI;
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rc = BQS_FLD(f.rw, j);
            const double e = emis_cell<true>(em, f.rw, j);
            const double a = e * p.bM;
            const double b2 = rc == 7 ? 0. : eibi;
            f.M[j] = a; f.I[j] = b2; f.D[j] = 0.;
            sum += a + b2;
        }
        S[at<LS>(1, ln)] = 1. / sum;             // (row 1 itself is normalised by divisions; the backward step to row 1 multiplies by 1 / s[1])
#pragma unroll
        for (int j = 0; j < NB; ++j) { f.M[j] /= sum; f.I[j] /= sum; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { d2 v = { f.M[j], f.I[j] }; st_d2<MODE>(&F2[at<LS>(j, ln)], v); }
    }
    f.w_next = IN[at<LS>(2, ln)]; f.w_next2 = lq >= 3 ? IN[at<LS>(3, ln)] : 0;
    f.w_next3 = (MODE == 6 && lq >= 4) ? IN[at<LS>(4, ln)] : 0; f.w_next4 = (MODE == 6 && lq >= 5) ? IN[at<LS>(5, ln)] : 0;
    const int e1 = (all_edge || BQS_TEST_FORCE_EDGE) ? lq : BW;
    int i = 2;
#pragma unroll 1
    for (; i <= e1; ++i) fwd_step<LS, true, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
#pragma unroll 1
    for (; i <= lq - 1; ++i) fwd_step<LS, false, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
#pragma unroll 1
    for (; i <= lq; ++i) fwd_step<LS, true, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
    {   // s[l_query + 1]
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += f.M[j] * p.sM + f.I[j] * p.sI;
        S[at<LS>(lq + 1, ln)] = sum;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward pass + MAP + the right-hand running maximum

// MAP of one row without an arg max: realn.c only asks whether the MAP state is the M state on the read's own diagonal,
// i.e. whether z of cell (M, JS) is the FIRST maximum of the row in the order M0, I0, M1, I1, ... (probaln_glocal: `if (z >
// max) max = z, max_k = ...` from max = 0).  Cells in front of it must stay strictly below it, cells behind it at or below;
// the quality then comes from that z over the row sum.  (A NaN z is never taken by the reference's `>` and is ignored here.)
struct MapAcc {
    double sum, maxE, zs; bool kill;
    BQS_HD void init() { sum = 0.; maxE = 0.; zs = 0.; kill = false; }
    template <int C> BQS_HD void add(double z)          // C = position in the order: 2 * j for M of cell j, 2 * j + 1 for I
    {
        if (C < 2 * JS) maxE = fmax_(maxE, z);
        else if (C == 2 * JS) { zs = z; kill = maxE >= z; }
        else kill |= z > zs;
        sum += z;
    }
};

// (int)v the way x86-64's cvttsd2si does it for out-of-range values and NaN (the CPU reference's behaviour), then probaln's cap
BQS_HD int map_quality(double zs, double sum)
{
    const double mx = zs / sum;
    const double v = -4.343 * log(1. - mx) + .499;
    int kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;
    return (int)(uint8_t)(kq > 100 ? 99 : kq);
}

struct BwdCtx {
    int ys, mlen;           // the M operation covers query indices [ys, ys + mlen)
    int run_r;              // running maximum of b from the right inside it
    bool plain;             // per-base BAQ (calmd -r without -E): no running maxima
};

// the result of row i: b (0 unless the MAP state is M on the read's diagonal), kept as a byte for the final pass, and the
// working quality lowered to the right-hand limit
template <int LS, class St>
BQS_HD void finish_row(BwdCtx &c, int i, const MapAcc &a, uint32_t w, uint32_t *IN, int ln, St state)
{
    const int q = i - 1;
    const int kq = map_quality(a.zs, a.sum);
    const bool in_m = q >= c.ys && q < c.ys + c.mlen;
    const int b = (in_m && !a.kill && a.zs > 0.) ? kq : 0;
    state[(size_t)q * LS] = (uint8_t)b;
    c.run_r = b > c.run_r ? b : c.run_r;                  // (b is 0 outside the M operation: no effect there)
    const int lim = c.plain ? b : c.run_r;
    const int q0 = (int)(w >> 24);
    const int q1 = (in_m && q0 > lim) ? lim : q0;
    IN[at<LS>(i, ln)] = (w & 0x00ffffffu) | ((uint32_t)q1 << 24);
}

// b[i] from b[i + 1] (in place), with the emissions of row i + 1 (band word rw1), then the division by s[i]
template <bool EDGE, int MODE = 0>
BQS_HD void bwd_apply(const Par &p, const Emis &em, uint64_t rw1, int i, double inv_i, double (&bM)[NB], double (&bI)[NB])
{
    double dnext = 0.;
    const double yv = i > 1 ? 1. : 0.;
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
        const double e = emis_cell<EDGE>(em, rw1, j) * bM[j];           // outside the window: 0 * b, as in the reference
        const double bi1 = j > 0 ? bI[j - 1] : 0.;
        const double bm = e * p.m0 + p.eim1 * bi1 + p.m2 * dnext;
        const double bi_ = e * p.m3 + p.eim4 * bi1;
        double bd = e * p.m6 + p.m8 * dnext;
        if (EDGE) bd *= yv;                                             // (interior rows: i > 1, the factor is 1)
        bM[j] = bm; bI[j] = bi_;
        dnext = bd;
    }
    if (MODE == 4 || MODE == 5) sched_fence();      // experiment: 1 / s[i] (a load of this pair) is first needed here, not a hundred instructions in
#pragma unroll
    for (int j = 0; j < NB; ++j) { bM[j] *= inv_i; bI[j] *= inv_i; }
    if (EDGE && i <= BW) {                  // cells with k < 1 do not exist in the reference: keep them at zero
#pragma unroll
        for (int j = 0; j < BW; ++j) if (j < BW + 1 - i) { bM[j] = 0.; bI[j] = 0.; }
    }
}

// MAP of a row whose normalised (M, I) are in registers
BQS_HD void map_row(MapAcc &a, const double (&fM)[NB], const double (&fI)[NB], const double (&bM)[NB], const double (&bI)[NB])
{
    a.init();
#define BQS_MAP_CELL(j) a.template add<2 * (j)>(fM[j] * bM[j]); a.template add<2 * (j) + 1>(fI[j] * bI[j]);
    BQS_MAP_CELL(0) BQS_MAP_CELL(1) BQS_MAP_CELL(2) BQS_MAP_CELL(3) BQS_MAP_CELL(4) BQS_MAP_CELL(5) BQS_MAP_CELL(6) BQS_MAP_CELL(7)
    BQS_MAP_CELL(8) BQS_MAP_CELL(9) BQS_MAP_CELL(10) BQS_MAP_CELL(11) BQS_MAP_CELL(12) BQS_MAP_CELL(13) BQS_MAP_CELL(14)
#undef BQS_MAP_CELL
}

// The even row i from the odd row i - 1 below it, fused with the even row's MAP terms.  (Mp, Ip): the RAW cells of row i - 1
// as stored (row 1: normalised); on return they are that row's normalised (M, I).  inv_o = 1 / s[i - 1], inv_i = 1 / s[i];
// em / rw: emissions and band word of row i.
template <bool EDGE, int J> struct EvenCell {
    static BQS_HD void run(const Par &p, const Emis &em, uint64_t rw, int i, int l_ref, double inv_o, double inv_i, double m2o, double m8o,
                           double (&Mp)[NB], double (&Ip)[NB], const double (&bM)[NB], const double (&bI)[NB], double &pm, double &pd, MapAcc &a)
    {
        double fd = m2o * pm + m8o * pd;
        if (EDGE) { const int idx = i - 1 - BW - 1 + J; fd = (idx < 0 || idx >= l_ref) ? 0. : fd; }
        pm = Mp[J]; pd = fd;
        const double Mn = Mp[J] * inv_o, In = Ip[J] * inv_o, Dn = fd * inv_o;
        Mp[J] = Mn; Ip[J] = In;
        if (J > 0) {
            const double fi = (kEI * (p.m1 * Mn + p.m4 * In)) * inv_i;             // the forward pass's I[i][J - 1]
            a.template add<2 * (J > 0 ? J - 1 : 0) + 1>(fi * bI[J > 0 ? J - 1 : 0]);
        }
        const double e = emis_cell<EDGE>(em, rw, J);
        const double fm = (e * (p.m0 * Mn + p.m3 * In + p.m6 * Dn)) * inv_i;        // the forward pass's M[i][J]
        a.template add<2 * J>(fm * bM[J]);
        EvenCell<EDGE, J + 1>::run(p, em, rw, i, l_ref, inv_o, inv_i, m2o, m8o, Mp, Ip, bM, bI, pm, pd, a);
    }
};
template <bool EDGE> struct EvenCell<EDGE, NB> {
    static BQS_HD void run(const Par &, const Emis &, uint64_t, int, int, double, double, double, double, double (&)[NB], double (&)[NB],
                           const double (&)[NB], const double (&bI)[NB], double &, double &, MapAcc &a)
    {
        a.template add<2 * (NB - 1) + 1>(0. * bI[NB - 1]);      // I[i][NB - 1] = 0: the last term of the row
    }
};

struct BwdState { double bM[NB], bI[NB]; uint64_t rw; uint32_t w_up; };

// one pair (i even, i - 1 odd).  b.rw is the band word of row min(i + 2, lq) when a pair starts, b.w_up the input word of row i + 1
// (of row lq for the first pair).
template <int LS, bool EDGE, int MODE, class St>
BQS_HD void bwd_pair(const Par &p, int lq, int l_ref, int i, uint32_t *IN, const d2 *F2, const double *S, int ln, const float *q2p, St state, BwdCtx &c, BwdState &b)
{
    const int t = ((i - 1) >> 1) * NB;
    // the small inputs in FRONT of the thirty cell loads: the first thing the pair needs is 1 / s[i], and loads come back in order
    const double s_i = S[at<LS>(hot_row<MODE>(i), ln)], s_o = S[at<LS>(hot_row<MODE>(i - 1), ln)];
    const uint32_t w_i = IN[at<LS>(hot_row<MODE>(i), ln)], w_o = IN[at<LS>(hot_row<MODE>(i - 1), ln)];
    double Mp[NB], Ip[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { const d2 v = ld_d2<MODE>(&F2[at<LS>(t + j, ln)]); Mp[j] = v.x; Ip[j] = v.y; }
    // band words: row i + 1 (emissions of the step to row i) and row i (re-evaluation of row i, step to row i - 1)
    uint64_t rw1 = b.rw;
    if (!EDGE || i < lq - 1) rw1 = ((b.rw << 3) | (uint64_t)((b.w_up >> 14) & 7u)) & WORD_MASK;
    const double inv_i = (EDGE && i >= lq) ? 1. / s_i : s_i;       // S[] holds 1 / s[row] below the top row, s[lq] itself for the top row
    const bool row1 = EDGE && i == 2;           // row 1 is stored normalised and has no D state
    const double inv_s = s_o;                   // the backward step to row i - 1 multiplies by 1 / s[i - 1] whatever the row
    const double inv_o = row1 ? 1. : inv_s;
    if (!EDGE || i < lq) { const Emis em1 = make_emis(b.w_up, rw1, q2p); bwd_apply<EDGE, MODE>(p, em1, rw1, i, inv_i, b.bM, b.bI); }
    // the emissions of row i only now: they hang on w_i, which was asked for at the top of this pair -- computed up there (where the
    // compiler would put them) the wave waits a whole memory round trip before its first fp64 instruction
    sched_fence();
    uint64_t rw0 = rw1;
    if (!EDGE || i < lq) rw0 = ((rw1 << 3) | (uint64_t)((w_i >> 14) & 7u)) & WORD_MASK;
    const Emis em0 = make_emis(w_i, rw0, q2p);
    MapAcc a;
    double pm = 0., pd = 0.;
    a.init();
    EvenCell<EDGE, 0>::run(p, em0, rw0, i, l_ref, inv_o, inv_i, row1 ? 0. : p.m2, row1 ? 0. : p.m8, Mp, Ip, b.bM, b.bI, pm, pd, a);
    if (MODE == 5) sched_fence();
    finish_row<LS>(c, i, a, w_i, IN, ln, state);
    if (MODE == 5) sched_fence();
    bwd_apply<EDGE, MODE>(p, em0, rw0, i - 1, inv_s, b.bM, b.bI);
    if (MODE == 5) sched_fence();
    map_row(a, Mp, Ip, b.bM, b.bI);
    finish_row<LS>(c, i - 1, a, w_o, IN, ln, state);
    b.rw = rw0; b.w_up = w_o;
}

// all_edge as in fwd_lane.  Otherwise the pairs with 10 <= i <= lq - 2 (rows i - 1 .. i + 1 have all cells inside the window)
// take the interior code: loops, not a branch per pair.
template <int LS, int MODE = 0, class St>
BQS_HD void bwd_lane(const Par &p, int lq, int l_ref, bool all_edge, uint32_t *IN, const d2 *F2, const double *S, int ln, const float *q2p, St state, BwdCtx &c)
{
    BwdState b;
    // band word of row lq: field j = code(lq - BW - 1 + j) = the upper-end code of row lq - 2 BW + j
    b.rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) b.rw |= (uint64_t)((IN[at<LS>(lq - 2 * BW + j, ln)] >> 11) & 7u) << (3 * j);
    const double s_top = S[at<LS>(lq, ln)];
    {
        const double sl1 = S[at<LS>(lq + 1, ln)];
        const double vM = p.sM / s_top / sl1, vI = p.sI / s_top / sl1;
#pragma unroll
        for (int j = 0; j < NB; ++j) { const bool valid = BQS_FLD(b.rw, j) != 7; b.bM[j] = valid ? vM : 0.; b.bI[j] = valid ? vI : 0.; }
    }
    c.run_r = 0;
    int i = lq;
    b.w_up = IN[at<LS>(lq, ln)];
    if (lq & 1) {
        // the top row is odd: stored raw, on its own
        const int t = ((lq - 1) >> 1) * NB;
        const double inv = 1. / s_top;
        double fM[NB], fI[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { const d2 v = ld_d2<MODE>(&F2[at<LS>(t + j, ln)]); fM[j] = v.x * inv; fI[j] = v.y * inv; }
        MapAcc a; map_row(a, fM, fI, b.bM, b.bI);
        finish_row<LS>(c, lq, a, b.w_up, IN, ln, state);
        --i;
    }
    const bool ae = all_edge || BQS_TEST_FORCE_EDGE;
    const int hi = ae ? 0 : lq - 2, lo = ae ? 2 : BW + 3;          // interior pairs: lo <= i <= hi
#pragma unroll 1
    for (; i >= 2 && i > hi; i -= 2) bwd_pair<LS, true, MODE>(p, lq, l_ref, i, IN, F2, S, ln, q2p, state, c, b);
#pragma unroll 1
    for (; i >= lo && !ae; i -= 2) bwd_pair<LS, false, MODE>(p, lq, l_ref, i, IN, F2, S, ln, q2p, state, c, b);
#pragma unroll 1
    for (; i >= 2; i -= 2) bwd_pair<LS, true, MODE>(p, lq, l_ref, i, IN, F2, S, ln, q2p, state, c, b);
}

// the left-hand running maximum (realn.c's extended BAQ: bq = min(left, right) inside the M operation) and the qualities' way home
template <int LS, class St>
BQS_HD void final_lane(int lq, const uint32_t *IN, int ln, St state, const BwdCtx &c, uint8_t *qual)
{
    int run = 0;
    for (int q = c.ys; q < c.ys + c.mlen; ++q) {
        const int b = state[(size_t)q * LS];
        run = b > run ? b : run;
        const int q1 = (int)(IN[at<LS>(q + 1, ln)] >> 24);
        qual[q] = (uint8_t)((!c.plain && q1 > run) ? run : q1);
    }
}

// is this read of class S?  (cigar: BAM encoding, op in the low 4 bits: M0 I1 D2 N3 S4 H5 P6 =7 X8)  On success the M operation's
// first query index, its length and the window start are returned.
struct Shape { bool ok; int ys, mlen; long long xb; };
BQS_HD Shape classify(const uint32_t *cigar, int n_cigar, long long rpos, int lq, long long ref_len)
{
    Shape s; s.ok = false; s.ys = 0; s.mlen = 0; s.xb = 0;
    int k = 0, y = 0;
    while (k < n_cigar && (cigar[k] & 0xf) == 5) ++k;
    if (k < n_cigar && (cigar[k] & 0xf) == 4) { y = (int)(cigar[k] >> 4); ++k; }
    if (k >= n_cigar) return s;
    const int op = cigar[k] & 0xf;
    if (!(op == 0 || op == 7 || op == 8)) return s;
    s.ys = y; s.mlen = (int)(cigar[k] >> 4); ++k;
    int tail = 0;
    if (k < n_cigar && (cigar[k] & 0xf) == 4) { tail = (int)(cigar[k] >> 4); ++k; }
    while (k < n_cigar && (cigar[k] & 0xf) == 5) ++k;
    if (k != n_cigar || s.mlen <= 0 || s.ys + s.mlen + tail != lq) return s;
    if (lq < 16 || lq > 256) return s;
    s.xb = rpos - s.ys - BW / 2;                          // realn.c: xb -= yb + bw / 2
    const long long xe = rpos + s.mlen + tail + BW / 2;   //          xe += l_qseq - ye + bw / 2
    if (s.xb < 0 || xe > ref_len) return s;               // a clipped window changes l_ref and the diagonal: general kernels
    s.ok = true;
    return s;
}

}   // namespace baq7s
