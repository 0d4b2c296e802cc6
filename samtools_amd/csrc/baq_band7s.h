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
//     F2  (M, I) pairs of doubles [stored rows][15][64]: the RAW forward cells of every THIRD row only (rows 1, 4, 7, ...; row 1: normalised)
//     S   double [lq_cap + 2][64]   the forward row sums s[0 .. lq + 1]
// The forward pass stores one row of three; the backward pass re-normalises it, re-runs its D chain and re-evaluates the two rows above
// it with the forward pass's own expressions (the scheme of k_baq_bwd<7, 2> in kernels_baq.hip -- one row of two there -- taken one row
// further: 80 + 8 + 8 instead of 120 + 8 + 8 bytes per query base and pass, for 8 % more vector instructions).  The middle row's
// normalised (M, I) wait for their own MAP step in LDS, [cell][lane] (15 KB per wave); the rows' results ride in their input words.
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
// MODE is a bit field.  Low four bits, the row stream: 0 = non-temporal (default), 1 = plain loads / stores; diagnostics with wrong
// results: 2 = no row stream at all (the arithmetic and the small per-row inputs), 3 = 2 with the small inputs taken from a handful of
// cache-hot rows.  (The scheduling experiments that were modes 4 - 8 in round 4 -- fences, deeper row-word prefetch, mixed load / store
// policies -- measured nothing and are gone: profiles/r04_baq_class_s.md.)  Feature bits, each its own instantiation so that builds can be
// compared inside one box:
//   M_LOGTAB  the MAP quality from the threshold table instead of an fp64 log (map_quality below): the default since round 5
// (Round 5 also built and measured M_DMA = 32 -- the backward pass's stored row by global -> LDS DMA one group ahead, into the LDS image the
// middle row is parked in afterwards -- and M_L2PF = 64 -- cache-warming loads of the next group's row: +0.4 and +1.0 ms, removed again;
// commit 43a95ec holds the code, profiles/r05_baq7s_counters.md the numbers and why: the kernel is issue-bound, not wait-bound.)
constexpr int M_MEM = 15, M_LOGTAB = 16;
template <int MODE> BQS_HD d2 ld_d2(const d2 *p) { if ((MODE & M_MEM) >= 2) { d2 v = { 1e-3, 1e-3 }; return v; } return (MODE & M_MEM) == 1 ? *p : __builtin_nontemporal_load(p); }
template <int MODE> BQS_HD void st_d2(d2 *p, d2 v) { if ((MODE & M_MEM) >= 2) return; if ((MODE & M_MEM) == 1) *p = v; else __builtin_nontemporal_store(v, p); }
template <int MODE> BQS_HD int hot_row(int i) { return (MODE & M_MEM) == 3 ? 1 + (i & 7) : i; }
BQS_HD double fmax_(double a, double b) { return __builtin_fmax(a, b); }
BQS_HD double fmin_(double a, double b) { return __builtin_fmin(a, b); }
BQS_HD void sched_fence() { __builtin_amdgcn_sched_barrier(0); }       // nothing is scheduled across this point
// bit `pos` of w set ? a : b, without a condition register: mask = the bit sign-extended (v_bfe_i32), then v_bfi_b32 on either half.
// The empty asm hides where the mask comes from: the compiler would turn the blend back into v_cmp + v_cndmask.
// `dep` (any value that becomes available just before the blend is needed) ties the mask to that point of the schedule: left free, the fifteen
// masks of a row are all computed up front and cost fifteen registers (measured: 240 spilled registers).
BQS_HD double blend_bit(uint64_t w, int pos, double a, double b, double dep)
{
    const uint32_t word = pos < 32 ? (uint32_t)w : (uint32_t)(w >> 32);
    uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)word, (unsigned)(pos & 31), 1u);
    asm("" : "+v"(m) : "v"((uint32_t)(uint64_t)__double_as_longlong(dep)));
    const uint64_t ua = (uint64_t)__double_as_longlong(a), ub = (uint64_t)__double_as_longlong(b);
    const uint32_t lo = ((uint32_t)ua & m) | ((uint32_t)ub & ~m), hi = ((uint32_t)(ua >> 32) & m) | ((uint32_t)(ub >> 32) & ~m);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
#else
BQS_HD bool wave_any(bool c) { return c; }
constexpr int M_MEM = 15, M_LOGTAB = 16;
template <int MODE> BQS_HD d2 ld_d2(const d2 *p) { return *p; }
template <int MODE> BQS_HD void st_d2(d2 *p, d2 v) { *p = v; }
template <int MODE> BQS_HD int hot_row(int i) { return i; }
BQS_HD double fmax_(double a, double b) { return fmax(a, b); }
BQS_HD double fmin_(double a, double b) { return fmin(a, b); }
BQS_HD void sched_fence() {}
BQS_HD double blend_bit(uint64_t w, int pos, double a, double b, double) { return ((w >> pos) & 1) ? a : b; }
#endif

BQS_HD uint64_t d_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
BQS_HD double bits_d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

// element (row, lane) of a [row][lane] array of a slot: wave-uniform slot pointer + a 32-bit BYTE offset (a slot is < 1 MiB), so that an access
// is `global_load/store v_off, s[base:base+1]` with one VGPR -- written as an element index the compiler forms a 64-bit address per access
// (v_add + v_mov + v_lshl_add_u64, and for a row of cells fifteen loop-invariant index registers: seen in the round-4 assembly).
template <int LS, class T> BQS_HD T *at(T *base, int row, int ln) { return (T *)((char *)base + (uint32_t)(((uint32_t)row * (uint32_t)LS + (uint32_t)ln) * (uint32_t)sizeof(T))); }

#define BQS_FLD(w, j) ((int)((uint32_t)((w) >> (3 * (j))) & 7u))

// packed input word of row r (1-based; query index r - 1):
//   bits  0.. 7  base quality (as staged; the emissions always use this byte)
//   bits  8..10  query code 0..3, 4 = anything else
//   bits 11..13  reference code entering the band at its upper end in row r  = code(r + BW - 1)   (forward pass)
//   bits 14..16  reference code entering the band at its lower end in row r  = code(r - BW - 1)   (backward pass)
//   bits 17..23  b of the row (backward pass): the MAP quality if the MAP state is M on the read's diagonal, else 0
//   bits 24..31  the quality being worked on: the backward pass lowers it to the right-hand running maximum, the final
//                pass to the left-hand one
// code(idx): 0..3 = A C G T, 4 = ambiguous, 7 = idx outside [0, l_ref)
BQS_HD int rcode(const char *ref, int l_ref, int idx, const uint8_t *refc) { return (idx >= 0 && idx < l_ref) ? (int)refc[(unsigned char)ref[idx]] : 7; }
BQS_HD int qcode(int nib) { return nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : 4; }

// returns true when the window holds an ambiguous reference base (such a group takes the all-tests code in every row).
// Eight rows per step: one 8-byte load of qualities, one 4-byte load of bases (both pools are padded to 8 bases per read and the read
// starts on an 8-byte boundary), the sixteen reference characters asked for together -- one row per step was a chain of dependent
// byte loads, 150 load latencies per read with nothing else to issue (round 5: the kernel is issue-bound, its latency-bound corners are
// where the SIMDs idle).
template <int LS>
BQS_HD bool pack_lane(int lq, int l_ref, const uint8_t *qual, const uint8_t *seq, const char *ref, const uint8_t *refc, uint32_t *IN, int ln)
{
    bool amb = false;
    for (int r0 = 1; r0 <= lq; r0 += 8) {
        const int i0 = r0 - 1;                                    // a multiple of 8
        uint64_t q8; uint32_t s4;
        __builtin_memcpy(&q8, (const uint8_t *)__builtin_assume_aligned(qual, 8) + i0, 8);
        __builtin_memcpy(&s4, (const uint8_t *)__builtin_assume_aligned(seq, 4) + (i0 >> 1), 4);
        unsigned char fch[8], bch[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int fi = r0 + k + BW - 1, bi = r0 + k - BW - 1;
            fch[k] = (fi >= 0 && fi < l_ref) ? (unsigned char)ref[fi] : 0; bch[k] = (bi >= 0 && bi < l_ref) ? (unsigned char)ref[bi] : 0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + k;
            if (r > lq) break;
            const int fi = r + BW - 1, bi = r - BW - 1;
            const uint32_t q = (uint32_t)(q8 >> (8 * k)) & 255u;
            const int nib = (int)(s4 >> (8 * (k >> 1) + ((~k & 1) << 2))) & 0xf;      // byte k / 2 of the word, high nibble first
            const int fc = (fi >= 0 && fi < l_ref) ? (int)refc[fch[k]] : 7, bc = (bi >= 0 && bi < l_ref) ? (int)refc[bch[k]] : 7;
            amb |= fc == 4 || bc == 4;
            const uint32_t w = q | (uint32_t)qcode(nib) << 8 | (uint32_t)fc << 11 | (uint32_t)bc << 14 | q << 24;
            *at<LS>(IN, r, ln) = w;
        }
    }
    return amb;
}

// emission of one band cell.  EDGE rows test everything; interior rows (all 15 cells inside the window, no ambiguous
// reference base) pick between the row's two values with the match bit of the cell.
// Interior rows hold only base codes 0..3 in their band word, so their words are kept two bits per field: fifteen fields in ONE 32-bit
// register (rw_pack2; the three-bit form needs 45 bits, and every shift, xor and mask on it is a 64-bit operation, the multiply that
// spreads the query code over the fields a full-rate-quarter v_mul_lo_u32).  The loops convert where they change between the two codes.
#ifndef BQS_TWOBIT
#define BQS_TWOBIT 1
#endif
constexpr bool TWOBIT = BQS_TWOBIT != 0;
BQS_HD uint64_t rw_pack2(uint64_t rw3) { if (!TWOBIT) return rw3; uint32_t r = 0; for (int j = 0; j < NB; ++j) r |= ((uint32_t)(rw3 >> (3 * j)) & 3u) << (2 * j); return r; }
BQS_HD uint64_t rw_unpack3(uint64_t rw2) { if (!TWOBIT) return rw2; uint64_t r = 0; for (int j = 0; j < NB; ++j) r |= (uint64_t)(((uint32_t)rw2 >> (2 * j)) & 3u) << (3 * j); return r; }
constexpr uint32_t ONE2 = 0x15555555u, WORD2_MASK = 0x3fffffffu;

struct Emis { double ematch, e_lo; uint64_t nm; int qyc; };
template <bool EDGE>
BQS_HD Emis make_emis(uint32_t w, uint64_t rw, const float *q2p)
{
    Emis e;
    const double qli = q2p[w & 255];
    const int qy = (int)((w >> 8) & 7);
    // (query code 4 = anything but A C G T: emission 1 whatever the reference says; bit 10 of the word is that code's bit 2)
    e.ematch = blend_bit(w, 10, 1., 1. - qli, qli); e.e_lo = blend_bit(w, 10, 1., qli * kEM, qli);
    e.qyc = qy + 5 * (qy >> 2);               // 9 for code 4: matches no reference code
    if (EDGE) { e.nm = 0; return e; }         // (the all-tests code compares the fields themselves)
    if (!TWOBIT) {                            // round 4's form: three-bit fields, bit 3j of nm
        const uint64_t x3 = rw ^ ((uint64_t)(qy & 3) * ONE_MASK);
        e.nm = ~(x3 | (x3 >> 1) | (x3 >> 2)) & ONE_MASK;
        return e;
    }
    // bit 2j of nm: field j of the two-bit band word equals the query code.  The code is spread over the fields by two masks (its bits 8 and
    // 9 of the word sign-extended), not by a multiplication.
    const uint32_t b0 = (uint32_t)((int32_t)(w << 23) >> 31), b1 = (uint32_t)((int32_t)(w << 22) >> 31);
    const uint32_t x = (uint32_t)rw ^ ((b0 & ONE2) | (b1 & (ONE2 << 1)));
    e.nm = (uint64_t)(~(x | (x >> 1)) & ONE2);
    return e;
}
template <bool EDGE>
BQS_HD double emis_cell(const Emis &e, uint64_t rw, int j, double dep)
{
    if (EDGE) {
        const int rc = BQS_FLD(rw, j);
        const double v = (rc == e.qyc) ? e.ematch : e.e_lo;
        const double hi = (rc == 7) ? 0. : 1.;
        return rc > 3 ? hi : v;
    }
    // A bit-wise blend by a register mask, not a select on a condition.  Measured on this chip (scripts/ubench/valu_cost.hip,
    // profiles/r04_valu_cost.md): v_cmp + two v_cndmask_b32 on VCC cost ~20 clocks of a saturated SIMD and ~50 of a single wave, v_bfe_i32 +
    // two v_bfi_b32 4.2 each.  (And written as `bit ? ematch : e_lo` the select became a two-entry table in scratch memory.)
    return blend_bit(e.nm, (TWOBIT ? 2 : 3) * j, e.ematch, e.e_lo, dep);
}
// band word of the next row up (forward pass), given that row's input word: its upper-end code comes in at cell NB - 1
template <bool EDGE> BQS_HD uint64_t word_up(uint64_t rw, uint32_t w)
{
    if (EDGE || !TWOBIT) return (rw >> 3) | ((uint64_t)((w >> 11) & 7u) << (3 * (NB - 1)));
    return (uint64_t)(((uint32_t)rw >> 2) | (((w >> 11) & 3u) << (2 * (NB - 1))));
}

// ------------------------------------------------------------------------------------------------------------------
// forward pass of one read
template <bool EDGE>
BQS_HD double fwd_row(const Par &p, const Emis &em, uint64_t rw, double (&M)[NB], double (&I)[NB], double (&D)[NB])
{
    double sum = 0., pm = 0., pd = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double t3 = p.m0 * M[j] + p.m3 * I[j] + p.m6 * D[j];
        const double e = emis_cell<EDGE>(em, rw, j, t3);
        const double fm = e * t3;
        const double fi = (j + 1 < NB) ? kEI * (p.m1 * M[j + 1] + p.m4 * I[j + 1]) : 0.;
        double fd = p.m2 * pm + p.m8 * pd;
        if (EDGE) fd = BQS_FLD(rw, j) == 7 ? 0. : fd;
        M[j] = fm; I[j] = fi; D[j] = fd;
        sum += fm + fi + fd;
        pm = fm; pd = fd;
    }
    return sum;
}

struct FwdState { double M[NB], I[NB], D[NB]; uint64_t rw; uint32_t w_next, w_next2; };

// one row i >= 2: inputs, the row, its sum, the raw store of an odd row, the normalisation
template <int LS, bool EDGE, int MODE>
BQS_HD void fwd_step(const Par &p, int lq, int i, const uint32_t *IN, d2 *F2, double *S, int ln, const float *q2p, FwdState &f)
{
    const uint32_t w = f.w_next;
    f.w_next = f.w_next2;
    if (i + 2 <= lq) f.w_next2 = *at<LS>(IN, hot_row<MODE>(i + 2), ln);
    f.rw = word_up<EDGE>(f.rw, w);
    const Emis em = make_emis<EDGE>(w, f.rw, q2p);
    const double sum = fwd_row<EDGE>(p, em, f.rw, f.M, f.I, f.D);
    if ((i - 1) % 3 == 0) {               // raw (M, I) of one row of three (rows 4, 7, ...); the others are not stored
        d2 *row = at<LS>(F2, ((i - 1) / 3) * NB, ln);
#pragma unroll
        for (int j = 0; j < NB; ++j) { d2 v = { f.M[j], f.I[j] }; st_d2<MODE>(row + j * LS, v); }
    }
    const double inv = 1. / sum;
    *at<LS>(S, i, ln) = i < lq ? inv : sum;      // rows below the top: 1 / s[i], the value the backward pass multiplies by (no division there)
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
    for (int j = 0; j < NB; ++j) f.rw |= (uint64_t)(j < BW ? 7u : ((*at<LS>(IN, j + 1, ln) >> 14) & 7u)) << (3 * j);
    *at<LS>(S, 0, ln) = 1.;
    {   // row 1 (no D state; the only row normalised by a division)
        const uint32_t w = *at<LS>(IN, 1, ln);
        const Emis em = make_emis<true>(w, f.rw, q2p);
        const double eibi = kEI * p.bI;
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rc = BQS_FLD(f.rw, j);
            const double e = emis_cell<true>(em, f.rw, j, 0.);
            const double a = e * p.bM;
            const double b2 = rc == 7 ? 0. : eibi;
            f.M[j] = a; f.I[j] = b2; f.D[j] = 0.;
            sum += a + b2;
        }
        *at<LS>(S, 1, ln) = 1. / sum;             // (row 1 itself is normalised by divisions; the backward step to row 1 multiplies by 1 / s[1])
#pragma unroll
        for (int j = 0; j < NB; ++j) { f.M[j] /= sum; f.I[j] /= sum; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { d2 v = { f.M[j], f.I[j] }; st_d2<MODE>(at<LS>(F2, 0, ln) + j * LS, v); }
    }
    f.w_next = *at<LS>(IN, 2, ln); f.w_next2 = lq >= 3 ? *at<LS>(IN, 3, ln) : 0;
    const int e1 = (all_edge || BQS_TEST_FORCE_EDGE) ? lq : BW;
    int i = 2;
#pragma unroll 1
    for (; i <= e1; ++i) fwd_step<LS, true, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
    if (i <= lq - 1) {
        f.rw = rw_pack2(f.rw);            // (row 7's word: its one field outside the window leaves with the first step)
#pragma unroll 1
        for (; i <= lq - 1; ++i) fwd_step<LS, false, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
        f.rw = rw_unpack3(f.rw);
    }
#pragma unroll 1
    for (; i <= lq; ++i) fwd_step<LS, true, MODE>(p, lq, i, IN, F2, S, ln, q2p, f);
    {   // s[l_query + 1]
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += f.M[j] * p.sM + f.I[j] * p.sI;
        *at<LS>(S, lq + 1, ln) = sum;
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

// probaln_glocal: k = (int)(-4.343 * log(1. - max) + .499); q = k > 100 ? 99 : k, stored in a byte.  (int)v is x86-64's cvttsd2si in the CPU
// reference: INT_MIN for v >= 2^31, +inf and NaN (a posterior of exactly 1 gives log(0) = -inf, v = +inf, k = INT_MIN, q = (uint8_t)k = 0).
// v >= .499 otherwise.  In integer arithmetic, without a condition register (v_cmp + v_cndmask cost 20-50 clocks a piece on this chip):
// the value is clamped to [0, 2^31] (a NaN goes to 2^31 through fmin), converted as unsigned, 2^31 is the one "bad" pattern.
// This is the formula as the reference writes it: the CPU harness checks the table form below against it, the general kernels still use
// its equivalent.
BQS_HD int map_quality_formula(double zs, double sum)
{
    const double mx = zs / sum;
    const double v = -4.343 * log(1. - mx) + .499;
    const double vc = fmax_(fmin_(v, 2147483648.0), 0.);
    const uint32_t ku = (uint32_t)vc;                            // <= 2^31: in range
    const int32_t bad = (int32_t)ku >> 31;                       // -1 for the INT_MIN cases
    const int32_t kk = (int32_t)(ku & ~(uint32_t)bad);           // those give the byte 0
    const int32_t k100 = kk < 100 ? kk : 100;
    return k100 - (int32_t)((uint32_t)(100 - kk) >> 31);         // k for k <= 100, 99 above
}

// The same value without the logarithm.  With x = 1. - max (the reference's own subtraction), F(x) = (int)(-4.343 * log(x) + .499) is a
// non-increasing step function of x on (0, 1] with steps at 0 .. 159; only "k >= 101" matters above 100.  LT[k], k = 1 .. 101, is the
// LARGEST double x with F(x) >= k, found on the host by bisection over the bit patterns with the host's own log() -- the routine the CPU
// reference calls -- so F(x) >= k  <=>  x <= LT[k]; LT[0] = +inf, LT[102] = -1 (never).  The table is exact, not an approximation: the
// neighbourhood of every step is walked double by double when the table is built (make_log_thresholds returns false unless F is a clean
// single step there; tests/test_baq_emul.py re-checks it over 2 x 10^5 doubles either side and against the formula on random posteriors),
// and away from a step a one-ulp wobble of log() cannot move F.  The device needs ~12 instructions instead of the ~45 of an fp64 log:
// k_a from a single-precision log2 (v_log_f32; off by at most one step), then k = k_a - 1 + [x <= LT[k_a]] + [x <= LT[k_a + 1]].
// x == 0 (a posterior that rounds to 1) is the reference's INT_MIN case: 0.  A NaN x fails both tests and is 0 as well (zs = 0 there).
constexpr int LT_N = 104;                                        // LT[0 .. 102] used; padded to an even count of 16-byte pairs
struct LogTab { double t[LT_N]; };
inline int log_step_host(double x) { const double v = -4.343 * log(x) + .499; return v >= 2147483648.0 ? 2147483647 : (int)v; }   // (host; x > 0)
inline bool make_log_thresholds(LogTab &T, int walk = 4096)
{
    T.t[0] = __builtin_inf(); for (int k = 102; k < LT_N; ++k) T.t[k] = -1.;
    bool clean = true;
    for (int k = 1; k <= 101; ++k) {
        // bit patterns of positive doubles order like the doubles: lo has F >= k (tiny x), hi has F < k (x = 1: F = 0)
        uint64_t lo, hi; { double a = 0x1p-60, b = 1.; __builtin_memcpy(&lo, &a, 8); __builtin_memcpy(&hi, &b, 8); }
        while (hi - lo > 1) { const uint64_t mid = lo + (hi - lo) / 2; double x; __builtin_memcpy(&x, &mid, 8); if (log_step_host(x) >= k) lo = mid; else hi = mid; }
        __builtin_memcpy(&T.t[k], &lo, 8);
        for (int d = 1; d <= walk; ++d) {                        // a clean single step: >= k at and below the threshold, < k above it
            double a, b; const uint64_t ua = lo - (uint64_t)(d - 1), ub = lo + (uint64_t)d; __builtin_memcpy(&a, &ua, 8); __builtin_memcpy(&b, &ub, 8);
            if (log_step_host(a) < k || log_step_host(b) >= k) clean = false;
        }
    }
    return clean;
}
#if defined(__HIP_DEVICE_COMPILE__)
BQS_HD float log2_fast(float x) { return __builtin_amdgcn_logf(x); }            // v_log_f32 (x is 0 or a normal number here)
#else
BQS_HD float log2_fast(float x) { return log2f(x); }
#endif
template <class Tab> BQS_HD int map_quality_x(double x, Tab LT);
template <class Tab>
BQS_HD int map_quality(double zs, double sum, Tab LT)
{
    const double mx = zs / sum;
    return map_quality_x(1. - mx, LT);
}
// x = 1. - max (the reference's own subtraction, done by the caller)
template <class Tab>
BQS_HD int map_quality_x(double x, Tab LT)
{
    const float va = log2_fast((float)x) * (float)(-4.343 * 0.693147180559945309) + .499f;      // ~ -4.343 ln x + .499, within 1e-4
    int ka = (int)fmaxf(fminf(va, 200.f), 0.f);                  // (x = 0: +inf -> 200; NaN -> 0)
    ka = ka < 1 ? 1 : (ka > 100 ? 100 : ka);
    const double t0 = LT[ka], t1 = LT[ka + 1];
    const int k = ka - 1 + (x <= t0 ? 1 : 0) + (x <= t1 ? 1 : 0);  // 0 .. 101
    const int k99 = k - 2 * (int)((uint32_t)(100 - k) >> 31);      // 101 -> 99
    return x > 0. ? k99 : 0;
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(3))) double *LtPtr;    // the threshold table lives in LDS on the device
#else
typedef const double *LtPtr;
#endif
struct BwdCtx {
    int ys, mlen;           // the M operation covers query indices [ys, ys + mlen)
    int run_r;              // running maximum of b from the right inside it
    int plain_mask;         // -1: per-base BAQ (calmd -r without -E): no running maxima; 0: extended BAQ
    LtPtr LT;               // LogTab::t (M_LOGTAB)
};

// the result of row i: b (0 unless the MAP state is M on the read's diagonal), kept in the row's word for the final pass, and the
// working quality lowered to the right-hand limit
template <int LS, int MODE>
BQS_HD void finish_row(BwdCtx &c, int i, const MapAcc &a, uint32_t w, uint32_t *IN, int ln)
{
    const int q = i - 1;
    const int kq = (MODE & M_LOGTAB) ? map_quality(a.zs, a.sum, c.LT) : map_quality_formula(a.zs, a.sum);
    // masks instead of conditions: inside the M operation (0 <= q - ys < mlen); the MAP state is M on the diagonal
    const int32_t t = q - c.ys;
    const int32_t m_in = ((t - c.mlen) & ~t) >> 31;
    const int32_t m_ok = (!a.kill && a.zs > 0.) ? -1 : 0;
    const int b = kq & m_in & m_ok;                             // 0 .. 100
    c.run_r = b > c.run_r ? b : c.run_r;                        // (b is 0 outside the M operation: no effect there)
    const int lim = c.run_r ^ ((c.run_r ^ b) & c.plain_mask);   // per-base BAQ: the row's own b
    const int q0 = (int)(w >> 24);
    const int qmin = q0 < lim ? q0 : lim;
    const int q1 = q0 ^ ((q0 ^ qmin) & m_in);
    *at<LS>(IN, i, ln) = (w & 0x0001ffffu) | ((uint32_t)b << 17) | ((uint32_t)q1 << 24);
}

// b[i] from b[i + 1] (in place), with the emissions of row i + 1 (band word rw1), then the division by s[i]
template <bool EDGE, int MODE = 0>
BQS_HD void bwd_apply(const Par &p, const Emis &em, uint64_t rw1, int i, double inv_i, double (&bM)[NB], double (&bI)[NB])
{
    double dnext = 0.;
    const double yv = i > 1 ? 1. : 0.;
#pragma unroll
    for (int j = NB - 1; j >= 0; --j) {
        const double e = emis_cell<EDGE>(em, rw1, j, dnext) * bM[j];    // outside the window: 0 * b, as in the reference
        const double bi1 = j > 0 ? bI[j - 1] : 0.;
        const double bm = e * p.m0 + p.eim1 * bi1 + p.m2 * dnext;
        const double bi_ = e * p.m3 + p.eim4 * bi1;
        double bd = e * p.m6 + p.m8 * dnext;
        if (EDGE) bd *= yv;                                             // (interior rows: i > 1, the factor is 1)
        bM[j] = bm; bI[j] = bi_;
        dnext = bd;
    }
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

// MAP of a row whose normalised (M, I) come back from LDS (the middle row of a group)
template <int LS, class Ld>
BQS_HD void map_row_lds(MapAcc &a, Ld Ln, const double (&bM)[NB], const double (&bI)[NB])
{
    a.init();
#define BQS_MAP_CELL(j) { const d2 v = Ln[(j) * LS]; a.template add<2 * (j)>(v.x * bM[j]); a.template add<2 * (j) + 1>(v.y * bI[j]); }
    BQS_MAP_CELL(0) BQS_MAP_CELL(1) BQS_MAP_CELL(2) BQS_MAP_CELL(3) BQS_MAP_CELL(4) BQS_MAP_CELL(5) BQS_MAP_CELL(6) BQS_MAP_CELL(7)
    BQS_MAP_CELL(8) BQS_MAP_CELL(9) BQS_MAP_CELL(10) BQS_MAP_CELL(11) BQS_MAP_CELL(12) BQS_MAP_CELL(13) BQS_MAP_CELL(14)
#undef BQS_MAP_CELL
}

// A group of rows (a, a + 1 [, a + 2]); row a is the stored one.  One ascending sweep over the cells, skewed by one cell:
//   step J   row a     : its D chain re-run from the raw cells, cell J normalised (kept in Mp / Ip for the row's own MAP step);
//            row a + 1 : M and D of cell J, I of cell J - 1, with the forward pass's expressions on the normalised row a (raw, then x 1 / s[a + 1]);
//                        cell J - 1 is complete with that I: with ROWS == 3 it goes to LDS (Ln) for the row's own MAP step, with ROWS == 2 its MAP
//                        terms are taken at once;
//            row a + 2 : (ROWS == 3) M of cell J - 1 and I of cell J - 2 from the complete cell J - 1 of row a + 1, and their MAP terms
//                        at once, in the order M0, I0, M1, I1, ... of probaln_glocal -- the top row of the group is never held.
// e1 / rw_1: emissions and band word of row a + 1; e2 / rw_2: of row a + 2; inv_a: 1 / s[a] (1 for row 1, which is stored normalised and has
// no D state: m2o = m8o = 0); inv1 = 1 / s[a + 1], inv2 = 1 / s[a + 2]; bM / bI: b of the group's top row.
template <bool EDGE, int ROWS, int J> struct GroupCell {
    template <int LS, class Ld>
    static BQS_HD void run(const Par &p, const Emis &e1, uint64_t rw_1, const Emis &e2, uint64_t rw_2, int a, int l_ref, double inv_a, double inv1, double inv2,
                           double m2o, double m8o, double (&Mp)[NB], double (&Ip)[NB], Ld Ln, const double (&bM)[NB], const double (&bI)[NB],
                           double &pm, double &pd, double &q1m, double &q1d, double &cM, double &cD, MapAcc &acc)
    {
        double i1_prev = 0., M1n = 0., D1n = 0., M1r = 0., D1r = 0.;
        if (J < NB) {
            double fd = m2o * pm + m8o * pd;
            if (EDGE) { const int idx = a - BW - 1 + J; fd = (idx < 0 || idx >= l_ref) ? 0. : fd; }
            pm = Mp[J < NB ? J : 0]; pd = fd;
            const double Mn = Mp[J < NB ? J : 0] * inv_a, In = Ip[J < NB ? J : 0] * inv_a, Dn = fd * inv_a;
            Mp[J < NB ? J : 0] = Mn; Ip[J < NB ? J : 0] = In;
            if (J > 0) i1_prev = (kEI * (p.m1 * Mn + p.m4 * In)) * inv1;                 // the forward pass's I[a + 1][J - 1]
            const double t3 = p.m0 * Mn + p.m3 * In + p.m6 * Dn;
            const double e = emis_cell<EDGE>(e1, rw_1, J < NB ? J : 0, t3);
            M1r = e * t3;                                                                  // the forward pass's raw M[a + 1][J] ...
            D1r = p.m2 * q1m + p.m8 * q1d;                                                 // ... and raw D[a + 1][J]
            if (EDGE) D1r = BQS_FLD(rw_1, J < NB ? J : 0) == 7 ? 0. : D1r;
            M1n = M1r * inv1; D1n = D1r * inv1;
        }
        if (J > 0) {
            constexpr int C = J > 0 ? J - 1 : 0;                                           // the cell of row a + 1 that is complete now
            const double I1n = J < NB ? i1_prev : 0. * inv1;                               // (I of the last cell: 0, scaled like every other)
            if (ROWS == 3) {
                { d2 v = { cM, I1n }; Ln[C * LS] = v; }
                if (C > 0) {
                    const double fi2 = (kEI * (p.m1 * cM + p.m4 * I1n)) * inv2;           // the forward pass's I[a + 2][C - 1]
                    acc.template add<2 * (C > 0 ? C - 1 : 0) + 1>(fi2 * bI[C > 0 ? C - 1 : 0]);
                }
                const double t3 = p.m0 * cM + p.m3 * I1n + p.m6 * cD;
                const double e = emis_cell<EDGE>(e2, rw_2, C, t3);
                const double fm2 = (e * t3) * inv2;                                        // the forward pass's M[a + 2][C]
                acc.template add<2 * C>(fm2 * bM[C]);
            } else {
                acc.template add<2 * C>(cM * bM[C]);
                acc.template add<2 * C + 1>(I1n * bI[C]);
            }
        }
        if (J < NB) { q1m = M1r; q1d = D1r; cM = M1n; cD = D1n; }
        GroupCell<EDGE, ROWS, J + 1>::template run<LS>(p, e1, rw_1, e2, rw_2, a, l_ref, inv_a, inv1, inv2, m2o, m8o, Mp, Ip, Ln, bM, bI, pm, pd, q1m, q1d, cM, cD, acc);
    }
};
template <bool EDGE, int ROWS> struct GroupCell<EDGE, ROWS, NB + 1> {
    template <int LS, class Ld>
    static BQS_HD void run(const Par &, const Emis &, uint64_t, const Emis &, uint64_t, int, int, double, double, double, double, double, double (&)[NB], double (&)[NB], Ld,
                           const double (&)[NB], const double (&bI)[NB], double &, double &, double &, double &, double &, double &, MapAcc &acc)
    {
        if (ROWS == 3) acc.template add<2 * (NB - 1) + 1>(0. * bI[NB - 1]);      // I[a + 2][NB - 1] = 0: the last term of the row
    }
};

struct BwdState { double bM[NB], bI[NB]; uint64_t rw; uint32_t w_up; };

// band word of the row below, given that row's input word: its lower-end code comes in at cell 0
template <bool EDGE> BQS_HD uint64_t word_down(uint64_t rw, uint32_t w)
{
    if (EDGE || !TWOBIT) return ((rw << 3) | (uint64_t)((w >> 14) & 7u)) & WORD_MASK;
    return (uint64_t)((((uint32_t)rw << 2) | ((w >> 14) & 3u)) & WORD2_MASK);
}

// One group: the stored row a and the ROWS - 1 rows above it.  b.rw is the band word of row a + ROWS (of row lq when that is beyond the read: the
// group is the topmost and its top row IS row lq), b.w_up the input word of that row.  In the interior groups (EDGE false) the band words
// are the two-bit form (rw_pack2).
template <int LS, bool EDGE, int ROWS, int MODE, class Ld>
BQS_HD void bwd_group(const Par &p, int lq, int l_ref, int a, uint32_t *IN, const d2 *F2, const double *S, int ln, const float *q2p, Ld Ln, BwdCtx &c, BwdState &b)
{
    const int top = a + ROWS - 1;
    // the small inputs in front of the cell loads: loads come back in order
    const double s_top = *at<LS>(S, hot_row<MODE>(top), ln), s_a = *at<LS>(S, hot_row<MODE>(a), ln);
    const double s_mid = ROWS == 3 ? *at<LS>(S, hot_row<MODE>(a + 1), ln) : 0.;
    const uint32_t w_top = *at<LS>(IN, hot_row<MODE>(top), ln), w_a = *at<LS>(IN, hot_row<MODE>(a), ln);
    const uint32_t w_mid = ROWS == 3 ? *at<LS>(IN, hot_row<MODE>(a + 1), ln) : 0u;
    double Mp[NB], Ip[NB];
    {
        const d2 *row = at<LS>(F2, ((a - 1) / 3) * NB, ln);
#pragma unroll
        for (int j = 0; j < NB; ++j) { const d2 v = ld_d2<MODE>(row + j * LS); Mp[j] = v.x; Ip[j] = v.y; }
    }
    const bool is_top = EDGE && top >= lq;                       // the group's top row is row lq: b is the start vector, nothing to step from
    // S[] holds 1 / s[row] below row lq, s[lq] itself for row lq
    const double inv_top = is_top ? 1. / s_top : s_top;
    const bool row1 = EDGE && a == 1;                            // row 1 is stored normalised and has no D state
    const double inv_a_step = s_a;                               // the backward step to row a multiplies by 1 / s[a] whatever the row
    const double inv_a = row1 ? 1. : s_a;
    const double inv_mid = ROWS == 3 ? s_mid : inv_top;          // 1 / s[a + 1]
    uint64_t rw_top = b.rw;
    if (!is_top) {
        // band word of the top row from the one above it; b[top] from b[top + 1] with the emissions of row top + 1
        const Emis e_up = make_emis<EDGE>(b.w_up, b.rw, q2p);
        bwd_apply<EDGE, MODE>(p, e_up, b.rw, top, inv_top, b.bM, b.bI);
        sched_fence();
        rw_top = word_down<EDGE>(b.rw, w_top);
    }
    MapAcc acc;
    if (ROWS == 1) {
        // the stored row on its own (the topmost group of a read whose length is 1 mod 3): normalise, MAP
#pragma unroll
        for (int j = 0; j < NB; ++j) { Mp[j] *= inv_top; Ip[j] *= inv_top; }      // (a == top here)
        map_row(acc, Mp, Ip, b.bM, b.bI);
        finish_row<LS, MODE>(c, a, acc, w_a, IN, ln);
        b.rw = rw_top; b.w_up = w_a;
        return;
    }
    const uint64_t rw_1 = ROWS == 3 ? word_down<EDGE>(rw_top, w_mid) : rw_top;  // band word of row a + 1
    const Emis e2 = make_emis<EDGE>(w_top, rw_top, q2p);                          // emissions of row a + 2 (ROWS == 3)
    const Emis e1 = ROWS == 3 ? make_emis<EDGE>(w_mid, rw_1, q2p) : e2;           // emissions of row a + 1
    double pm = 0., pd = 0., q1m = 0., q1d = 0., cM = 0., cD = 0.;
    acc.init();
    GroupCell<EDGE, ROWS, 0>::template run<LS>(p, e1, rw_1, e2, rw_top, a, l_ref, inv_a, inv_mid, inv_top, row1 ? 0. : p.m2, row1 ? 0. : p.m8, Mp, Ip, Ln,
                                               b.bM, b.bI, pm, pd, q1m, q1d, cM, cD, acc);
    finish_row<LS, MODE>(c, top, acc, w_top, IN, ln);
    if (ROWS == 3) {
        bwd_apply<EDGE, MODE>(p, e2, rw_top, a + 1, inv_mid, b.bM, b.bI);
        map_row_lds<LS>(acc, Ln, b.bM, b.bI);
        finish_row<LS, MODE>(c, a + 1, acc, w_mid, IN, ln);
    }
    bwd_apply<EDGE, MODE>(p, e1, rw_1, a, inv_a_step, b.bM, b.bI);
    map_row(acc, Mp, Ip, b.bM, b.bI);
    finish_row<LS, MODE>(c, a, acc, w_a, IN, ln);
    b.rw = word_down<EDGE>(rw_1, w_a); b.w_up = w_a;
}

// all_edge as in fwd_lane.  Otherwise the groups whose rows a - 1 .. a + 3 have all cells inside the window take the interior code: loops, not a
// branch per group.  Ln: this lane's 15 (M, I) pairs of LDS, stride LS.
template <int LS, int MODE = 0, class Ld>
BQS_HD void bwd_lane(const Par &p, int lq, int l_ref, bool all_edge, uint32_t *IN, const d2 *F2, const double *S, int ln, const float *q2p, Ld Ln, BwdCtx &c)
{
    BwdState b;
    // band word of row lq: field j = code(lq - BW - 1 + j) = the upper-end code of row lq - 2 BW + j
    b.rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) b.rw |= (uint64_t)((*at<LS>(IN, lq - 2 * BW + j, ln) >> 11) & 7u) << (3 * j);
    {
        const double s_top = *at<LS>(S, lq, ln), sl1 = *at<LS>(S, lq + 1, ln);
        const double vM = p.sM / s_top / sl1, vI = p.sI / s_top / sl1;
#pragma unroll
        for (int j = 0; j < NB; ++j) { const bool valid = BQS_FLD(b.rw, j) != 7; b.bM[j] = valid ? vM : 0.; b.bI[j] = valid ? vI : 0.; }
    }
    c.run_r = 0;
    b.w_up = *at<LS>(IN, lq, ln);
    // the topmost group: the last stored row and the 0, 1 or 2 rows above it
    int a = 3 * ((lq - 1) / 3) + 1;
    const int above = lq - a;
    if (above == 0) bwd_group<LS, true, 1, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
    else if (above == 1) bwd_group<LS, true, 2, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
    else bwd_group<LS, true, 3, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
    a -= 3;
    const bool ae = all_edge || BQS_TEST_FORCE_EDGE;
    // interior groups: rows a .. a + 3 are all between row BW + 1 and row lq - 1 (row a + 3 lends its emissions to the first backward step)
    const int hi = ae ? 0 : lq - 4, lo = ae ? 1 : BW + 3;          // interior groups: lo <= a <= hi
#pragma unroll 1
    for (; a >= 1 && a > hi; a -= 3) bwd_group<LS, true, 3, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
    if (a >= lo && !ae) {
        b.rw = rw_pack2(b.rw);                                     // (row a + 3 is an interior row: nothing is lost)
#pragma unroll 1
        for (; a >= lo; a -= 3) bwd_group<LS, false, 3, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
        b.rw = rw_unpack3(b.rw);
    }
#pragma unroll 1
    for (; a >= 1; a -= 3) bwd_group<LS, true, 3, MODE>(p, lq, l_ref, a, IN, F2, S, ln, q2p, Ln, c, b);
}

// the left-hand running maximum (realn.c's extended BAQ: bq = min(left, right) inside the M operation) and the qualities' way home.
// Eight rows per step: the eight row words are asked for together and the eight bytes leave as one store (the read's qualities start on an
// 8-byte boundary) -- one row per step was a chain of 150 dependent load latencies.  Outside the M operation a row's word holds the staged
// byte in its working-quality field and b = 0, so every byte of the read can be written.
template <int LS>
BQS_HD void final_lane(int lq, const uint32_t *IN, int ln, const BwdCtx &c, uint8_t *qual)
{
    int run = 0;
    const int m0 = c.ys, m1 = c.ys + c.mlen;
    for (int q0 = 0; q0 < lq; q0 += 8) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = q0 + k < lq ? *at<LS>(IN, q0 + k + 1, ln) : 0u;
        uint64_t out = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = q0 + k;
            const bool in_m = q >= m0 && q < m1;
            const int b = in_m ? (int)((w[k] >> 17) & 127u) : 0;
            run = b > run ? b : run;
            const int q1 = (int)(w[k] >> 24);
            const int v = (in_m && !c.plain_mask && q1 > run) ? run : q1;
            out |= (uint64_t)(uint32_t)v << (8 * k);
        }
        if (q0 + 8 <= lq) __builtin_memcpy((uint8_t *)__builtin_assume_aligned(qual, 8) + q0, &out, 8);
        else for (int k = 0; q0 + k < lq; ++k) qual[q0 + k] = (uint8_t)(out >> (8 * k));
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
