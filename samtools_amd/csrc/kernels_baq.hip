// kernels_baq.hip -- BAQ (per-base alignment quality) on the device (gfx950, fp64 vector ALU).
//
// Replaces HTSlib realn.c sam_prob_realn() + probaln.c probaln_glocal() (absent from the reference
// tree; call site bam_plcmd.c:451 with flag 3, or 7 under -E).  Recurrences, operation order and
// scaling follow SURVEY.md Appendix A.4 / A.4.1 exactly; this file is compiled with
// -ffp-contract=off so no multiply-add is fused, and every sum is accumulated in the reference's
// order, which makes the integer BAQ values reproduce the CPU's.
//
// Mapping: one lane per read (reads are independent; each is a strictly sequential banded HMM).
// The forward matrix of a lane lives in an HBM scratch slab laid out [cell][lane] so that the 64
// lanes of a wave touch 64 consecutive doubles (coalesced 512-byte accesses); the backward pass
// keeps only two rolling rows.  Reads are processed in chunks sized to the scratch budget.
#include "dev_util.h"
#include <math.h>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>

#define EI .25
#define EM .33333333333

#include "baq_band7s.h"

// g_qual2prob; the MAP quality's thresholds (baq7s::LogTab, baq_band7s.h: 104 doubles in device memory, NULL = this host's log() did not
// give a clean step function and the kernels evaluate the formula)
struct BaqTables { float q2p[256]; const double *lt; };

__device__ __forceinline__ int nt16_int_dev(int c)   // seq_nt16_int
{
    return c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 4;
}

#define SET_U(u, b, i, k) { int x_ = (i) - (b); x_ = x_ > 0 ? x_ : 0; (u) = ((k) - x_ + 1) * 3; }

// per-read geometry computed once (realn.c)
struct BaqGeom { int lq, l_ref, bw, xb; };

__global__ void __launch_bounds__(64) k_baq(StaReadsDev R, StaWinDev W, BaqTables T, int redo, int64_t first, int64_t count,
                                            double *scratch, size_t dbl_per_read, int idim_max, int lq_max,
                                            int32_t *state_s, uint8_t *q_s)
{
    int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    int lane = threadIdx.x;
    if (t >= count) return;
    int64_t r = R.chain[1 + first + t];        // compact list of the reads the band-7 kernel does not take
    uint32_t info = R.info[r];
    if (!(info & RI_BAQ_SLOW) || ((info >> RI_BAQ_BW_SHIFT) & 31) != 0) return;   // band-8 reads share the list but have their own kernel

    const uint32_t *cigar = R.cigar + R.cig_off[r];
    int n_cigar = (int)(R.cig_off[r + 1] - R.cig_off[r]);
    int lq = R.l_qseq[r];
    uint64_t boff = (uint64_t)R.base_off8[r] << 3;
    uint8_t *qual = R.qual + boff;
    long long rpos = W.origin + R.pos[r];

    BaqGeo g = baq_geometry(cigar, n_cigar, rpos, lq, W.ref, W.ref_len);
    if (!g.ok) return;                          // probaln_glocal returns 0: qualities unchanged
    long long xb = g.xb;
    int l_ref = g.l_ref, l_query = lq, bw = g.bw;
    const char *ref = W.ref + xb;               // ref[k-1] -> code
#define REFC(k1) nt16_int_dev(nt16_from_char((unsigned char)ref[(k1)]))
#define QRYC(i1) nt16_int_dev(seq_nib(R.seq, boff >> 1, (i1)))
    int bw2 = bw * 2 + 1;
    int i_dim = bw2 * 3 + 6;
    if (i_dim > idim_max || l_query > lq_max) return;   // cannot happen: sizes come from the same function

    size_t wave_base = (size_t)(t >> 6) * 64 * dbl_per_read;
    double *fS = scratch + wave_base + lane;                                  // f[(row*idim_max + u)*64]
    double *bS = fS + (size_t)(lq_max + 1) * idim_max * 64;                   // two rolling rows
    double *sS = bS + (size_t)2 * idim_max * 64;                              // s[0..l_query+1]
#define Fc(i, u) fS[((size_t)(i) * idim_max + (size_t)(u)) * 64]
#define Bc(w, u) bS[((size_t)(w) * idim_max + (size_t)(u)) * 64]
#define Sc(i) sS[(size_t)(i) * 64]
    int32_t *state = state_s + (size_t)(t >> 6) * 64 * lq_max + lane;         // state[i*64]
    uint8_t *qq = q_s + (size_t)(t >> 6) * 64 * lq_max + lane;

    double m[9], sI, sM, bI, bM;
    const float cd = 0.001f, ce = 0.1f;     // probaln_par_t { float d, e; int bw; }
    sM = sI = 1. / (2 * l_query + 2);
    m[0] = (1 - cd - cd) * (1 - sM); m[1] = m[2] = cd * (1 - sM);
    m[3] = (1 - ce) * (1 - sI); m[4] = ce * (1 - sI); m[5] = 0.;
    m[6] = 1 - ce; m[7] = 0.; m[8] = ce;
    bM = (1 - cd) / l_ref; bI = cd / l_ref;

    int k;
    /*** forward ***/
    for (int u = 0; u < i_dim; ++u) Fc(0, u) = 0.;
    SET_U(k, bw, 0, 0);
    Fc(0, k) = 1.; Sc(0) = 1.;
    {   // f[1]
        double sum;
        int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1, _beg, _end;
        for (int u = 0; u < i_dim; ++u) Fc(1, u) = 0.;
        float q0 = T.q2p[qual[0]];
        int qy0 = QRYC(0);
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u;
            int rc = REFC(k - 1);
            double e = (rc > 3 || qy0 > 3) ? 1. : rc == qy0 ? 1. - q0 : q0 * EM;
            SET_U(u, bw, 1, k);
            double a = e * bM, b2 = EI * bI;
            Fc(1, u) = a; Fc(1, u + 1) = b2;
            sum += a + b2;
        }
        Sc(1) = sum;
        SET_U(_beg, bw, 1, beg); SET_U(_end, bw, 1, end); _end += 2;
        for (k = _beg; k <= _end; ++k) Fc(1, k) /= sum;
    }
    for (int i = 2; i <= l_query; ++i) {
        double sum, qli = T.q2p[qual[i - 1]];
        int beg = 1, end = l_ref, xx, _beg, _end;
        int qyi = QRYC(i - 1);
        xx = i - bw; beg = beg > xx ? beg : xx;
        xx = i + bw; end = end < xx ? end : xx;
        // boundary cells read by this row / the next one must be zero (calloc in the reference)
        for (int u = 0; u < 6; ++u) Fc(i, u) = 0.;
        { int ue; SET_U(ue, bw, i, end); for (int u = ue + 3; u < ue + 6 && u < i_dim; ++u) Fc(i, u) = 0.; }
        double dprev_m = 0., dprev_d = 0.;     // f[i][v01+0], f[i][v01+2] of the previous k (unscaled, this row)
        {
            int v01; SET_U(v01, bw, i, beg - 1);
            dprev_m = Fc(i, v01); dprev_d = Fc(i, v01 + 2);   // zeros just written
        }
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u, v11, v10;
            int rc = REFC(k - 1);
            double e = (rc > 3 || qyi > 3) ? 1. : rc == qyi ? 1. - qli : qli * EM;
            SET_U(u, bw, i, k); SET_U(v11, bw, i - 1, k - 1); SET_U(v10, bw, i - 1, k);
            double fm = e * (m[0] * Fc(i - 1, v11) + m[3] * Fc(i - 1, v11 + 1) + m[6] * Fc(i - 1, v11 + 2));
            double fi = EI * (m[1] * Fc(i - 1, v10) + m[4] * Fc(i - 1, v10 + 1));
            double fd = m[2] * dprev_m + m[8] * dprev_d;
            Fc(i, u) = fm; Fc(i, u + 1) = fi; Fc(i, u + 2) = fd;
            sum += fm + fi + fd;
            dprev_m = fm; dprev_d = fd;
        }
        Sc(i) = sum;
        SET_U(_beg, bw, i, beg); SET_U(_end, bw, i, end); _end += 2;
        double inv = 1. / sum;
        for (k = _beg; k <= _end; ++k) Fc(i, k) *= inv;
    }
    {   // f[l_query+1]
        double sum = 0.;
        for (k = 1; k <= l_ref; ++k) {
            int u;
            SET_U(u, bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            sum += Fc(l_query, u) * sM + Fc(l_query, u + 1) * sI;
        }
        Sc(l_query + 1) = sum;
    }
    /*** backward + MAP, row by row ***/
    int cur = 0;
    for (int u = 0; u < i_dim; ++u) { Bc(0, u) = 0.; Bc(1, u) = 0.; }
    {
        double sl = Sc(l_query), sl1 = Sc(l_query + 1);
        for (k = 1; k <= l_ref; ++k) {
            int u;
            SET_U(u, bw, l_query, k);
            if (u < 3 || u >= bw2 * 3 + 3) continue;
            Bc(cur, u) = sM / sl / sl1; Bc(cur, u + 1) = sI / sl / sl1;
        }
    }
    for (int i = l_query; i >= 1; --i) {
        int beg = 1, end = l_ref, xx;
        xx = i - bw; beg = beg > xx ? beg : xx;
        xx = i + bw; end = end < xx ? end : xx;
        if (i < l_query) {
            // b[i] from b[i+1]
            int nxt = cur ^ 1;       // row i goes to nxt, row i+1 is cur
            int _beg, _end;
            double yv = (i > 1), qli1 = T.q2p[qual[i]];
            int qyi1 = QRYC(i);
            for (int u = 0; u < i_dim; ++u) Bc(nxt, u) = 0.;
            double dnext = 0.;       // b[i][v01+2] of k+1 (unscaled, this row)
            for (k = end; k >= beg; --k) {
                int u, v11, v10;
                SET_U(u, bw, i, k); SET_U(v11, bw, i + 1, k + 1); SET_U(v10, bw, i + 1, k);
                double e;
                if (k >= l_ref) e = 0 * Bc(cur, v11);
                else {
                    int rc = REFC(k);
                    e = ((rc > 3 || qyi1 > 3) ? 1. : rc == qyi1 ? 1. - qli1 : qli1 * EM) * Bc(cur, v11);
                }
                double bi1_i = Bc(cur, v10 + 1);
                double bm = e * m[0] + EI * m[1] * bi1_i + m[2] * dnext;
                double bi_ = e * m[3] + EI * m[4] * bi1_i;
                double bd = (e * m[6] + m[8] * dnext) * yv;
                Bc(nxt, u) = bm; Bc(nxt, u + 1) = bi_; Bc(nxt, u + 2) = bd;
                dnext = bd;
            }
            SET_U(_beg, bw, i, beg); SET_U(_end, bw, i, end); _end += 2;
            double ys = 1. / Sc(i);
            for (k = _beg; k <= _end; ++k) Bc(nxt, k) *= ys;
            cur = nxt;
        }
        // MAP for row i
        double sum = 0., max = 0.;
        int max_k = -1;
        for (k = beg; k <= end; ++k) {
            int u;
            double z;
            SET_U(u, bw, i, k);
            z = Fc(i, u) * Bc(cur, u); if (z > max) max = z, max_k = (k - 1) << 2 | 0; sum += z;
            z = Fc(i, u + 1) * Bc(cur, u + 1); if (z > max) max = z, max_k = (k - 1) << 2 | 1; sum += z;
        }
        max /= sum;
        state[(size_t)(i - 1) * 64] = max_k;
        int kq;
        if (T.lt) kq = baq7s::map_quality_x(1. - max, T.lt);        // the exact threshold table (baq_band7s.h), straight from device memory: a rare path
        else {
            double v = -4.343 * log(1. - max) + .499;
            // (int)v with x86 cvttsd2si semantics for out-of-range / NaN (the CPU reference's behaviour)
            kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;
        }
        qq[(size_t)(i - 1) * 64] = (uint8_t)(kq > 100 ? 99 : kq);
    }
    // note: e == 0*b for k >= l_ref keeps the reference's NaN/Inf propagation (cells are finite here)

    /*** realn.c: turn (state, q) into the BAQ and apply it (flag bit 1 always set by mpileup; bit 2 = extended) ***/
    // extended BAQ: within each M block bq = min(running max from the left, running max from the right)
    {
        long long xx = rpos; int yy = 0;
        for (int c = 0; c < n_cigar; ++c) {
            int op = cigar[c] & 0xf, l = (int)(cigar[c] >> 4);
            if (cg_is_mop(op)) {
                if (l > lq - yy) l = lq - yy;
                if (l > 0) {
                    // pass 1: bq[i] (0 unless the MAP state is M on the expected diagonal); left running max in place
                    int run = 0;
                    for (int i = yy; i < yy + l; ++i) {
                        int st = state[(size_t)i * 64];
                        int b = ((st & 3) != 0 || (long long)(st >> 2) != xx - xb + (i - yy)) ? 0 : (int)qq[(size_t)i * 64];
                        state[(size_t)i * 64] = b;                 // raw bq
                        run = b > run ? b : run;
                        qq[(size_t)i * 64] = (uint8_t)run;          // left[i]
                    }
                    // pass 2: right running max, combine, finalise and apply
                    run = 0;
                    for (int i = yy + l - 1; i >= yy; --i) {
                        int b = state[(size_t)i * 64];
                        run = b > run ? b : run;                    // rght[i]
                        int left = qq[(size_t)i * 64];
                        int bqv = W.baq_plain ? b : (left < run ? left : run);    // plain (calmd -r without -E): min(qual, q) per base
                        int q0 = qual[i];
                        int tag = 64 + (q0 <= bqv ? 0 : q0 - bqv);  // bq[i] as stored in ZQ
                        qual[i] = (uint8_t)(q0 - (tag - 64));
                    }
                }
                xx += l; yy += l;
            } else if (op == CG_S || op == CG_I) {
                if (l > lq - yy) l = lq - yy;
                yy += l;
            } else if (op == CG_D) xx += l;
        }
    }
#undef Fc
#undef Bc
#undef Sc
#undef REFC
#undef QRYC
}

// scratch sizing: the engine passes n_reads only; geometry bounds come from the staged window

size_t sta_baq_scratch_bytes(int64_t n_reads, int max_lq, int max_bw)
{
    (void)n_reads; (void)max_lq; (void)max_bw;
    return (size_t)3 << 30;     // fixed 3 GiB slab, reads are processed in chunks that fit it
}

// The tables the BAQ kernels take by value: g_qual2prob, and a pointer to the MAP-quality threshold table in device memory.  The host
// copies are built once (function-local static: thread safe); the device copy exists once PER DEVICE, made under a lock by the first
// launch on that device -- an engine on a second device (sta_engine_create(device = N)) or a second device thread of a driver
// (STA_DEV_THREADS) must never see another device's pointer or a half-built table.
static BaqTables baq_tables(bool *logtab_ok = nullptr);
void sta_launch_baq(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int redo, void *scratch, size_t scratch_bytes,
                         int lq_max, int bw_max, int64_t n_slow)
{
    if (r.n == 0 || lq_max <= 0 || n_slow <= 0) return;
    const BaqTables g_tables = baq_tables();
    int idim_max = (bw_max * 2 + 1) * 3 + 6;
    size_t dbl_per_read = (size_t)(lq_max + 1) * idim_max + (size_t)2 * idim_max + (size_t)(lq_max + 2);
    size_t bytes_per_read = dbl_per_read * 8 + (size_t)lq_max * 5;
    size_t chunk = scratch_bytes / bytes_per_read;
    chunk &= ~(size_t)63;
    if (chunk < 64) chunk = 64;      // caller guarantees the slab holds at least one wave
    if (chunk > (size_t)1 << 20) chunk = (size_t)1 << 20;
    double *dscr = (double *)scratch;
    int32_t *state_s = (int32_t *)(dscr + chunk * dbl_per_read);
    uint8_t *q_s = (uint8_t *)(state_s + chunk * (size_t)lq_max);
    for (int64_t first = 0; first < n_slow; first += (int64_t)chunk) {
        int64_t count = n_slow - first < (int64_t)chunk ? n_slow - first : (int64_t)chunk;
        unsigned nb = (unsigned)((count + 63) / 64);
        hipLaunchKernelGGL(k_baq, dim3(nb), dim3(64), 0, s, r, w, g_tables, redo, first, count, dscr, dbl_per_read, idim_max, lq_max,
                           state_s, q_s);
    }
}

// ================================================================================================
// Band-in-registers kernels: the common case.  realn.c picks bw = 7 unless the alignment's net
// indel is large, and probaln_glocal widens it to |l_ref - l_query| (8 for reads with a 2-3 bp
// deletion), so BW = 7 and BW = 8 cover nearly every read; the rest go to k_baq above.
// One lane per read as above, but the band lives in registers in diagonal-relative coordinates
//     j = k - i + BW   (k = 1-based reference index, i = query row, 0 <= j < NB = 2*BW+1)
//     (i-1,k-1) -> j      (i-1,k) -> j+1      (i,k-1) -> j-1
// so every row update is a fully unrolled, statically indexed sweep over NB x {M,I,D} doubles and
// the only bulk memory traffic is ONE coalesced store (forward kernel) and ONE coalesced load
// (backward kernel) per (row, M/I cell): 2*NB doubles per row per read, laid out
// [row][cell][lane] inside the wave's scratch slot (512 B per wave access).  The backward kernel
// keeps its row in registers too (the backward D state is only a running scalar).  Cells outside
// 1 <= k <= l_ref are held at exactly 0.0 (the reference never touches them: calloc), which keeps
// every sum bit-identical because x + 0.0 == x for the non-negative values involved.
// Per-row inputs (quality, query base, new reference base) are fetched two rows ahead and
// converted (LDS tables) one row ahead, so no row waits on memory.
// Arithmetic order is the reference's; the file is built with -ffp-contract=off.

__device__ __forceinline__ double emis_sel(int rc, int qyc, double ematch, double e_lo)
{
    // rc: reference code 0..3, 4 = ambiguous, 7 = outside the reference window
    double e = (rc == qyc) ? ematch : e_lo;
    double hi = (rc == 7) ? 0. : 1.;
    return rc > 3 ? hi : e;
}
#define FLD(w, j) ((int)((uint32_t)((w) >> (3 * (j))) & 7u))
// (M, I) of one band cell travel together: 16 bytes per lane and store (the forward kernel is store-ISSUE bound, so half as
// many, twice as wide stores; 1 KiB per wave access)
typedef double baq_d2 __attribute__((ext_vector_type(2)));

struct BaqRd {           // what both kernels need to know about one read
    const uint32_t *cigar; int n_cigar, lq, l_ref;
    uint8_t *qual; const uint8_t *seq; const char *ref;
    long long rpos, xb;
};
__device__ __forceinline__ BaqRd baq_rd(const StaReadsDev &R, const StaWinDev &W, int64_t r)
{
    BaqRd d;
    d.cigar = R.cigar + R.cig_off[r];
    d.n_cigar = (int)(R.cig_off[r + 1] - R.cig_off[r]);
    d.lq = R.l_qseq[r];
    uint64_t boff = (uint64_t)R.base_off8[r] << 3;
    d.qual = R.qual + boff;
    d.seq = R.seq + (boff >> 1);
    d.rpos = W.origin + R.pos[r];
    BaqGeo g = baq_geometry(d.cigar, d.n_cigar, d.rpos, d.lq, W.ref, W.ref_len);
    d.xb = g.xb; d.l_ref = g.l_ref;
    d.ref = W.ref + g.xb;
    return d;
}
struct BaqPar { double m0, m1, m2, m3, m4, m6, m8, sM, sI, bM, bI, eim1, eim4; };
__device__ __forceinline__ BaqPar baq_par(int lq, int l_ref)
{
    BaqPar p;
    const float cd = 0.001f, ce = 0.1f;     // probaln_par_t { float d, e; int bw; }
    p.sM = p.sI = 1. / (2 * lq + 2);
    p.m0 = (1 - cd - cd) * (1 - p.sM); p.m1 = p.m2 = cd * (1 - p.sM);
    p.m3 = (1 - ce) * (1 - p.sI); p.m4 = ce * (1 - p.sI);
    p.m6 = 1 - ce; p.m8 = ce;
    p.bM = (1 - cd) / l_ref; p.bI = cd / l_ref;
    p.eim1 = EI * p.m1; p.eim4 = EI * p.m4;
    return p;
}

#define RCODE(idx) (((idx) >= 0 && (idx) < l_ref) ? (int)refc[(unsigned char)ref[(idx)]] : 7)
#define RRAW(idx) (((idx) >= 0 && (idx) < l_ref) ? (int)(unsigned char)ref[(idx)] : 256)
#define RCONV(raw) ((raw) < 256 ? (int)refc[(raw)] : 7)
#define SEQB(i0) ((int)seq[(i0) >> 1])
#define QCONV(sb, i0) nt16_int_dev(((sb) >> ((~(i0) & 1) << 2)) & 0xf)

// which lane handles which read: direct (group of 64 consecutive reads) or through the list in `chain`
__device__ __forceinline__ int64_t baq_pick(const StaReadsDev &R, int64_t g, int lane, int use_list, int bw)
{
    int64_t r;
    if (use_list) {
        int64_t t = g * 64 + lane;
        if (t >= R.chain[0]) return -1;
        r = R.chain[1 + t];
    } else {
        r = g * 64 + lane;
        if (r >= R.n) return -1;
    }
    uint32_t info = R.info[r];
    if (!(info & RI_BAQ) || (int)((info >> RI_BAQ_BW_SHIFT) & 31) != bw) return -1;
    if (((info & RI_BAQ_SLOW) != 0) != (use_list != 0)) return -1;
    return r;
}

// ---- scratch layout of one group (64 reads = one wave), in "lane rows" of 64 doubles (512 B) ----
// Forward rows.  DEC (band width 7): the I state is stored only for ODD rows; an even row's I is a function of the row below it,
//     I[i][j] = (EI * (m1 * M[i-1][j+1] + m4 * I[i-1][j+1])) * (1 / s[i]),
// which the backward kernel re-evaluates with the forward kernel's operations in the forward kernel's order, so the stream is
// 3 * NB doubles per row PAIR instead of 4 * NB (-25 %).  Pair t = (i - 1) / 2 starts at lane row t * 3NB: the odd row's cells are
// (M, I) pairs of 16 bytes per lane (two lane rows per cell), the even row's are 8-byte M values after them.
// DEC == 2 (band width 7, the default): only the ODD rows are stored, and -- except row 1 -- RAW, before the division by the row
// sum s[i].  The backward kernel redoes `x * (1 / s[i])` (the forward kernel's own operation on the same operands), re-runs the
// odd row's D chain from the raw M values, and re-evaluates the whole even row above it from the normalised (M, I, D) with the
// forward kernel's expressions: bit-identical, 2 * NB doubles per row PAIR (-33 % against DEC == 1, -50 % against every row).
// Pair t = (i - 1) / 2 (i odd) at lane row t * 2NB.
// Without DEC every row stores (M, I) pairs: row i at lane row (i - 1) * 2NB.
// After the rows: s[0 .. lq_cap + 1] (one lane row each), then -- only when the per-row states do not fit LDS -- one int32 per row.
template <int NB, int DEC> __host__ __device__ constexpr size_t baq_rows_lr(int lq_cap) { return DEC == 2 ? (size_t)((lq_cap + 1) / 2) * (2 * NB) : DEC ? (size_t)((lq_cap + 1) / 2) * (3 * NB) : (size_t)lq_cap * (2 * NB); }
#define BAQ_LDS_ROWS_MAX 256          // per-row state bytes live in LDS up to this read length (16 KB per wave)

template <int BW, int DEC>
__device__ __forceinline__ void baq_fwd_body(const StaReadsDev &R, const StaWinDev &W, const float *q2p, const uint8_t *refc, int64_t g0, int64_t ngroups, int use_list,
                                             double *scratch, size_t slot_dbl, int lq_cap, int64_t vb /* the workgroup's place in the launch's groups (blockIdx.x, or a strided walk) */)
{
    constexpr int NB = 2 * BW + 1;
    const int lane = threadIdx.x & 63;
    const int64_t gl = vb * 4 + (threadIdx.x >> 6);      // group within this launch == scratch slot
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    double *slot = scratch + (size_t)gl * slot_dbl;
    baq_d2 *F2 = reinterpret_cast<baq_d2 *>(slot) + lane;
    double *F1 = slot + lane;
    double *S = slot + baq_rows_lr<NB, DEC>(lq_cap) * 64 + lane;
    const BaqRd d = baq_rd(R, W, r);
    const int lq = d.lq, l_ref = d.l_ref;
    const uint8_t *qual = d.qual, *seq = d.seq; const char *ref = d.ref;
    const BaqPar p = baq_par(lq, l_ref);

    double M[NB], I[NB], D[NB];
    uint64_t rw = 0;                 // 3-bit field j = code of reference index (i - BW - 1 + j) for the current row i
#pragma unroll
    for (int j = 0; j < NB; ++j) rw |= (uint64_t)RCODE(j - BW) << (3 * j);

    S[0] = 1.;
    {   // row 1: k = j - BW + 1
        int qy = QCONV(SEQB(0), 0);
        double q0 = q2p[qual[0]];
        double ematch = 1. - q0, e_lo = qy > 3 ? 1. : q0 * EM;
        int qyc = qy > 3 ? 9 : qy;
        double sum = 0.;
        const double eibi = EI * p.bI;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int rc = FLD(rw, j);
            double e = emis_sel(rc, qyc, ematch, e_lo);
            double a = e * p.bM;
            double b2 = rc == 7 ? 0. : eibi;
            M[j] = a; I[j] = b2; D[j] = 0.;
            sum += a + b2;
        }
        S[64] = sum;
#pragma unroll
        for (int j = 0; j < NB; ++j) { M[j] /= sum; I[j] /= sum; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &F2[(size_t)(DEC ? 2 * j : j) * (DEC ? 32 : 64)]); }
    }
    // software pipeline of the per-row inputs: raw bytes two rows ahead, converted one row ahead
    int c_sb = 0, c_rc = 7;                   // converted, for the next row
    float c_qf = 0.f;
    int r_q = 0, r_sb = 0, r_rr = 256;        // raw, for the row after
    if (lq >= 2) { c_qf = q2p[qual[1]]; c_sb = SEQB(1); c_rc = RCODE(2 + BW - 1); }
    if (lq >= 3) { r_q = qual[2]; r_sb = SEQB(2); r_rr = RRAW(3 + BW - 1); }
#pragma unroll 1
    for (int i = 2; i <= lq; ++i) {
        const float qf = c_qf; const int sb = c_sb; const int nrc = c_rc;
        // convert what was fetched during the previous row, fetch for row i+2
        c_qf = q2p[r_q]; c_sb = r_sb; c_rc = RCONV(r_rr);
        if (i + 2 <= lq) { r_q = qual[i + 1]; r_sb = SEQB(i + 1); r_rr = RRAW(i + 2 + BW - 1); }
        rw = (rw >> 3) | ((uint64_t)nrc << (3 * (NB - 1)));
        const int qy = QCONV(sb, i - 1);
        const double qli = qf;
        const double ematch = 1. - qli, e_lo = qy > 3 ? 1. : qli * EM;
        const int qyc = qy > 3 ? 9 : qy;
        double sum = 0., pm = 0., pd = 0.;    // pm, pd: this row's M and D at j-1
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int rc = FLD(rw, j);
            double e = emis_sel(rc, qyc, ematch, e_lo);
            double fm = e * (p.m0 * M[j] + p.m3 * I[j] + p.m6 * D[j]);
            double fi = (j + 1 < NB) ? EI * (p.m1 * M[j + 1] + p.m4 * I[j + 1]) : 0.;
            double fd = p.m2 * pm + p.m8 * pd;
            fd = rc == 7 ? 0. : fd;
            M[j] = fm; I[j] = fi; D[j] = fd;
            sum += fm + fi + fd;
            pm = fm; pd = fd;
        }
        S[(size_t)i * 64] = sum;
        if (DEC == 2 && (i & 1)) {           // raw (M, I) of an odd row; even rows are not stored at all
            const size_t t2 = (size_t)((i - 1) >> 1) * (2 * NB);
#pragma unroll
            for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &F2[(t2 + 2 * j) * 32]); }
        }
        double inv = 1. / sum;
#pragma unroll
        for (int j = 0; j < NB; ++j) { M[j] *= inv; I[j] *= inv; D[j] *= inv; }
        if (DEC == 2) {
        } else if (DEC) {
            const size_t t = (size_t)((i - 1) >> 1) * (3 * NB);
            if (i & 1) {
#pragma unroll
                for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &F2[(t + 2 * j) * 32]); }
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) __builtin_nontemporal_store(M[j], &F1[(t + 2 * NB + j) * 64]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &F2[((size_t)(i - 1) * NB + j) * 64]); }
        }
    }
    {   // s[l_query+1]
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += M[j] * p.sM + I[j] * p.sI;
        S[(size_t)(lq + 1) * 64] = sum;
    }
}

// the tables both passes look things up in: q2p = g_qual2prob, refc = reference character -> code 0..3 / 4
#define BAQ_TABLES_INIT()                                                                                                   \
    __shared__ float q2p[256];                                                                                              \
    __shared__ uint8_t refc[256];                                                                                           \
    __shared__ double lt_s[baq7s::LT_N];                                                                                    \
    const bool lt_ok = T.lt != nullptr;                                                                                     \
    for (int k_ = threadIdx.x; k_ < 256; k_ += blockDim.x) { q2p[k_] = T.q2p[k_]; refc[k_] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)k_)); } \
    if (lt_ok) for (int k_ = threadIdx.x; k_ < baq7s::LT_N; k_ += blockDim.x) lt_s[k_] = T.lt[k_];                          \
    __syncthreads();

template <int BW, int DEC>
__global__ void __launch_bounds__(256) k_baq_fwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                 double *scratch, size_t slot_dbl, int lq_cap)
{
    BAQ_TABLES_INIT()
    baq_fwd_body<BW, DEC>(R, W, q2p, refc, g0, ngroups, use_list, scratch, slot_dbl, lq_cap, (int64_t)blockIdx.x);
}

// ---- backward + MAP + apply ----
// Per-lane CIGAR cursor walking the query backwards: which operation covers query index q, and where it starts (realn.c's block
// walk, evaluated on the fly so that the per-row MAP state never leaves the wave).
struct BaqCur { int c; int ys, xs_off; int qlen; bool is_m; };      // op index, its first query index, its first reference index minus xb
__device__ __forceinline__ void baq_cur_seek(BaqCur &cu, const uint32_t *cigar, int q)
{
    // move to earlier operations until the current one covers q (operations that consume no query base are stepped over)
    while (cu.c > 0 && (q < cu.ys || cu.qlen == 0)) {
        --cu.c;
        const uint32_t w = cigar[cu.c];
        const int op = w & 0xf, l = (int)(w >> 4);
        const bool qop = cg_is_mop(op) || op == CG_I || op == CG_S, rop = cg_is_mop(op) || op == CG_D;   // (no N: such reads are never re-aligned)
        cu.qlen = qop ? l : 0;
        cu.is_m = cg_is_mop(op);
        if (qop) cu.ys -= l;
        if (rop) cu.xs_off -= l;
    }
}

template <int BW, int DEC, bool PLDS>
__device__ __forceinline__ void baq_bwd_body(const StaReadsDev &R, const StaWinDev &W, const float *q2p, const uint8_t *refc, const double *lt_tab, uint8_t *baq_state, int64_t g0, int64_t ngroups,
                                             int use_list, double *scratch, size_t slot_dbl, int lq_cap, int lds_rows, int64_t vb)
{
    constexpr int NB = 2 * BW + 1;
    const int lane = threadIdx.x & 63;
    const int64_t gl = vb * 4 + (threadIdx.x >> 6);
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    double *slot = scratch + (size_t)gl * slot_dbl;
    const baq_d2 *F2 = reinterpret_cast<const baq_d2 *>(slot) + lane;
    const double *F1 = slot + lane;
    const double *S = slot + baq_rows_lr<NB, DEC>(lq_cap) * 64 + lane;
    // per-row MAP result b (0 unless the MAP state is M on the read's own diagonal): a byte in LDS, or an int in the scratch slot
    uint8_t *Pl = baq_state + (size_t)(threadIdx.x >> 6) * (size_t)lds_rows * 64 + lane;
    int32_t *Pg = reinterpret_cast<int32_t *>(slot + baq_rows_lr<NB, DEC>(lq_cap) * 64 + (size_t)(lq_cap + 2) * 64) + lane;
    const BaqRd d = baq_rd(R, W, r);
    const int lq = d.lq, l_ref = d.l_ref;
    uint8_t *qual = d.qual; const uint8_t *seq = d.seq; const char *ref = d.ref;
    const BaqPar p = baq_par(lq, l_ref);
    const bool plain = W.baq_plain != 0;

    // CIGAR cursor: starts behind the last operation
    BaqCur cu; cu.c = d.n_cigar; cu.qlen = 0; cu.is_m = false;
    {
        int tq = 0; long long tx = d.rpos;
        for (int c = 0; c < d.n_cigar; ++c) {
            const int op = d.cigar[c] & 0xf, l = (int)(d.cigar[c] >> 4);
            if (cg_is_mop(op)) { tq += l; tx += l; } else if (op == CG_I || op == CG_S) tq += l; else if (op == CG_D) tx += l;
        }
        cu.ys = tq; cu.xs_off = (int)(tx - d.xb);
    }
    int run_r = 0, run_c = -1;           // running maximum of b from the right inside the current M operation

    // row lq: field j = code(lq - BW - 1 + j); it doubles as the backward word of row lq-1
    uint64_t rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) rw |= (uint64_t)RCODE(lq - BW - 1 + j) << (3 * j);
    double bMr[NB], bIr[NB];
    const double s_top = S[(size_t)lq * 64];
    {
        double sl = s_top, sl1 = S[(size_t)(lq + 1) * 64];
        double vM = p.sM / sl / sl1, vI = p.sI / sl / sl1;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bool valid = FLD(rw, j) != 7;             // row lq: 1 <= k <= l_ref
            bMr[j] = valid ? vM : 0.; bIr[j] = valid ? vI : 0.;
        }
    }
    // pipeline: row i needs qual[i], seq(i), S[i], code(i - BW) (all for i < lq)
    float c_qf = 0.f; int c_sb = 0, c_rc = 7; double c_s = 1.;
    int r_q = 0, r_sb = 0, r_rr = 256; double r_s = 1.;
    if (lq >= 2) { int i = lq - 1; c_qf = q2p[qual[i]]; c_sb = SEQB(i); c_s = S[(size_t)i * 64]; }
    if (lq >= 3) { int i = lq - 2; r_q = qual[i]; r_sb = SEQB(i); r_s = S[(size_t)i * 64]; r_rr = RRAW(i - BW); }

    // One row: backward update (b[i] from b[i+1]) unless i == lq, then the MAP step against the forward row.  Forward row of
    // row i: (fM, fI) loaded, or -- EVEN row under DEC -- fM loaded and fI re-evaluated from the row below (Mp, Ip) and 1 / s[i].
    double inv_i = 1. / s_top;
    // row inputs of the backward step i (the emission of row i + 1): advances the prefetch pipeline and the reference word
    int u_qyc = 9; double u_ematch = 0., u_elo = 1., u_yv = 0., u_si = 1.;
#define BAQ_BWD_INPUTS(i)                                                                                                   \
    {                                                                                                                       \
        const float qf = c_qf; const int sb = c_sb; const int nrc = c_rc; u_si = c_s;                                       \
        c_qf = q2p[r_q]; c_sb = r_sb; c_rc = RCONV(r_rr); c_s = r_s;                                                        \
        if ((i) - 2 >= 1) { int i2 = (i) - 2; r_q = qual[i2]; r_sb = SEQB(i2); r_s = S[(size_t)i2 * 64]; r_rr = RRAW(i2 - BW); } \
        if ((i) < lq - 1) rw = (rw << 3) | (uint64_t)nrc;                                                                   \
        const int qy = QCONV(sb, (i));                                                                                      \
        const double qli1 = qf;                                                                                             \
        u_ematch = 1. - qli1; u_elo = qy > 3 ? 1. : qli1 * EM;                                                              \
        u_qyc = qy > 3 ? 9 : qy;                                                                                            \
        u_yv = (i) > 1 ? 1. : 0.;                                                                                           \
    }
#define BAQ_BWD_APPLY(i)                                                                                                    \
    {                                                                                                                       \
        double dnext = 0.;                                                                                                  \
        _Pragma("unroll")                                                                                                   \
        for (int j = NB - 1; j >= 0; --j) {                                                                                 \
            int rc = FLD(rw, j);                                                                                            \
            double e = emis_sel(rc, u_qyc, u_ematch, u_elo) * bMr[j];     /* rc == 7 <=> k >= l_ref: e = 0 * b */            \
            double bi1 = j > 0 ? bIr[j - 1] : 0.;                                                                           \
            double bm = e * p.m0 + p.eim1 * bi1 + p.m2 * dnext;                                                             \
            double bi_ = e * p.m3 + p.eim4 * bi1;                                                                           \
            double bd = (e * p.m6 + p.m8 * dnext) * u_yv;                                                                   \
            bMr[j] = bm; bIr[j] = bi_;                                                                                      \
            dnext = bd;                                                                                                     \
        }                                                                                                                   \
        inv_i = 1. / u_si;                                                                                                  \
        _Pragma("unroll")                                                                                                   \
        for (int j = 0; j < NB; ++j) { bMr[j] *= inv_i; bIr[j] *= inv_i; }                                                  \
        if ((i) <= BW) {            /* cells with k < 1 do not exist in the reference: keep them at zero */                  \
            _Pragma("unroll")                                                                                               \
            for (int j = 0; j < BW; ++j) if (j < BW + 1 - (i)) { bMr[j] = 0.; bIr[j] = 0.; }                                \
        }                                                                                                                   \
    }
#define BAQ_BWD_UPDATE(i) if ((i) < lq) { BAQ_BWD_INPUTS(i) BAQ_BWD_APPLY(i) }
    // MAP of row i given z-terms; then the per-row result: quality from the right-hand running maximum straight away (the
    // left-hand one follows in the forward pass at the end), state byte kept for that pass
#define BAQ_MAP_FINISH(i, sum, max, max_k)                                                                                  \
    {                                                                                                                       \
        double mx_ = (max) / (sum);                                                                                         \
        int kq;                                                                                                             \
        if (lt_tab) kq = baq7s::map_quality_x(1. - mx_, (baq7s::LtPtr)lt_tab);   /* the exact threshold table (baq_band7s.h), in LDS */ \
        else {                                                                                                              \
            double v = -4.343 * log(1. - mx_) + .499;                                                                       \
            kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;                                   \
        }                                                                                                                   \
        kq = (int)(uint8_t)(kq > 100 ? 99 : kq);                                                                            \
        if (PLDS) {                                                                                                         \
            const int q = (i) - 1;                                                                                          \
            baq_cur_seek(cu, d.cigar, q);                                                                                   \
            int b = 0;                                                                                                      \
            const bool in_m = cu.is_m && q >= cu.ys && q < cu.ys + cu.qlen;                                                 \
            if (in_m && ((max_k) & 3) == 0 && ((max_k) >> 2) == cu.xs_off + (q - cu.ys)) b = kq;                            \
            Pl[(size_t)q * 64] = (uint8_t)b;                                                                                \
            if (in_m) {                                                                                                     \
                if (run_c != cu.c) { run_c = cu.c; run_r = 0; }                                                             \
                run_r = b > run_r ? b : run_r;                                                                              \
                const int q0 = qual[q];                                                                                     \
                const int lim = plain ? b : run_r;                                                                          \
                if (q0 > lim) qual[q] = (uint8_t)lim;                                                                       \
            }                                                                                                               \
        } else Pg[(size_t)((i) - 1) * 64] = (int32_t)(((uint32_t)(max_k) << 8) | (uint32_t)kq);                             \
    }

    int i = lq;
    if (!DEC || (lq & 1)) {
        // (without DEC: every row this way)  top row odd: a fully stored row on its own
#pragma unroll 1
        for (; i >= 1; --i) {
            double fM[NB], fI[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                baq_d2 v = __builtin_nontemporal_load(DEC == 2 ? &F2[((size_t)((i - 1) >> 1) * (2 * NB) + 2 * j) * 32]
                                                      : DEC ? &F2[((size_t)((i - 1) >> 1) * (3 * NB) + 2 * j) * 32] : &F2[((size_t)(i - 1) * NB + j) * 64]);
                fM[j] = v.x; fI[j] = v.y;
            }
            BAQ_BWD_UPDATE(i)
            if (DEC == 2 && i > 1) {         // a raw row (the top row of a read of odd length): the forward kernel's M *= inv, I *= inv
#pragma unroll
                for (int j = 0; j < NB; ++j) { fM[j] *= inv_i; fI[j] *= inv_i; }
            }
            double sum = 0., max = 0.; int max_k = -1;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                double z;
                z = fM[j] * bMr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 0; } sum += z;
                z = fI[j] * bIr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 1; } sum += z;
            }
            BAQ_MAP_FINISH(i, sum, max, max_k)
            if (DEC) { --i; break; }
        }
    }
    if (DEC == 2) {
        // pairs (i even, i - 1 odd): only the odd row was stored -- raw, unless it is row 1
#pragma unroll 1
        for (; i >= 2; i -= 2) {
            const size_t t2 = (size_t)((i - 1) >> 1) * (2 * NB);
            double Mp[NB], Ip[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) { baq_d2 v = __builtin_nontemporal_load(&F2[(t2 + 2 * j) * 32]); Mp[j] = v.x; Ip[j] = v.y; }
            BAQ_BWD_UPDATE(i)
            // inputs of the backward step i - 1 = the emission of row i, which the re-evaluation of that row needs first
            BAQ_BWD_INPUTS(i - 1)
            {
                // One sweep over the cells: cell j of the odd row is normalised (and its D state re-run from the raw chain), which
                // completes I of cell j - 1 and M of cell j of the even row; their MAP terms are taken at once, in the order
                // M0, I0, M1, I1, ... of the other variants, so the even row is never held as a whole.
                const bool row1 = i == 2;                   // row 1 is stored normalised and has no D state
                const double inv_o = row1 ? 1. : 1. / u_si, m2o = row1 ? 0. : p.m2, m8o = row1 ? 0. : p.m8;
                double pm = 0., pd = 0.;                    // the odd row's raw M and D at j - 1
                double sum = 0., max = 0.; int max_k = -1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int idx = i - 1 - BW - 1 + j;     // reference index of cell j of row i - 1
                    double fd = m2o * pm + m8o * pd;
                    fd = (idx < 0 || idx >= l_ref) ? 0. : fd;
                    pm = Mp[j]; pd = fd;
                    const double Mn = Mp[j] * inv_o, In = Ip[j] * inv_o, Dn = fd * inv_o;
                    Mp[j] = Mn; Ip[j] = In;
                    double z;
                    if (j > 0) {
                        const double fi = (EI * (p.m1 * Mn + p.m4 * In)) * inv_i;           // the forward kernel's I[i][j - 1]
                        z = fi * bIr[j - 1]; if (z > max) { max = z; max_k = (i - BW - 1 + j - 1) << 2 | 1; } sum += z;
                    }
                    const double e = emis_sel(FLD(rw, j), u_qyc, u_ematch, u_elo);
                    const double fm = (e * (p.m0 * Mn + p.m3 * In + p.m6 * Dn)) * inv_i;     // the forward kernel's M[i][j]
                    z = fm * bMr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 0; } sum += z;
                }
                sum += 0. * bIr[NB - 1];                    // I[i][NB - 1] = 0: the last term of the other variants' sum
                BAQ_MAP_FINISH(i, sum, max, max_k)
            }
            BAQ_BWD_APPLY(i - 1)
            {
                double sum = 0., max = 0.; int max_k = -1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double z;
                    z = Mp[j] * bMr[j]; if (z > max) { max = z; max_k = (i - 1 - BW - 1 + j) << 2 | 0; } sum += z;
                    z = Ip[j] * bIr[j]; if (z > max) { max = z; max_k = (i - 1 - BW - 1 + j) << 2 | 1; } sum += z;
                }
                BAQ_MAP_FINISH(i - 1, sum, max, max_k)
            }
        }
    } else if (DEC) {
        // pairs (i even, i - 1 odd): M of the even row + the full odd row are fetched before the even row's update
#pragma unroll 1
        for (; i >= 2; i -= 2) {
            const size_t t = (size_t)((i - 1) >> 1) * (3 * NB);
            double fM[NB], Mp[NB], Ip[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) fM[j] = __builtin_nontemporal_load(&F1[(t + 2 * NB + j) * 64]);
#pragma unroll
            for (int j = 0; j < NB; ++j) { baq_d2 v = __builtin_nontemporal_load(&F2[(t + 2 * j) * 32]); Mp[j] = v.x; Ip[j] = v.y; }
            BAQ_BWD_UPDATE(i)
            {
                double sum = 0., max = 0.; int max_k = -1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double z;
                    z = fM[j] * bMr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 0; } sum += z;
                    const double fi = (j + 1 < NB) ? (EI * (p.m1 * Mp[j + 1] + p.m4 * Ip[j + 1])) * inv_i : 0.;     // the forward kernel's I[i][j]
                    z = fi * bIr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 1; } sum += z;
                }
                BAQ_MAP_FINISH(i, sum, max, max_k)
            }
            BAQ_BWD_UPDATE(i - 1)
            {
                double sum = 0., max = 0.; int max_k = -1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double z;
                    z = Mp[j] * bMr[j]; if (z > max) { max = z; max_k = (i - 1 - BW - 1 + j) << 2 | 0; } sum += z;
                    z = Ip[j] * bIr[j]; if (z > max) { max = z; max_k = (i - 1 - BW - 1 + j) << 2 | 1; } sum += z;
                }
                BAQ_MAP_FINISH(i - 1, sum, max, max_k)
            }
        }
    }
#undef BAQ_BWD_UPDATE
#undef BAQ_BWD_INPUTS
#undef BAQ_BWD_APPLY
#undef BAQ_MAP_FINISH

    /*** realn.c, extended BAQ: bq = min(running max from the left, from the right) inside each M block; qual = min(qual, bq) ***/
    if (PLDS) {
        // the right-hand maximum was applied row by row above; the left-hand one needs the forward direction
        if (!plain) {
            int yy = 0;
            for (int c = 0; c < d.n_cigar; ++c) {
                int op = d.cigar[c] & 0xf, l = (int)(d.cigar[c] >> 4);
                if (cg_is_mop(op)) {
                    if (l > lq - yy) l = lq - yy;
                    int run = 0;
                    for (int q = yy; q < yy + l; ++q) {
                        const int b = Pl[(size_t)q * 64];
                        run = b > run ? b : run;
                        const int q0 = qual[q];
                        if (q0 > run) qual[q] = (uint8_t)run;
                    }
                    yy += l;
                } else if (op == CG_S || op == CG_I) {
                    if (l > lq - yy) l = lq - yy;
                    yy += l;
                }
            }
        }
    } else {
        long long xx = d.rpos; int yy = 0;
        for (int c = 0; c < d.n_cigar; ++c) {
            int op = d.cigar[c] & 0xf, l = (int)(d.cigar[c] >> 4);
            if (cg_is_mop(op)) {
                if (l > lq - yy) l = lq - yy;
                if (l > 0) {
                    int run = 0;
                    for (int q = yy; q < yy + l; ++q) {
                        int32_t pk = Pg[(size_t)q * 64];
                        int st = pk >> 8;
                        int b = ((st & 3) != 0 || (long long)(st >> 2) != xx - d.xb + (q - yy)) ? 0 : (pk & 0xff);
                        run = b > run ? b : run;
                        Pg[(size_t)q * 64] = (b << 8) | run;        // raw bq, left running max
                    }
                    run = 0;
                    for (int q = yy + l - 1; q >= yy; --q) {
                        int32_t pk = Pg[(size_t)q * 64];
                        int b = pk >> 8, left = pk & 0xff;
                        run = b > run ? b : run;
                        int bqv = plain ? b : (left < run ? left : run);    // plain (calmd -r without -E): min(qual, q) per base
                        int q0 = qual[q];
                        int tag = 64 + (q0 <= bqv ? 0 : q0 - bqv);
                        qual[q] = (uint8_t)(q0 - (tag - 64));
                    }
                }
                xx += l; yy += l;
            } else if (op == CG_S || op == CG_I) {
                if (l > lq - yy) l = lq - yy;
                yy += l;
            } else if (op == CG_D) xx += l;
        }
    }
}

template <int BW, int DEC, bool PLDS>
__global__ void __launch_bounds__(256, 2) k_baq_bwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                    double *scratch, size_t slot_dbl, int lq_cap, int lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t baq_state[];        // PLDS: [wave][row][lane], one byte per row and read
    BAQ_TABLES_INIT()
    baq_bwd_body<BW, DEC, PLDS>(R, W, q2p, refc, lt_ok ? lt_s : nullptr, baq_state, g0, ngroups, use_list, scratch, slot_dbl, lq_cap, lds_rows, (int64_t)blockIdx.x);
}

// Both passes of one group in one launch, for the reads that go through the list (a few dozen waves, latency bound): once its waves
// are placed they run to the end beside the persistent class-S kernel, instead of the second pass queueing behind it for a free slot.
// A wave's own forward rows are visible to its backward pass in program order; nothing is shared between waves.
template <int BW, int DEC, bool PLDS, bool STRIDED>
__global__ void __launch_bounds__(256, 2) k_baq_list(StaReadsDev R, StaWinDev W, BaqTables T, int64_t ngroups, double *scratch, size_t slot_dbl, int lq_cap, int lds_rows,
                                                     const int32_t *__restrict__ range /* [lo, hi): the list groups that hold this kernel's reads (k_baq_list_partition); NULL: all */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t baq_state[];
    // The launch holds at most BAQ_LIST_MAX_WG workgroups (sta_launch_baq_list): a workgroup walks the groups of four with the grid's stride.
    // With one workgroup per four groups the two list kernels of an indel-rich 16 M-column window (650 workgroups each, two fit a CU) took every
    // register file of the chip for their first 2.4 ms and the persistent class-S kernel -- and the gather in front of it -- waited for them.
    const int64_t nwg = (ngroups + 3) / 4;
    int64_t lo = 0, hi = nwg;
    if (range) { lo = range[0] / 4; hi = (range[1] + 3) / 4; if (hi > nwg) hi = nwg; }      // (the workgroups that hold this kernel's groups)
    if (lo + (int64_t)blockIdx.x >= hi) return;                                            // before the tables are loaded
    BAQ_TABLES_INIT()
    if (!STRIDED) {          // one workgroup per four groups (the loop costs this kernel 19 more spilled registers: its own instantiation)
        const int64_t vb = lo + (int64_t)blockIdx.x;
        baq_fwd_body<BW, DEC>(R, W, q2p, refc, 0, ngroups, 1, scratch, slot_dbl, lq_cap, vb);
        baq_bwd_body<BW, DEC, PLDS>(R, W, q2p, refc, lt_ok ? lt_s : nullptr, baq_state, 0, ngroups, 1, scratch, slot_dbl, lq_cap, lds_rows, vb);
        return;
    }
    for (int64_t vb = lo + (int64_t)blockIdx.x; vb < hi; vb += (int64_t)gridDim.x) {
        baq_fwd_body<BW, DEC>(R, W, q2p, refc, 0, ngroups, 1, scratch, slot_dbl, lq_cap, vb);
        baq_bwd_body<BW, DEC, PLDS>(R, W, q2p, refc, lt_ok ? lt_s : nullptr, baq_state, 0, ngroups, 1, scratch, slot_dbl, lq_cap, lds_rows, vb);
    }
}

// The list (R.chain[1 .. 1 + chain[0])) holds the reads of three kernels in the order k_prep_reads' blocks appended them: band width 7
// outside class S, band width 8, general band.  Every list kernel walks groups of 64 list entries and a lane sits idle where the entry
// is another kernel's -- on indel-rich input (mpileup30_indel: 42 000 list reads, half of them band width 8) both band kernels ran over
// the whole list at half their lanes: twice the waves beside the persistent class-S kernel.  This puts the list in class order (7 | 8 |
// general; the order inside a class does not matter: reads are independent) and says which groups hold which class, so that a list
// kernel's workgroups outside its range leave at once.  One workgroup: LDS counters and cursors, the permutation through `tmp`, then back.
__global__ void __launch_bounds__(1024) k_baq_list_partition(StaReadsDev R, int32_t *__restrict__ tmp)
{
    __shared__ int cnt[3], cur[3];
    const int t = threadIdx.x;
    if (t < 3) { cnt[t] = 0; cur[t] = 0; }
    __syncthreads();
    const int n = R.chain[0];
    auto cls = [&](int32_t r) { const uint32_t info = R.info[r]; const int bw = (info & RI_BAQ) ? (int)((info >> RI_BAQ_BW_SHIFT) & 31) : 0; return bw == 7 ? 0 : bw == 8 ? 1 : 2; };
    for (int i = t; i < n; i += 1024) atomicAdd(&cnt[cls(R.chain[1 + i])], 1);
    __syncthreads();
    const int n7 = cnt[0], n8 = cnt[1];
    for (int i = t; i < n; i += 1024) {
        const int32_t r = R.chain[1 + i];
        const int k = cls(r);
        tmp[(k == 0 ? 0 : k == 1 ? n7 : n7 + n8) + atomicAdd(&cur[k], 1)] = r;
    }
    __syncthreads();
    for (int i = t; i < n; i += 1024) R.chain[1 + i] = tmp[i];
    // the list groups (of 64 entries) that hold entries of band width 7 / 8: what k_baq_list<7 / 8> has to look at
    if (t == 0) { tmp[n] = 0; tmp[n + 1] = (n7 + 63) / 64; tmp[n + 2] = n7 / 64; tmp[n + 3] = (n7 + n8 + 63) / 64; }
}
// tmp: chain[0] + 4 words; the group ranges [lo7, hi7, lo8, hi8] are left in its last four
void sta_launch_baq_list_partition(hipStream_t s, const StaReadsDev &r, int32_t *tmp)
{
    hipLaunchKernelGGL(k_baq_list_partition, dim3(1), dim3(1024), 0, s, r, tmp);
}

template <int NB, int DEC>
static size_t baq_slot_dbl_t(int lq_cap)
{
    // forward rows + s[] + the per-row state ints of the fallback path (reads longer than BAQ_LDS_ROWS_MAX)
    return baq_rows_lr<NB, DEC>(lq_cap) * 64 + (size_t)(lq_cap + 2) * 64 + (lq_cap > BAQ_LDS_ROWS_MAX ? (size_t)(lq_cap + 1) / 2 * 64 : 0);
}
// band width 7 stores I rows every second row (DEC), band width 8 (a few reads with a 2-3 bp deletion: it would not fit the register
// file at two waves per SIMD) stores every row
// STA_BAQ_DEC=1 selects the previous layout (M every row + I every second row) for A/B measurements
static int baq_dec_mode() { static const int m = [] { const char *e = getenv("STA_BAQ_DEC"); return e && atoi(e) == 1 ? 1 : 2; }(); return m; }
static size_t baq_slot_dbl(int lq_cap, int bw) { return bw == 7 ? (baq_dec_mode() == 2 ? baq_slot_dbl_t<15, 2>(lq_cap) : baq_slot_dbl_t<15, 1>(lq_cap)) : baq_slot_dbl_t<17, 0>(lq_cap); }

// bytes of forward-row stream per query base of the band-7 kernel pair (written once, read once): what bench.py reports as DRAM traffic
extern "C" double sta_baq_stream_bytes_per_base(void) { return (baq_dec_mode() == 2 ? 2 : 3) * 15 * 8 / 2.0; }
// the same for the class-S kernel: one row of three stored, (M, I) of 15 cells, 16 bytes each -- written once and read once inside the launch
extern "C" double sta_baq7s_stream_bytes_per_base(void) { return 15 * 16 / 3.0; }

size_t sta_baq_band_scratch_bytes(int64_t n_reads, int lq_cap, int *groups_per_launch, int slab_gib_cap)
{
    int64_t ngroups = (n_reads + 63) / 64;
    // One launch per pass over ALL groups when the slab fits (no partially filled last round of waves);
    // 288 GB of HBM makes a tens-of-GiB slab affordable.  STA_BAQ_SLAB_GIB bounds it (default 48).
    int64_t gpl = ngroups;
    const char *ev = getenv("STA_BAQ_GROUPS");
    if (ev && atoi(ev) > 0) gpl = atoi(ev);
    size_t slab_gib = 48;
    const char *es = getenv("STA_BAQ_SLAB_GIB");
    if (es && atoi(es) > 0) slab_gib = (size_t)atoi(es);
    if (slab_gib_cap > 0 && (size_t)slab_gib_cap < slab_gib) slab_gib = (size_t)slab_gib_cap;     // the engine's fallback after a failed allocation
    size_t slot = std::max(baq_slot_dbl(lq_cap, 8), baq_slot_dbl(lq_cap, 7)) * 8;
    if ((size_t)gpl * slot > (slab_gib << 30)) {
        // split into equal chunks that are multiples of 6144 groups (LCM of the 3072 / 2048 resident waves of the two kernels)
        int64_t fit = (int64_t)((slab_gib << 30) / slot);
        int64_t nchunk = (gpl + fit - 1) / fit;
        gpl = (ngroups + nchunk - 1) / nchunk;
        if (gpl > 6144) gpl = (gpl + 6143) / 6144 * 6144;
        while (gpl > 4 && (size_t)gpl * slot > (slab_gib << 30)) gpl -= gpl > 6144 ? 6144 : gpl / 2;
    }
    if (gpl > ngroups) gpl = ngroups;
    if (gpl < 1) gpl = 1;
    if (groups_per_launch) *groups_per_launch = (int)gpl;
    return (size_t)gpl * slot;
}

template <int BW, int DEC>
static void run_band(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int64_t g0, int64_t ng, int use_list, int pass)
{
    const BaqTables g_tables = baq_tables();
    unsigned nb = (unsigned)((ng + 3) / 4);
    // the slot stride is the same for both band widths (the engine sizes one slab for either): the larger of the two layouts
    size_t slot = std::max(baq_slot_dbl(lq_cap, 8), baq_slot_dbl(lq_cap, 7));
    if (pass == 0) { hipLaunchKernelGGL((k_baq_fwd<BW, DEC>), dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, slot, lq_cap); return; }
    if (lq_cap <= BAQ_LDS_ROWS_MAX) {
        const int rows = (lq_cap + 3) & ~3;
        hipLaunchKernelGGL((k_baq_bwd<BW, DEC, true>), dim3(nb), dim3(256), (size_t)4 * rows * 64, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, slot, lq_cap, rows);
    } else
        hipLaunchKernelGGL((k_baq_bwd<BW, DEC, false>), dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, slot, lq_cap, 0);
}

// both passes of the list's band-width-bw reads (ng groups of 64 list entries) in one launch, blocks of one wave
void sta_launch_baq_list(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int bw, int64_t ng, const int32_t *range)
{
    if (r.n == 0 || lq_cap <= 0 || ng <= 0) return;
    const BaqTables g_tables = baq_tables();
    const size_t slot = std::max(baq_slot_dbl(lq_cap, 8), baq_slot_dbl(lq_cap, 7));
    static const int64_t max_wg = [] { const char *e = getenv("STA_BAQ_LIST_MAX_WG"); const long v = e ? atol(e) : 128; return (int64_t)(v < 1 ? 1 : v); }();
    // (a list of up to 256 workgroups is launched whole, as before: measured equal at 4 M columns; 650 workgroups per class at 16 M: 27.7 -> 27.0 ms)
    const int64_t nwg = (ng + 3) / 4;
    const unsigned nb = (unsigned)(nwg > 2 * max_wg || getenv("STA_BAQ_LIST_MAX_WG") ? std::min<int64_t>(nwg, max_wg) : nwg);      // (k_baq_list walks the rest with the grid's stride)
    const bool plds = lq_cap <= BAQ_LDS_ROWS_MAX;
    const int rows = plds ? (lq_cap + 3) & ~3 : 0;
    const size_t lds = plds ? (size_t)4 * rows * 64 : 0;
    const bool strided = (int64_t)nb < nwg;
#define BAQ_LIST_LAUNCH(BW_, DEC_, P_) do { if (strided) hipLaunchKernelGGL((k_baq_list<BW_, DEC_, P_, true>), dim3(nb), dim3(256), lds, s, r, w, g_tables, ng, (double *)scratch, slot, lq_cap, rows, range); \
                                            else hipLaunchKernelGGL((k_baq_list<BW_, DEC_, P_, false>), dim3(nb), dim3(256), lds, s, r, w, g_tables, ng, (double *)scratch, slot, lq_cap, rows, range); } while (0)
    if (bw == 7) {
        if (baq_dec_mode() == 2) { if (plds) BAQ_LIST_LAUNCH(7, 2, true); else BAQ_LIST_LAUNCH(7, 2, false); }
        else { if (plds) BAQ_LIST_LAUNCH(7, 1, true); else BAQ_LIST_LAUNCH(7, 1, false); }
    } else if (bw == 8) { if (plds) BAQ_LIST_LAUNCH(8, 0, true); else BAQ_LIST_LAUNCH(8, 0, false); }
#undef BAQ_LIST_LAUNCH
}

// One pass (0 = forward, 1 = backward + MAP + apply) over groups [g0, g0 + ng) of 64 reads; the engine calls the two passes
// chunk by chunk (a chunk = what fits the scratch slab) so that each launch can be timed on its own.
void sta_launch_baq_band(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int bw,
                         int64_t g0, int64_t ng, int use_list, int pass)
{
    if (r.n == 0 || lq_cap <= 0 || ng <= 0) return;
    if (bw == 7) { if (baq_dec_mode() == 2) run_band<7, 2>(s, r, w, scratch, lq_cap, g0, ng, use_list, pass); else run_band<7, 1>(s, r, w, scratch, lq_cap, g0, ng, use_list, pass); }
    else if (bw == 8) run_band<8, 0>(s, r, w, scratch, lq_cap, g0, ng, use_list, pass);
}

// ================================================================================================
// Class S (baq_band7s.h): one persistent, fused kernel.  A wave takes groups of 64 consecutive reads from a counter; for each it
// packs the per-row inputs, runs the forward pass into its scratch slot and, one group later, the backward pass out of it.  Half
// of the waves run one forward pass ahead of the other half (two slots per wave), so that at any time about half of a CU's waves
// are in the store-heavy forward phase and half in the issue-heavy backward phase: the forward rows' write stream, their read
// stream and the fp64 issue overlap inside one launch instead of alternating between two.

#ifndef BAQ7S_DEFAULT_MODE
#define BAQ7S_DEFAULT_MODE 16          // baq7s::M_LOGTAB
#endif
struct BaqHostTables { float q2p[256]; baq7s::LogTab lt; bool lt_ok; };
static const BaqHostTables &baq_host_tables()
{
    static const BaqHostTables t = [] {
        BaqHostTables h;
        for (int i = 0; i < 256; ++i) h.q2p[i] = (float)pow(10, -i / 10.);   // g_qual2prob (probaln.c), host libm
        h.lt_ok = baq7s::make_log_thresholds(h.lt);       // false: this host's log() is not a clean step function around a threshold -> the formula
        return h;
    }();
    return t;
}

static BaqTables baq_tables(bool *logtab_ok)
{
    const BaqHostTables &h = baq_host_tables();
    BaqTables T;
    memcpy(T.q2p, h.q2p, sizeof T.q2p);
    T.lt = nullptr;
    if (h.lt_ok && !getenv("STA_BAQ_NO_LOGTAB")) {
        static std::mutex m;
        static std::map<int, const double *> by_dev;       // device -> its copy of the threshold table (nullptr: no memory for it, the formula then)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
        std::lock_guard<std::mutex> g(m);
        auto it = by_dev.find(dev);
        if (it == by_dev.end()) {
            double *dp = nullptr;
            if (dev < 0 || hipMalloc((void **)&dp, sizeof h.lt.t) != hipSuccess || hipMemcpy(dp, h.lt.t, sizeof h.lt.t, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipGetLastError();
                if (dp) (void)hipFree(dp);
                dp = nullptr;
            }
            it = by_dev.emplace(dev, dp).first;
        }
        T.lt = it->second;
    }
    if (logtab_ok) *logtab_ok = T.lt != nullptr;
    return T;
}

__device__ __forceinline__ double baq_uni(double x)       // a wave-uniform double into scalar registers
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

struct Baq7sSlot { uint32_t *IN; baq7s::d2 *F2; double *S; };
__host__ __device__ inline size_t baq7s_in_bytes(int lq_cap) { return (((size_t)(lq_cap + 2) * 64 * 4) + 1023) & ~(size_t)1023; }
__host__ __device__ inline size_t baq7s_f2_bytes(int lq_cap) { return (size_t)((lq_cap + 2) / 3) * baq7s::NB * 64 * 16; }      // one row of three is stored
__host__ __device__ inline size_t baq7s_slot_bytes(int lq_cap) { return baq7s_in_bytes(lq_cap) + baq7s_f2_bytes(lq_cap) + (size_t)(lq_cap + 2) * 64 * 8; }
__device__ __forceinline__ Baq7sSlot baq7s_slot(uint8_t *scratch, size_t slot_bytes, int64_t idx, int lq_cap, int lane)
{
    uint8_t *b = scratch + (size_t)idx * slot_bytes;
    Baq7sSlot s;
    (void)lane;             // the slot pointers stay wave-uniform: the lane goes into the 32-bit element index (baq7s::at)
    s.IN = reinterpret_cast<uint32_t *>(b);
    s.F2 = reinterpret_cast<baq7s::d2 *>(b + baq7s_in_bytes(lq_cap));
    s.S = reinterpret_cast<double *>(b + baq7s_in_bytes(lq_cap) + baq7s_f2_bytes(lq_cap));
    return s;
}

struct Baq7sRead { bool active; int lq; baq7s::Shape sh; uint8_t *qual; const uint8_t *seq; };
__device__ __forceinline__ Baq7sRead baq7s_read(const StaReadsDev &R, const StaWinDev &W, int64_t g, int lane)
{
    Baq7sRead d; d.active = false; d.lq = 0; d.qual = nullptr; d.seq = nullptr; d.sh.ok = false; d.sh.ys = d.sh.mlen = 0; d.sh.xb = 0;
    // group g = entries [64 g, 64 g + 64) of the per-length list (one read length per group; -1 = padding behind a length's last read)
    const int64_t r = R.s_ws[STA_SLIST_HEAD + g * 64 + lane];
    if (r >= 0) {
        d.active = true;
        d.lq = R.l_qseq[r];
        const uint32_t c0 = R.cig_off[r];
        d.sh = baq7s::classify(R.cigar + c0, (int)(R.cig_off[r + 1] - c0), W.origin + R.pos[r], d.lq, W.ref_len);
        const uint64_t boff = (uint64_t)R.base_off8[r] << 3;
        d.qual = R.qual + boff; d.seq = R.seq + (boff >> 1);
    }
    return d;
}

// the class-S candidates into their list: position = the length's first position (host) + the block's share of the length's cursor + the
// read's place inside the block.  One DEVICE atomic per block and distinct length (one per wave was 13 000 atomics on the one word of the
// common length: 0.15 ms); a wave's reads keep their order.
__global__ void __launch_bounds__(1024) k_baq7s_gather(StaReadsDev R)
{
    __shared__ int s_cnt[STA_SLIST_BINS], s_gb[STA_SLIST_BINS];
    for (int k = threadIdx.x; k < STA_SLIST_BINS; k += blockDim.x) s_cnt[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool s = i < R.n && (R.info[i] & RI_BAQ_S);
    const int lq = s ? R.l_qseq[i] : -1;
    int my_off = 0;
    unsigned long long todo = __ballot(s);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int l0 = __shfl(lq, src);
        const unsigned long long same = __ballot(s && lq == l0);
        int at = 0;
        if (lane == src) at = atomicAdd(&s_cnt[l0], (int)__popcll(same));
        at = __shfl(at, src);
        if (s && lq == l0) my_off = at + (int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    __syncthreads();
    int32_t *cur = R.s_ws + 2 * STA_SLIST_BINS;
    for (int k = threadIdx.x; k < STA_SLIST_BINS; k += blockDim.x) if (s_cnt[k]) s_gb[k] = atomicAdd(&cur[k], s_cnt[k]);
    __syncthreads();
    if (s) R.s_ws[STA_SLIST_HEAD + R.s_ws[STA_SLIST_BINS + lq] + s_gb[lq] + my_off] = (int32_t)i;
}

void sta_launch_baq7s_gather(hipStream_t s, const StaReadsDev &r)
{
    if (r.n <= 0 || !r.s_ws) return;
    hipLaunchKernelGGL(k_baq7s_gather, dim3((unsigned)((r.n + 1023) / 1024)), dim3(1024), 0, s, r);
}

// two waves per SIMD: the backward pass holds 120 doubles of band state (256 VGPRs; 168 would spill 230 of them)
template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_baq7s(StaReadsDev R, StaWinDev W, BaqTables T, int64_t ngroups, unsigned *next,
                                              uint8_t *scratch, size_t slot_bytes, int lq_cap, int lead_mask)
{
    __shared__ __attribute__((aligned(16))) baq7s::d2 mid_row[baq7s::NB * 64];   // [cell][lane]: the normalised middle row of the group in work (backward pass)
    __shared__ float q2p[256];
    __shared__ uint8_t refc[256];
    __shared__ double lt[(MODE & baq7s::M_LOGTAB) ? baq7s::LT_N : 2];
    const int lane = threadIdx.x;
    for (int k = lane; k < 256; k += 64) { q2p[k] = T.q2p[k]; refc[k] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)k)); }
    if (MODE & baq7s::M_LOGTAB) for (int k = lane; k < baq7s::LT_N; k += 64) lt[k] = T.lt[k];
    __syncthreads();
    // volatile, and in the LDS address space: these stores and loads must BE ds_write_b128 / ds_read_b128 (forwarded through registers
    // they would cost 60 VGPRs; through a generic pointer they become flat accesses)
    typedef __attribute__((address_space(3))) volatile baq7s::d2 lds_d2;
    lds_d2 *Ln = (lds_d2 *)mid_row + lane;
    int64_t pend0 = -1, pend1 = -1;                   // groups whose forward rows wait in a slot (slot index in bit 62), oldest first
    int np = 0, cur = 0;
    bool extra = (blockIdx.x & lead_mask) != 0;         // this wave runs one forward pass ahead
    for (;;) {
        unsigned gt = 0;
        if (lane == 0) gt = atomicAdd(next, 1u);
        const int64_t g = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)gt);
        const bool have = g < ngroups;
        if (have) {
            const Baq7sRead d = baq7s_read(R, W, g, lane);
            const unsigned long long act = __ballot(d.active);
            if (act) {
                const int lq = __builtin_amdgcn_readfirstlane(__shfl(d.lq, __ffsll((long long)act) - 1));
                const Baq7sSlot sl = baq7s_slot(scratch, slot_bytes, (int64_t)blockIdx.x * 2 + cur, lq_cap, lane);
                bool all_edge = false;
                if (d.active) {
                    baq7s::Par p = baq7s::make_par(lq, lq + 6);
                    p.m0 = baq_uni(p.m0); p.m1 = baq_uni(p.m1); p.m2 = baq_uni(p.m2); p.m3 = baq_uni(p.m3); p.m4 = baq_uni(p.m4);
                    p.m6 = baq_uni(p.m6); p.m8 = baq_uni(p.m8); p.sM = baq_uni(p.sM); p.sI = baq_uni(p.sI); p.bM = baq_uni(p.bM); p.bI = baq_uni(p.bI);
                    const bool amb = baq7s::pack_lane<64>(lq, lq + 6, d.qual, d.seq, W.ref + d.sh.xb, refc, sl.IN, lane);
                    all_edge = baq7s::wave_any(amb);
                    baq7s::fwd_lane<64, MODE>(p, lq, all_edge, sl.IN, sl.F2, sl.S, lane, q2p);
                }
                all_edge = __ballot(all_edge) != 0;
                const int64_t tag = g | ((int64_t)cur << 62) | ((int64_t)(all_edge ? 1 : 0) << 61);
                if (np == 0) pend0 = tag; else pend1 = tag;
                ++np;
                cur ^= 1;
            }
        }
        if (extra && have) { extra = false; continue; }
        if (np > 0) {
            const int64_t pg = pend0;
            pend0 = pend1; --np;
            const int64_t g2 = pg & (((int64_t)1 << 61) - 1);
            const int sidx = (int)(pg >> 62) & 1;
            const bool all_edge = ((pg >> 61) & 1) != 0;
            const Baq7sRead d = baq7s_read(R, W, g2, lane);
            const unsigned long long act = __ballot(d.active);
            const int lq = __builtin_amdgcn_readfirstlane(__shfl(d.lq, __ffsll((long long)act) - 1));
            const Baq7sSlot sl = baq7s_slot(scratch, slot_bytes, (int64_t)blockIdx.x * 2 + sidx, lq_cap, lane);
            if (d.active) {
                baq7s::Par p = baq7s::make_par(lq, lq + 6);
                p.m0 = baq_uni(p.m0); p.m1 = baq_uni(p.m1); p.m2 = baq_uni(p.m2); p.m3 = baq_uni(p.m3); p.m4 = baq_uni(p.m4);
                p.m6 = baq_uni(p.m6); p.m8 = baq_uni(p.m8); p.sM = baq_uni(p.sM); p.sI = baq_uni(p.sI);
                p.eim1 = baq_uni(p.eim1); p.eim4 = baq_uni(p.eim4);
                baq7s::BwdCtx c; c.ys = d.sh.ys; c.mlen = d.sh.mlen; c.run_r = 0; c.plain_mask = W.baq_plain != 0 ? -1 : 0;
                c.LT = (baq7s::LtPtr)lt;
                baq7s::bwd_lane<64, MODE>(p, lq, lq + 6, all_edge, sl.IN, sl.F2, sl.S, lane, q2p, Ln, c);
                baq7s::final_lane<64>(lq, sl.IN, lane, c, d.qual);
            }
        } else if (!have) break;
    }
}

static int g_baq7s_waves = 0;
// scratch for the class-S kernel: a 256-byte header (the group counter) + two slots per resident wave
size_t sta_baq7s_scratch_bytes(int lq_cap, int64_t ngroups, int *waves_out)
{
    if (!g_baq7s_waves) {
        int dev = 0, cus = 256, per = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_baq7s<0>, 64, 0) != hipSuccess || per < 1) { (void)hipGetLastError(); per = 8; }
        const char *e = getenv("STA_BAQ7S_WAVES_PER_CU");
        if (e && atoi(e) > 0 && atoi(e) < per) per = atoi(e);
        g_baq7s_waves = cus * per;
    }
    int64_t waves = g_baq7s_waves;
    if (waves > ngroups) waves = ngroups;
    if (waves < 1) waves = 1;
    if (waves_out) *waves_out = (int)waves;
    return 256 + (size_t)waves * 2 * baq7s_slot_bytes(lq_cap);
}

void sta_launch_baq7s(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int waves, int64_t ngroups)
{
    if (ngroups <= 0 || waves <= 0 || !r.s_ws) return;
    bool g_logtab_ok = false;
    const BaqTables g_tables = baq_tables(&g_logtab_ok);       // (an M_LOGTAB instantiation is never launched without this device's table)
    static const int lead = [] { const char *e = getenv("STA_BAQ7S_LEAD"); return e ? atoi(e) : 1; }();   // 0: every wave forward-then-backward; 1: odd waves one forward pass ahead
    hipMemsetAsync(scratch, 0, 256, s);
    // STA_BAQ7S_MODE: 0 non-temporal row stream (default), 1 plain loads / stores; 2, 3: diagnostics with wrong results (baq_band7s.h)
    // (read per launch, not once per process: the parity tests switch builds inside one process)
    int mode = BAQ7S_DEFAULT_MODE;
    if (const char *e = getenv("STA_BAQ7S_MODE")) mode = atoi(e);
    if (!g_logtab_ok) mode &= ~baq7s::M_LOGTAB;
#define BAQ7S_LAUNCH(M) hipLaunchKernelGGL(k_baq7s<M>, dim3((unsigned)waves), dim3(64), 0, s, r, w, g_tables, ngroups, (unsigned *)scratch, \
                                           (uint8_t *)scratch + 256, baq7s_slot_bytes(lq_cap), lq_cap, lead)
    switch (mode) {
    case 0: BAQ7S_LAUNCH(0); break;
    case 1: BAQ7S_LAUNCH(1); break;
    case 2: BAQ7S_LAUNCH(2); break;
    case 3: BAQ7S_LAUNCH(3); break;
    case 16: BAQ7S_LAUNCH(16); break;          // M_LOGTAB (default)
    case 17: BAQ7S_LAUNCH(17); break;
    default: fprintf(stderr, "samtools-amd: STA_BAQ7S_MODE=%d is not built\n", mode); abort();
    }
#undef BAQ7S_LAUNCH
}
