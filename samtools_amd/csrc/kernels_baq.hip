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

#define EI .25
#define EM .33333333333

struct BaqTables { float q2p[256]; };

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
        double v = -4.343 * log(1. - max) + .499;
        // (int)v with x86 cvttsd2si semantics for out-of-range / NaN (the CPU reference's behaviour)
        int kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;
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

static BaqTables g_tables;
static bool g_tables_init = false;

void sta_launch_baq(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int redo, void *scratch, size_t scratch_bytes,
                         int lq_max, int bw_max, int64_t n_slow)
{
    if (!g_tables_init) {
        for (int i = 0; i < 256; ++i) g_tables.q2p[i] = (float)pow(10, -i / 10.);   // g_qual2prob (probaln.c), host libm
        g_tables_init = true;
    }
    if (r.n == 0 || lq_max <= 0 || n_slow <= 0) return;
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

template <int BW>
__global__ void __launch_bounds__(256) k_baq_fwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                 double *scratch, size_t slot_dbl, int lq_cap)
{
    constexpr int NB = 2 * BW + 1;
    __shared__ float q2p[256];
    __shared__ uint8_t refc[256];
    q2p[threadIdx.x] = T.q2p[threadIdx.x];
    refc[threadIdx.x] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)threadIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t gl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // group within this launch == scratch slot
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    double *slot = scratch + (size_t)gl * slot_dbl;
    baq_d2 *F = reinterpret_cast<baq_d2 *>(slot) + lane;
    double *S = slot + (size_t)lq_cap * (2 * NB) * 64 + lane;
#define Fcell(i, j) F[((size_t)((i) - 1) * NB + (j)) * 64]
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
        for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &Fcell(1, j)); }
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
        double inv = 1. / sum;
#pragma unroll
        for (int j = 0; j < NB; ++j) { M[j] *= inv; I[j] *= inv; D[j] *= inv; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { baq_d2 v = { M[j], I[j] }; __builtin_nontemporal_store(v, &Fcell(i, j)); }
    }
    {   // s[l_query+1]
        double sum = 0.;
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += M[j] * p.sM + I[j] * p.sI;
        S[(size_t)(lq + 1) * 64] = sum;
    }
#undef Fcell
}

#ifndef BAQ_PREFETCH_F
#define BAQ_PREFETCH_F 1
#endif
template <int BW>
__global__ void __launch_bounds__(256, BAQ_PREFETCH_F ? 2 : 3) k_baq_bwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                 double *scratch, size_t slot_dbl, int lq_cap)
{
    constexpr int NB = 2 * BW + 1;
    __shared__ float q2p[256];
    __shared__ uint8_t refc[256];
    q2p[threadIdx.x] = T.q2p[threadIdx.x];
    refc[threadIdx.x] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)threadIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t gl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    double *slot = scratch + (size_t)gl * slot_dbl;
    const baq_d2 *F = reinterpret_cast<const baq_d2 *>(slot) + lane;
    const double *S = slot + (size_t)lq_cap * (2 * NB) * 64 + lane;
    int32_t *P = reinterpret_cast<int32_t *>(slot + (size_t)lq_cap * (2 * NB) * 64 + (size_t)(lq_cap + 2) * 64) + lane;
#define Fcell(i, j) F[((size_t)((i) - 1) * NB + (j)) * 64]
    const BaqRd d = baq_rd(R, W, r);
    const int lq = d.lq, l_ref = d.l_ref;
    uint8_t *qual = d.qual; const uint8_t *seq = d.seq; const char *ref = d.ref;
    const BaqPar p = baq_par(lq, l_ref);

    // row lq: field j = code(lq - BW - 1 + j); it doubles as the backward word of row lq-1
    uint64_t rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) rw |= (uint64_t)RCODE(lq - BW - 1 + j) << (3 * j);
    double bMr[NB], bIr[NB];
    {
        double sl = S[(size_t)lq * 64], sl1 = S[(size_t)(lq + 1) * 64];
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
#pragma unroll 1
    for (int i = lq; i >= 1; --i) {
        // forward row i for the MAP step.  BAQ_PREFETCH_F: issue the loads first so they fly during the row update
        // (costs 4*NB VGPRs: 2 waves/SIMD); otherwise load at the MAP step and rely on 3 waves/SIMD to hide the latency.
        double fM[NB], fI[NB];
#if BAQ_PREFETCH_F
#pragma unroll
        for (int j = 0; j < NB; ++j) { baq_d2 v = __builtin_nontemporal_load(&Fcell(i, j)); fM[j] = v.x; fI[j] = v.y; }
#endif
        if (i < lq) {
            const float qf = c_qf; const int sb = c_sb; const int nrc = c_rc; const double si = c_s;
            c_qf = q2p[r_q]; c_sb = r_sb; c_rc = RCONV(r_rr); c_s = r_s;
            if (i - 2 >= 1) { int i2 = i - 2; r_q = qual[i2]; r_sb = SEQB(i2); r_s = S[(size_t)i2 * 64]; r_rr = RRAW(i2 - BW); }
            if (i < lq - 1) rw = (rw << 3) | (uint64_t)nrc;
            const int qy = QCONV(sb, i);
            const double qli1 = qf;
            const double ematch = 1. - qli1, e_lo = qy > 3 ? 1. : qli1 * EM;
            const int qyc = qy > 3 ? 9 : qy;
            const double yv = i > 1 ? 1. : 0.;
            double dnext = 0.;
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                int rc = FLD(rw, j);
                double e = emis_sel(rc, qyc, ematch, e_lo) * bMr[j];     // rc == 7 <=> k >= l_ref: e = 0 * b
                double bi1 = j > 0 ? bIr[j - 1] : 0.;
                double bm = e * p.m0 + p.eim1 * bi1 + p.m2 * dnext;
                double bi_ = e * p.m3 + p.eim4 * bi1;
                double bd = (e * p.m6 + p.m8 * dnext) * yv;
                bMr[j] = bm; bIr[j] = bi_;
                dnext = bd;
            }
            double ys = 1. / si;
#pragma unroll
            for (int j = 0; j < NB; ++j) { bMr[j] *= ys; bIr[j] *= ys; }
            if (i <= BW) {            // cells with k < 1 do not exist in the reference: keep them at zero
#pragma unroll
                for (int j = 0; j < BW; ++j) if (j < BW + 1 - i) { bMr[j] = 0.; bIr[j] = 0.; }
            }
        }
        // MAP for row i
#if !BAQ_PREFETCH_F
#pragma unroll
        for (int j = 0; j < NB; ++j) { baq_d2 v = __builtin_nontemporal_load(&Fcell(i, j)); fM[j] = v.x; fI[j] = v.y; }
#endif
        double sum = 0., max = 0.;
        int max_k = -1;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double z;
            z = fM[j] * bMr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 0; } sum += z;
            z = fI[j] * bIr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 1; } sum += z;
        }
        max /= sum;
        double v = -4.343 * log(1. - max) + .499;
        int kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;
        P[(size_t)(i - 1) * 64] = (int32_t)(((uint32_t)max_k << 8) | (uint32_t)(uint8_t)(kq > 100 ? 99 : kq));
    }

    /*** realn.c, extended BAQ: bq = min(running max from the left, from the right) inside each M block; apply ***/
    {
        long long xx = d.rpos; int yy = 0;
        for (int c = 0; c < d.n_cigar; ++c) {
            int op = d.cigar[c] & 0xf, l = (int)(d.cigar[c] >> 4);
            if (cg_is_mop(op)) {
                if (l > lq - yy) l = lq - yy;
                if (l > 0) {
                    int run = 0;
                    for (int i = yy; i < yy + l; ++i) {
                        int32_t pk = P[(size_t)i * 64];
                        int st = pk >> 8;
                        int b = ((st & 3) != 0 || (long long)(st >> 2) != xx - d.xb + (i - yy)) ? 0 : (pk & 0xff);
                        run = b > run ? b : run;
                        P[(size_t)i * 64] = (b << 8) | run;        // raw bq, left running max
                    }
                    run = 0;
                    for (int i = yy + l - 1; i >= yy; --i) {
                        int32_t pk = P[(size_t)i * 64];
                        int b = pk >> 8, left = pk & 0xff;
                        run = b > run ? b : run;
                        int bqv = W.baq_plain ? b : (left < run ? left : run);    // plain (calmd -r without -E): min(qual, q) per base
                        int q0 = qual[i];
                        int tag = 64 + (q0 <= bqv ? 0 : q0 - bqv);
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
#undef Fcell
}

// ================================================================================================
// Checkpointed variant (opt-in with STA_BAQ_CHECKPOINT=1; measured slower on MI355X): the forward kernel stores only every C-th row (M, I and D: the
// full state), and the backward kernel re-computes the C rows of a block from the checkpoint below
// it, keeping their (M, I) cells in REGISTERS (one wave per SIMD, up to 512 VGPRs) while it walks the
// block backwards.  HBM stream per query base: 3*NB*8/C bytes written + read (90 B at BW 7, C 4)
// instead of 2*NB*8 (240 B) -- the fp64 work grows by one extra forward pass, which the kernel pair
// has headroom for (it was bound by the scratch stream).  Same arithmetic, same order, same results.
//
// Block word: reference codes for indices a-BW .. a+C+BW-1 of a block of rows a+1..a+C in ONE 64-bit
// word (3 bits per code); the forward word of row a+1+c is (word >> 3c), the backward word of that
// row is (word >> 3(c+1)), and the next lower block's word is (word << 3C) | C new codes.

template <int BW>
__device__ __forceinline__ double baq_row1(double (&M)[2 * BW + 1], double (&I)[2 * BW + 1], double (&D)[2 * BW + 1], uint64_t rw, int qy, double q0, const BaqPar &p)
{
    constexpr int NB = 2 * BW + 1;
    const double ematch = 1. - q0, e_lo = qy > 3 ? 1. : q0 * EM;
    const int qyc = qy > 3 ? 9 : qy;
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
#pragma unroll
    for (int j = 0; j < NB; ++j) { M[j] /= sum; I[j] /= sum; }
    return sum;
}

template <int BW>
__device__ __forceinline__ double baq_fwd_row(double (&M)[2 * BW + 1], double (&I)[2 * BW + 1], double (&D)[2 * BW + 1], uint64_t rw, int qy, double qli, const BaqPar &p)
{
    constexpr int NB = 2 * BW + 1;
    const double ematch = 1. - qli, e_lo = qy > 3 ? 1. : qli * EM;
    const int qyc = qy > 3 ? 9 : qy;
    double sum = 0., pm = 0., pd = 0.;
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
    const double inv = 1. / sum;
#pragma unroll
    for (int j = 0; j < NB; ++j) { M[j] *= inv; I[j] *= inv; D[j] *= inv; }
    return sum;
}

template <int BW, int C>
__global__ void __launch_bounds__(256) k_baq_ck_fwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                    double *scratch, size_t slot_dbl, int lq_cap)
{
    constexpr int NB = 2 * BW + 1;
    __shared__ float q2p[256];
    __shared__ uint8_t refc[256];
    q2p[threadIdx.x] = T.q2p[threadIdx.x];
    refc[threadIdx.x] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)threadIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t gl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    double *CK = scratch + (size_t)gl * slot_dbl + lane;          // CK[((t-1)*3NB + cell)*64], checkpoint after row t*C
    const BaqRd d = baq_rd(R, W, r);
    const int lq = d.lq, l_ref = d.l_ref;
    const uint8_t *qual = d.qual, *seq = d.seq; const char *ref = d.ref;
    const BaqPar p = baq_par(lq, l_ref);

    double M[NB], I[NB], D[NB];
    uint64_t rw = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) rw |= (uint64_t)RCODE(j - BW) << (3 * j);
    baq_row1<BW>(M, I, D, rw, QCONV(SEQB(0), 0), (double)q2p[qual[0]], p);
    int c_sb = 0, c_rc = 7; float c_qf = 0.f;
    int r_q = 0, r_sb = 0, r_rr = 256;
    if (lq >= 2) { c_qf = q2p[qual[1]]; c_sb = SEQB(1); c_rc = RCODE(2 + BW - 1); }
    if (lq >= 3) { r_q = qual[2]; r_sb = SEQB(2); r_rr = RRAW(3 + BW - 1); }
#pragma unroll 1
    for (int i = 2; i <= lq; ++i) {
        const float qf = c_qf; const int sb = c_sb; const int nrc = c_rc;
        c_qf = q2p[r_q]; c_sb = r_sb; c_rc = RCONV(r_rr);
        if (i + 2 <= lq) { r_q = qual[i + 1]; r_sb = SEQB(i + 1); r_rr = RRAW(i + 2 + BW - 1); }
        rw = (rw >> 3) | ((uint64_t)nrc << (3 * (NB - 1)));
        baq_fwd_row<BW>(M, I, D, rw, QCONV(sb, i - 1), (double)qf, p);
        if (i % C == 0 && i < lq) {
            double *ck = CK + (size_t)(i / C - 1) * (3 * NB) * 64;
#pragma unroll
            for (int j = 0; j < NB; ++j) { ck[(size_t)(3 * j) * 64] = M[j]; ck[(size_t)(3 * j + 1) * 64] = I[j]; ck[(size_t)(3 * j + 2) * 64] = D[j]; }
        }
    }
}

template <int BW, int C>
__global__ void __launch_bounds__(256) k_baq_ck_bwd(StaReadsDev R, StaWinDev W, BaqTables T, int64_t g0, int64_t ngroups, int use_list,
                                                    double *scratch, size_t slot_dbl, int lq_cap)
{
    constexpr int NB = 2 * BW + 1;
    constexpr int NF = NB + C;                 // fields of a block word
    __shared__ float q2p[256];
    __shared__ uint8_t refc[256];
    q2p[threadIdx.x] = T.q2p[threadIdx.x];
    refc[threadIdx.x] = (uint8_t)nt16_int_dev(nt16_from_char((unsigned char)threadIdx.x));
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t gl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gl >= ngroups) return;
    const int64_t r = baq_pick(R, g0 + gl, lane, use_list, BW);
    if (r < 0) return;
    const double *CK = scratch + (size_t)gl * slot_dbl + lane;
    const int nck_cap = (lq_cap - 1) / C;
    int32_t *P = reinterpret_cast<int32_t *>(scratch + (size_t)gl * slot_dbl + (size_t)nck_cap * (3 * NB) * 64) + lane;
    const BaqRd d = baq_rd(R, W, r);
    const int lq = d.lq, l_ref = d.l_ref;
    uint8_t *qual = d.qual; const uint8_t *seq = d.seq; const char *ref = d.ref;
    const BaqPar p = baq_par(lq, l_ref);

    double M[NB], I[NB], D[NB];                // forward state while a block is re-computed
    double sM_[C][NB], sI_[C][NB], Sb[C];      // the block's forward rows (M, I) and their scaling sums
    double bMr[NB], bIr[NB];                   // backward row
    int t = (lq - 1) / C;                      // top block: rows a+1 .. lq, a = t*C
    int a = t * C;
    // block word of the top block
    uint64_t bw_ = 0;
#pragma unroll
    for (int g = 0; g < NF; ++g) bw_ |= (uint64_t)RCODE(a - BW + g) << (3 * g);
    // inputs of the top block's rows (row a+1+c: quality / base of query index a+c), raw
    int rq[C], rsb[C], rnew[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { rq[c] = 0; rsb[c] = 0; rnew[c] = 256; if (a + c < lq) { rq[c] = qual[a + c]; rsb[c] = SEQB(a + c); } }
    if (t >= 1) {
        const double *ck = CK + (size_t)(t - 1) * (3 * NB) * 64;
#pragma unroll
        for (int j = 0; j < NB; ++j) { M[j] = ck[(size_t)(3 * j) * 64]; I[j] = ck[(size_t)(3 * j + 1) * 64]; D[j] = ck[(size_t)(3 * j + 2) * 64]; }
    }
    float carry_qf = 0.f; int carry_qy = 4;    // inputs of the first row of the block above (row a+C+1): backward row a+C needs them
#pragma unroll 1
    for (; t >= 0; --t) {
        a = t * C;
        // convert this block's inputs
        float inq[C]; int inqy[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { inq[c] = q2p[rq[c]]; inqy[c] = QCONV(rsb[c], a + c); }
        /*** re-compute rows a+1 .. a+C ***/
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int row = a + 1 + c;
            if (row <= lq) {
                const uint64_t rw = bw_ >> (3 * c);
                double sum;
                if (c == 0 && a == 0) sum = baq_row1<BW>(M, I, D, rw, inqy[0], (double)inq[0], p);
                else sum = baq_fwd_row<BW>(M, I, D, rw, inqy[c], (double)inq[c], p);
                Sb[c] = sum;
#pragma unroll
                for (int j = 0; j < NB; ++j) { sM_[c][j] = M[j]; sI_[c][j] = I[j]; }
            }
        }
        if (a + C >= lq) {
            // top block: s[lq+1] from row lq (still in M, I), then the backward row of lq
            double sum = 0.;
#pragma unroll
            for (int j = 0; j < NB; ++j) sum += M[j] * p.sM + I[j] * p.sI;
            const int ctop = lq - a - 1;
            double sl = Sb[0];
#pragma unroll
            for (int c = 1; c < C; ++c) if (c == ctop) sl = Sb[c];
            const double vM = p.sM / sl / sum, vI = p.sI / sl / sum;
            const uint64_t rwl = bw_ >> (3 * ctop);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bool valid = FLD(rwl, j) != 7;
                bMr[j] = valid ? vM : 0.; bIr[j] = valid ? vI : 0.;
            }
        }
        /*** fetch for the block below while this block walks backwards: its checkpoint into M/I/D, its raw inputs ***/
        const float first_qf = inq[0]; const int first_qy = inqy[0];
        if (t >= 1) {
            const int a2 = a - C;
            if (t >= 2) {
                const double *ck = CK + (size_t)(t - 2) * (3 * NB) * 64;
#pragma unroll
                for (int j = 0; j < NB; ++j) { M[j] = ck[(size_t)(3 * j) * 64]; I[j] = ck[(size_t)(3 * j + 1) * 64]; D[j] = ck[(size_t)(3 * j + 2) * 64]; }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) { rq[c] = qual[a2 + c]; rsb[c] = SEQB(a2 + c); rnew[c] = RRAW(a2 - BW + c); }
        }
        /*** backward rows a+C .. a+1 with MAP ***/
#pragma unroll
        for (int c = C - 1; c >= 0; --c) {
            const int i = a + 1 + c;
            if (i <= lq) {
                if (i < lq) {
                    const float qf = (c + 1 < C) ? inq[c + 1 < C ? c + 1 : 0] : carry_qf;
                    const int qy = (c + 1 < C) ? inqy[c + 1 < C ? c + 1 : 0] : carry_qy;
                    const uint64_t rw = bw_ >> (3 * (c + 1));
                    const double qli1 = qf;
                    const double ematch = 1. - qli1, e_lo = qy > 3 ? 1. : qli1 * EM;
                    const int qyc = qy > 3 ? 9 : qy;
                    const double yv = i > 1 ? 1. : 0.;
                    double dnext = 0.;
#pragma unroll
                    for (int j = NB - 1; j >= 0; --j) {
                        int rc = FLD(rw, j);
                        double e = emis_sel(rc, qyc, ematch, e_lo) * bMr[j];
                        double bi1 = j > 0 ? bIr[j - 1] : 0.;
                        double bm = e * p.m0 + p.eim1 * bi1 + p.m2 * dnext;
                        double bi_ = e * p.m3 + p.eim4 * bi1;
                        double bd = (e * p.m6 + p.m8 * dnext) * yv;
                        bMr[j] = bm; bIr[j] = bi_;
                        dnext = bd;
                    }
                    const double ys = 1. / Sb[c];
#pragma unroll
                    for (int j = 0; j < NB; ++j) { bMr[j] *= ys; bIr[j] *= ys; }
                    if (i <= BW) {
#pragma unroll
                        for (int j = 0; j < BW; ++j) if (j < BW + 1 - i) { bMr[j] = 0.; bIr[j] = 0.; }
                    }
                }
                double sum = 0., max = 0.;
                int max_k = -1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    double z;
                    z = sM_[c][j] * bMr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 0; } sum += z;
                    z = sI_[c][j] * bIr[j]; if (z > max) { max = z; max_k = (i - BW - 1 + j) << 2 | 1; } sum += z;
                }
                max /= sum;
                double v = -4.343 * log(1. - max) + .499;
                int kq = (v >= 2147483648.0 || v < -2147483648.0 || v != v) ? INT32_MIN : (int)v;
                P[(size_t)(i - 1) * 64] = (int32_t)(((uint32_t)max_k << 8) | (uint32_t)(uint8_t)(kq > 100 ? 99 : kq));
            }
        }
        carry_qf = first_qf; carry_qy = first_qy;
        // block word of the block below: shift up by C fields, C new codes at the bottom
        if (t >= 1) {
            uint64_t nw = bw_ << (3 * C);
#pragma unroll
            for (int c = 0; c < C; ++c) nw |= (uint64_t)RCONV(rnew[c]) << (3 * c);
            bw_ = nw;
        }
    }

    /*** realn.c, extended BAQ ***/
    {
        long long xx = d.rpos; int yy = 0;
        for (int c = 0; c < d.n_cigar; ++c) {
            int op = d.cigar[c] & 0xf, l = (int)(d.cigar[c] >> 4);
            if (cg_is_mop(op)) {
                if (l > lq - yy) l = lq - yy;
                if (l > 0) {
                    int run = 0;
                    for (int i = yy; i < yy + l; ++i) {
                        int32_t pk = P[(size_t)i * 64];
                        int st = pk >> 8;
                        int b = ((st & 3) != 0 || (long long)(st >> 2) != xx - d.xb + (i - yy)) ? 0 : (pk & 0xff);
                        run = b > run ? b : run;
                        P[(size_t)i * 64] = (b << 8) | run;
                    }
                    run = 0;
                    for (int i = yy + l - 1; i >= yy; --i) {
                        int32_t pk = P[(size_t)i * 64];
                        int b = pk >> 8, left = pk & 0xff;
                        run = b > run ? b : run;
                        int bqv = W.baq_plain ? b : (left < run ? left : run);    // plain (calmd -r without -E): min(qual, q) per base
                        int q0 = qual[i];
                        int tag = 64 + (q0 <= bqv ? 0 : q0 - bqv);
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
}

static bool baq_use_ck() { const char *e = getenv("STA_BAQ_CHECKPOINT"); return e && atoi(e) > 0; }

static size_t baq_ck_slot_dbl(int lq_cap, int bw, int c)
{
    int nb = 2 * bw + 1;
    size_t nck = (size_t)((lq_cap - 1) / c);
    return nck * (3 * nb) * 64 + (size_t)(lq_cap + 1) / 2 * 64;
}

static size_t baq_slot_dbl(int lq_cap, int bw)
{
    int nb = 2 * bw + 1;
    return (size_t)lq_cap * (2 * nb) * 64 + (size_t)(lq_cap + 2) * 64 + (size_t)(lq_cap + 1) / 2 * 64;
}

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
    size_t slot = baq_slot_dbl(lq_cap, 8) * 8;
    if (baq_use_ck()) { size_t a = baq_ck_slot_dbl(lq_cap, 7, 4), b = baq_ck_slot_dbl(lq_cap, 8, 3); slot = (a > b ? a : b) * 8; }
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

template <int BW>
static void run_band(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int64_t g0, int64_t ng, int use_list, int pass)
{
    unsigned nb = (unsigned)((ng + 3) / 4);
    if (baq_use_ck()) {
        constexpr int C = BW == 7 ? 4 : 3;
        size_t cslot = baq_ck_slot_dbl(lq_cap, BW, C);
        if (pass == 0)
            hipLaunchKernelGGL((k_baq_ck_fwd<BW, C>), dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, cslot, lq_cap);
        else
            hipLaunchKernelGGL((k_baq_ck_bwd<BW, C>), dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, cslot, lq_cap);
        return;
    }
    size_t slot = baq_slot_dbl(lq_cap, BW);
    if (pass == 0)
        hipLaunchKernelGGL(k_baq_fwd<BW>, dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, slot, lq_cap);
    else
        hipLaunchKernelGGL(k_baq_bwd<BW>, dim3(nb), dim3(256), 0, s, r, w, g_tables, g0, ng, use_list, (double *)scratch, slot, lq_cap);
}

// One pass (0 = forward, 1 = backward + MAP + apply) over groups [g0, g0 + ng) of 64 reads; the engine calls the two passes
// chunk by chunk (a chunk = what fits the scratch slab) so that each launch can be timed on its own.
void sta_launch_baq_band(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, void *scratch, int lq_cap, int bw,
                         int64_t g0, int64_t ng, int use_list, int pass)
{
    if (!g_tables_init) {
        for (int i = 0; i < 256; ++i) g_tables.q2p[i] = (float)pow(10, -i / 10.);   // g_qual2prob (probaln.c), host libm
        g_tables_init = true;
    }
    if (r.n == 0 || lq_cap <= 0 || ng <= 0) return;
    if (bw == 7) run_band<7>(s, r, w, scratch, lq_cap, g0, ng, use_list, pass);
    else if (bw == 8) run_band<8>(s, r, w, scratch, lq_cap, g0, ng, use_list, pass);
}
