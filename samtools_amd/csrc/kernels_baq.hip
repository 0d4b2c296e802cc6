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
    int64_t r = first + t;
    uint32_t info = R.info[r];
    if (!(info & RI_BAQ)) return;

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
                        int bqv = left < run ? left : run;
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
static const int BAQ_LQ_MAX_DEFAULT = 0;

size_t sta_baq_scratch_bytes(int64_t n_reads, int max_lq, int max_bw)
{
    (void)n_reads; (void)max_lq; (void)max_bw;
    return (size_t)3 << 30;     // fixed 3 GiB slab, reads are processed in chunks that fit it
}

static BaqTables g_tables;
static bool g_tables_init = false;

void sta_launch_baq(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int redo, void *scratch, size_t scratch_bytes,
                         int lq_max, int bw_max)
{
    if (!g_tables_init) {
        for (int i = 0; i < 256; ++i) g_tables.q2p[i] = (float)pow(10, -i / 10.);   // g_qual2prob (probaln.c), host libm
        g_tables_init = true;
    }
    if (r.n == 0 || lq_max <= 0) return;
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
    for (int64_t first = 0; first < r.n; first += (int64_t)chunk) {
        int64_t count = r.n - first < (int64_t)chunk ? r.n - first : (int64_t)chunk;
        unsigned nb = (unsigned)((count + 63) / 64);
        hipLaunchKernelGGL(k_baq, dim3(nb), dim3(64), 0, s, r, w, g_tables, redo, first, count, dscr, dbl_per_read, idim_max, lq_max,
                           state_s, q_s);
    }
}
