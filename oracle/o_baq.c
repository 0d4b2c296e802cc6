/*
 * oracle/o_baq.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of HTSlib 1.23.1 BAQ: realn.c sam_prob_realn() and probaln.c
 * probaln_glocal() (absent from /root/reference; call site bam_plcmd.c:451,
 * flag 3 or 7 with -E).  Arithmetic follows SURVEY.md Appendix A.4/A.4.1 in
 * IEEE double, in the written order, compiled with -ffp-contract=off.
 * Pinned by goldens test/mpileup/expected/{16,19,21,22,33,34,a0}.out and
 * test/dat/mpileup.out.1.
 */
#include "o_plp.h"
#include <math.h>
#include <limits.h>

#define EI .25
#define EM .33333333333

static float g_qual2prob[256];
static int g_q2p_init = 0;

#define set_u(u, b, i, k) { int x_ = (i) - (b); x_ = x_ > 0 ? x_ : 0; (u) = ((k) - x_ + 1) * 3; }

typedef struct { float d, e; int bw; } probaln_par_t;

static int probaln_glocal(const uint8_t *ref, int l_ref, const uint8_t *query, int l_query,
                          const uint8_t *iqual, const probaln_par_t *c, int *state, uint8_t *q)
{
    double *f = NULL, *b = NULL, *s = NULL, m[9], sI, sM, bI, bM;
    float *qual = NULL;
    int bw, bw2, i, k, is_backward = 1, Pr;

    if (l_ref < 0 || l_query < 0 || l_query >= INT_MAX - 2) return INT_MIN;
    if (l_ref == 0 || l_query == 0) return 0;

    if (!g_q2p_init) {
        for (i = 0; i < 256; ++i) g_qual2prob[i] = (float)pow(10, -i / 10.);
        g_q2p_init = 1;
    }
    is_backward = state && q ? 1 : 0;
    bw = l_ref > l_query ? l_ref : l_query;
    if (bw > c->bw) bw = c->bw;
    if (bw < abs(l_ref - l_query)) bw = abs(l_ref - l_query);
    bw2 = bw * 2 + 1;
    size_t i_dim = (size_t)bw2 * 3 + 6;   /* wide rows; out-of-band cells stay zero */

    f = (double *)calloc(((size_t)l_query + 1) * i_dim, sizeof(double));
    if (is_backward) b = (double *)calloc(((size_t)l_query + 1) * i_dim, sizeof(double));
    s = (double *)calloc((size_t)l_query + 2, sizeof(double));
    qual = (float *)calloc((size_t)l_query, sizeof(float));
    for (i = 0; i < l_query; ++i) qual[i] = g_qual2prob[iqual ? iqual[i] : 30];

    sM = sI = 1. / (2 * l_query + 2);
    m[0*3+0] = (1 - c->d - c->d) * (1 - sM); m[0*3+1] = m[0*3+2] = c->d * (1 - sM);
    m[1*3+0] = (1 - c->e) * (1 - sI); m[1*3+1] = c->e * (1 - sI); m[1*3+2] = 0.;
    m[2*3+0] = 1 - c->e; m[2*3+1] = 0.; m[2*3+2] = c->e;
    bM = (1 - c->d) / l_ref; bI = c->d / l_ref;

    /*** forward ***/
    set_u(k, bw, 0, 0);
    f[0 * i_dim + (size_t)k] = s[0] = 1.;
    { /* f[1] */
        double *fi = &f[1 * i_dim], sum;
        int beg = 1, end = l_ref < bw + 1 ? l_ref : bw + 1, _beg, _end;
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u;
            double e = (ref[k-1] > 3 || query[0] > 3) ? 1. : ref[k-1] == query[0] ? 1. - qual[0] : qual[0] * EM;
            set_u(u, bw, 1, k);
            fi[u+0] = e * bM; fi[u+1] = EI * bI;
            sum += fi[u] + fi[u+1];
        }
        s[1] = sum;
        set_u(_beg, bw, 1, beg); set_u(_end, bw, 1, end); _end += 2;
        for (k = _beg; k <= _end; ++k) fi[k] /= sum;
    }
    for (i = 2; i <= l_query; ++i) {
        double *fi = &f[(size_t)i * i_dim], *fi1 = &f[(size_t)(i-1) * i_dim], sum, qli = qual[i-1];
        int beg = 1, end = l_ref, x, _beg, _end;
        uint8_t qyi = query[i - 1];
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = beg, sum = 0.; k <= end; ++k) {
            int u, v11, v01, v10;
            double e = (ref[k-1] > 3 || qyi > 3) ? 1. : ref[k-1] == qyi ? 1. - qli : qli * EM;
            set_u(u, bw, i, k); set_u(v11, bw, i-1, k-1); set_u(v10, bw, i-1, k); set_u(v01, bw, i, k-1);
            fi[u+0] = e * (m[0] * fi1[v11+0] + m[3] * fi1[v11+1] + m[6] * fi1[v11+2]);
            fi[u+1] = EI * (m[1] * fi1[v10+0] + m[4] * fi1[v10+1]);
            fi[u+2] = m[2] * fi[v01+0] + m[8] * fi[v01+2];
            sum += fi[u] + fi[u+1] + fi[u+2];
        }
        s[i] = sum;
        set_u(_beg, bw, i, beg); set_u(_end, bw, i, end); _end += 2;
        for (k = _beg, sum = 1./sum; k <= _end; ++k) fi[k] *= sum;
    }
    { /* f[l_query+1] */
        double sum;
        for (k = 1, sum = 0.; k <= l_ref; ++k) {
            int u;
            set_u(u, bw, l_query, k);
            if (u < 3 || u >= bw2*3+3) continue;
            sum += f[(size_t)l_query * i_dim + (size_t)u + 0] * sM + f[(size_t)l_query * i_dim + (size_t)u + 1] * sI;
        }
        s[l_query+1] = sum;
    }
    { /* likelihood */
        double p = 1., Pr1 = 0.;
        for (i = 0; i <= l_query + 1; ++i) {
            p *= s[i];
            if (p < 1e-100) Pr1 += -4.343 * log(p), p = 1.;
        }
        Pr1 += -4.343 * log(p * l_ref * l_query);
        Pr = (int)(Pr1 + .499);
        if (!is_backward) { free(f); free(s); free(qual); return Pr; }
    }
    /*** backward ***/
    for (k = 1; k <= l_ref; ++k) {
        int u;
        double *bi = &b[(size_t)l_query * i_dim];
        set_u(u, bw, l_query, k);
        if (u < 3 || u >= bw2*3+3) continue;
        bi[u+0] = sM / s[l_query] / s[l_query+1]; bi[u+1] = sI / s[l_query] / s[l_query+1];
    }
    for (i = l_query - 1; i >= 1; --i) {
        int beg = 1, end = l_ref, x, _beg, _end;
        double *bi = &b[(size_t)i * i_dim], *bi1 = &b[(size_t)(i+1) * i_dim], y = (i > 1), qli1 = qual[i];
        uint8_t qyi1 = query[i];
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = end; k >= beg; --k) {
            int u, v11, v01, v10;
            set_u(u, bw, i, k); set_u(v11, bw, i+1, k+1); set_u(v10, bw, i+1, k); set_u(v01, bw, i, k+1);
            double e = (k >= l_ref ? 0 : (ref[k] > 3 || qyi1 > 3) ? 1. : ref[k] == qyi1 ? 1. - qli1 : qli1 * EM) * bi1[v11];
            bi[u+0] = e * m[0] + EI * m[1] * bi1[v10+1] + m[2] * bi[v01+2];
            bi[u+1] = e * m[3] + EI * m[4] * bi1[v10+1];
            bi[u+2] = (e * m[6] + m[8] * bi[v01+2]) * y;
        }
        set_u(_beg, bw, i, beg); set_u(_end, bw, i, end); _end += 2;
        for (k = _beg, y = 1./s[i]; k <= _end; ++k) bi[k] *= y;
    }
    /*** MAP ***/
    for (i = 1; i <= l_query; ++i) {
        double sum = 0., *fi = &f[(size_t)i * i_dim], *bi = &b[(size_t)i * i_dim], max = 0.;
        int beg = 1, end = l_ref, x, max_k = -1;
        x = i - bw; beg = beg > x ? beg : x;
        x = i + bw; end = end < x ? end : x;
        for (k = beg; k <= end; ++k) {
            int u;
            double z;
            set_u(u, bw, i, k);
            z = fi[u+0] * bi[u+0]; if (z > max) max = z, max_k = (k-1)<<2 | 0; sum += z;
            z = fi[u+1] * bi[u+1]; if (z > max) max = z, max_k = (k-1)<<2 | 1; sum += z;
        }
        max /= sum; sum *= s[i];
        if (state) state[i-1] = max_k;
        if (q) k = (int)(-4.343 * log(1. - max) + .499), q[i-1] = (uint8_t)(k > 100 ? 99 : k);
    }
    free(f); free(b); free(s); free(qual);
    return Pr;
}

/* realn.c sam_prob_realn (Appendix A.4).  flag bits: 1 apply, 2 extended, 4 redo */
int o_prob_realn(orec_t *b, const char *ref, hpos_t ref_len, int flag)
{
    int k, bw, y, yb, ye, xb, xe, apply_baq = flag & 1, extend_baq = flag & 2, redo_baq = flag & 4;
    hpos_t i, x;
    uint32_t *cigar = b->cigar;
    probaln_par_t conf = { 0.001f, 0.1f, 10 };
    uint8_t *qual = b->qual;
    const uint8_t *bqt = NULL, *zqt = NULL;
    if ((b->flag & F_UNMAP) || b->l_qseq == 0 || qual[0] == (uint8_t)-1) return -1;

    bqt = rec_aux_get(b, "BQ");
    if (bqt && *bqt != 'Z') return -1;
    zqt = rec_aux_get(b, "ZQ");
    if (zqt && *zqt != 'Z') return -1;
    if (bqt && redo_baq) bqt = NULL;          /* tag deleted and recomputed */
    if (bqt && zqt) {                         /* both: the ZQ tag is removed from the record */
        rec_aux_del(b, zqt); zqt = NULL;
        bqt = rec_aux_get(b, "BQ");
    }
    if (bqt || zqt) {
        if ((apply_baq && zqt) || (!apply_baq && bqt)) return -3;
        if (bqt && apply_baq) {
            const uint8_t *bq = bqt + 1;
            for (i = 0; i < b->l_qseq; ++i)
                qual[i] = (uint8_t)(qual[i] + 64 < bq[i] ? 0 : qual[i] - ((int)bq[i] - 64));
            /* tag renamed BQ->ZQ in the record: mark so a second call is a no-op */
            ((uint8_t *)bqt)[-2] = 'Z';
        } else if (zqt && !apply_baq) {       /* ZQ back to BQ: the qualities get the stored difference back */
            const uint8_t *zq = zqt + 1;
            for (i = 0; i < b->l_qseq; ++i) qual[i] = (uint8_t)(qual[i] + ((int)zq[i] - 64));
            ((uint8_t *)zqt)[-2] = 'B';
        }
        return 0;
    }
    x = b->pos; y = 0; yb = ye = xb = xe = -1;
    for (k = 0; k < (int)b->n_cigar; ++k) {
        int op = cig_op(cigar[k]), l = (int)cig_len(cigar[k]);
        if (op == C_M || op == C_EQ || op == C_X) {
            if (yb < 0) yb = y;
            if (xb < 0) xb = (int)x;
            ye = y + l; xe = (int)x + l;
            x += l; y += l;
        } else if (op == C_S || op == C_I) y += l;
        else if (op == C_D) x += l;
        else if (op == C_N) return -1;
    }
    if (xb == -1) return -1;
    bw = 7;
    if (abs((xe - xb) - (ye - yb)) > bw) bw = abs((xe - xb) - (ye - yb)) + 3;
    conf.bw = bw;
    xb -= yb + bw / 2; if (xb < 0) xb = 0;
    xe += b->l_qseq - ye + bw / 2;
    if (xe - xb - b->l_qseq > bw)
        xb += (xe - xb - b->l_qseq - bw) / 2, xe -= (xe - xb - b->l_qseq - bw) / 2;
    {
        int L = b->l_qseq;
        size_t lref = xe > xb ? (size_t)(xe - xb) : 1;
        if (lref < (size_t)L) lref = (size_t)L;
        uint8_t *tseq = (uint8_t *)calloc((size_t)L + 1, 1);
        uint8_t *tref = (uint8_t *)calloc(lref + 1, 1);
        uint8_t *q = (uint8_t *)calloc((size_t)L + 1, 1);
        uint8_t *bq = (uint8_t *)calloc((size_t)L + 1, 1);
        int *state = (int *)calloc((size_t)L + 1, sizeof(int));
        for (i = 0; i < L; ++i) tseq[i] = (uint8_t)nt16_int[rec_seqi(b->seq, i)];
        for (i = xb; i < xe; ++i) {
            if (i >= ref_len || ref[i] == '\0') { xe = (int)i; break; }
            tref[i - xb] = (uint8_t)nt16_int[nt16_table[(unsigned char)ref[i]]];
        }
        if (xe - xb <= 0) {
            /* calmd only (mpileup skips reads that start behind the FASTA contig's end, bam_plcmd.c:440-445): the whole window lies
               behind the end of the sequence.  HTSlib's probaln_glocal returns 0 at once for l_ref <= 0 and sam_prob_realn goes on
               to read state[] and q[] it malloc'ed and never wrote: undefined in the reference.  Defined here (and in the engine,
               dev_util.h baq_geometry: !ok) as "the record is left alone". */
            free(tseq); free(tref); free(q); free(bq); free(state);
            return -1;
        }
        memcpy(bq, qual, (size_t)L);
        if (probaln_glocal(tref, xe - xb, tseq, L, qual, &conf, state, q) == INT_MIN) {
            free(tseq); free(tref); free(q); free(bq); free(state);
            return -1;
        }
        if (!extend_baq) {
            for (k = 0, x = b->pos, y = 0; k < (int)b->n_cigar; ++k) {
                int op = cig_op(cigar[k]), l = (int)cig_len(cigar[k]);
                if (op == C_M || op == C_EQ || op == C_X) {
                    if (l > L - y) l = L - y;
                    for (i = y; i < y + l; ++i) {
                        if ((state[i] & 3) != 0 || state[i] >> 2 != x - xb + (i - y)) bq[i] = 0;
                        else bq[i] = bq[i] < q[i] ? bq[i] : q[i];
                    }
                    x += l; y += l;
                } else if (op == C_S || op == C_I) {
                    if (l > L - y) l = L - y;
                    y += l;
                } else if (op == C_D) x += l;
            }
            for (i = 0; i < L; ++i) bq[i] = (uint8_t)(qual[i] - bq[i] + 64);
        } else {
            uint8_t *left = tseq, *rght = tref;
            for (k = 0, x = b->pos, y = 0; k < (int)b->n_cigar; ++k) {
                int op = cig_op(cigar[k]), l = (int)cig_len(cigar[k]);
                if (op == C_M || op == C_EQ || op == C_X) {
                    if (l > L - y) l = L - y;
                    if (l > 0) {
                        for (i = y; i < y + l; ++i)
                            bq[i] = (uint8_t)(((state[i] & 3) != 0 || state[i] >> 2 != x - xb + (i - y)) ? 0 : q[i]);
                        for (left[y] = bq[y], i = y + 1; i < y + l; ++i)
                            left[i] = bq[i] > left[i-1] ? bq[i] : left[i-1];
                        for (rght[y+l-1] = bq[y+l-1], i = y + l - 2; i >= y; --i)
                            rght[i] = bq[i] > rght[i+1] ? bq[i] : rght[i+1];
                        for (i = y; i < y + l; ++i)
                            bq[i] = left[i] < rght[i] ? left[i] : rght[i];
                    }
                    x += l; y += l;
                } else if (op == C_S || op == C_I) {
                    if (l > L - y) l = L - y;
                    y += l;
                } else if (op == C_D) x += l;
            }
            for (i = 0; i < L; ++i) bq[i] = (uint8_t)(64 + (qual[i] <= bq[i] ? 0 : qual[i] - bq[i]));
        }
        if (apply_baq)
            for (i = 0; i < L; ++i) qual[i] -= bq[i] - 64;
        free(tseq); free(tref); free(q); free(bq); free(state);
    }
    return 0;
}
