/*
 * oracle/o_consensus.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of `samtools consensus`: the column iterator of
 * consensus_pileup.c:69-286 (get_next_base) and :301-608 (pileup_loop), the two callers
 * bam_consensus.c:2191-2317 (basic_pileup) and :2323-2455 (basic_fasta), the frequency
 * caller :1907-2014 (calculate_consensus_simple), the Bayesian caller :1258-1793
 * (calculate_consensus_gap5, without the compiled-out K2/DO_* blocks) + :1799-1880
 * (mixed mode) + :2139-2183 (consensus_base), the per-read preparation :1012-1206
 * (nm_init) and :943-973 (homopoly_qual_fix), the tables :740-883 (consensus_init) and
 * bam_consensus_tab.h (generated here from the formulas that header documents), and the
 * serial driver :2898-3075 (pileup_loop_serial).  Everything it needs is inside the
 * reference tree, so this oracle is pinned on test/consensus/consensus.reg.
 *
 * Not restated: the threaded driver (:2626-2890; same text by construction), the built-in
 * platform calibration tables of :448-664 (-X / --qual-calibration :name other than :flat)
 * and CRAM.  Region iterators are emulated by a filtered scan (o_io.c rd_set_region); the
 * --regions-file contig order is khash's bucket order (FNV-1a string hash), emulated below.
 */
#include <ctype.h>
#include <float.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <strings.h>
#include <zlib.h>
#include "o_common.h"

enum { FMT_FASTQ, FMT_FASTA, FMT_PILEUP, FMT_DUMP };       /* FMT_DUMP: every pileup_t field seq_column sees (for the pileup_loop() surface tests) */
enum { MODE_SIMPLE, MODE_BAYES_116, MODE_RECALL, MODE_PRECISE, MODE_MIXED };

typedef struct { int smap[101], umap[101], omap[101]; } qcal_t;

typedef struct {
    const char *reg;
    int use_qual, min_qual, adj_qual, use_mqual;
    double scale_mqual;
    int nm_adjust, nm_halo, sc_cost, low_mqual, high_mqual, min_depth;
    double call_fract, het_fract;
    int mode, fmt, cons_cutoff, ambig, line_len, default_qual, all_bases, show_del, show_ins, mark_ins;
    int excl_flags, incl_flags, min_mqual;
    double P_het, P_indel, het_scale, homopoly_fix, homopoly_redux;
    qcal_t qcal;
    const char *ref_fn;
    int ref_qual;
    const char *bed_fn;
    FILE *out;
    ofasta_t *fa;
    ohdr_t *h;
} copts_t;

/* ---------------- tables (bam_consensus.c:354-358, 361-377, 740-883; bam_consensus_tab.h) ---------------- */
static double e_tab_a[1002], *e_tab = &e_tab_a[500];
static double e_tab2_a[1002], *e_tab2 = &e_tab2_a[500];
static double q2p[101], mqual_pow_1m[256];

typedef struct {
    double prior[25], lprior15[15];
    double pMM[101], pxx[101], pxM[101], pox[101], poM[101], poo[101], puu[101], pum[101], pmm[101];
    double poly_mul;
} cprobs_t;
static cprobs_t cp_recall, cp_precise;

static void static_tables(void)
{
    for (int i = 0; i <= 100; i++) q2p[i] = pow(10, -i / 10.0);
    for (int i = 0; i < 255; i++) mqual_pow_1m[i] = pow(10, -(i * .9) / 10.0);
    mqual_pow_1m[255] = mqual_pow_1m[10];
}

static void cons_init(double p_het, double p_indel, double het_scale, double poly_mul, const qcal_t *qc, int mode, cprobs_t *cp)
{
    for (int i = -500; i <= 500; i++) { e_tab[i] = exp(i); e_tab2[i] = exp(i / 10.); }
    cp->poly_mul = poly_mul;
    for (int i = 0; i < 25; i++) cp->prior[i] = p_het / 6;
    for (int i = 0; i < 25; i += 6) cp->prior[i] = 1;
    for (int i = 4; i < 24; i += 5) cp->prior[i] = p_indel / 6;
    for (int i = 20; i < 24; i++) cp->prior[i] = p_indel / 6;
    static const int upper[15] = { 0, 1, 2, 3, 4, 6, 7, 8, 9, 12, 13, 14, 18, 19, 24 };
    for (int j = 0; j < 15; j++) cp->lprior15[j] = log(cp->prior[upper[j]]);

    for (int i = 1; i < 101; i++) {
        double prob = 1 - pow(10, -qc->smap[i] / 10.0);
        cp->pMM[i] = log(prob);
        cp->pxx[i] = log((1 - prob) / 3);
        cp->pxM[i] = log((exp(cp->pMM[i]) + exp(cp->pxx[i])) / 2);
        cp->pxM[i] += log(het_scale);
        if (mode == MODE_BAYES_116) {
            cp->pmm[i] = cp->pMM[i];
            cp->poM[i] = cp->pum[i] = cp->pxM[i];
            cp->pox[i] = cp->poo[i] = cp->puu[i] = cp->pxx[i];
            continue;
        }
        prob = 1 - pow(10, -qc->omap[i] / 10.0);
        cp->poo[i] = log((1 - prob) / 3);
        if (cp->poo[i] > cp->pMM[i] - .5) cp->poo[i] = cp->pMM[i] - .5;
        cp->pox[i] = log((exp(cp->poo[i]) + exp(cp->pxx[i])) / 2);
        cp->poM[i] = log((exp(cp->poo[i]) + exp(cp->pMM[i])) / 2);
        if (cp->poM[i] > cp->pxM[i] + .5) cp->poM[i] = cp->pxM[i] + .5;
        prob = 1 - pow(10, -qc->umap[i] / 10.0);
        cp->pmm[i] = log(prob);
        cp->puu[i] = log((1 - prob) / 3);
        if (cp->puu[i] > cp->pMM[i] - .5) cp->puu[i] = cp->pMM[i] - .5;
        cp->pum[i] = log((exp(cp->puu[i]) + exp(cp->pmm[i])) / 2);
    }
    double *all[9] = { cp->pMM, cp->pxx, cp->pxM, cp->pmm, cp->poo, cp->pox, cp->poM, cp->puu, cp->pum };
    for (int k = 0; k < 9; k++) all[k][0] = all[k][1];
}

/* bam_consensus.c:885-916 */
static inline double fast_exp(double y)
{
    if (y >= -50 && y <= 50) return e_tab2[(int)(y * 10)];
    if (y < -500) y = -500;
    if (y > 500) y = 500;
    return e_tab[(int)y];
}
static inline double fast_log2(double val)
{
    union { double d; uint64_t x; } u = { val };
    const int E = (int)((u.x >> 52) & 2047) - 1024;
    u.x &= ~(2047ULL << 52);
    u.x += 1023ULL << 52;
    val = ((-1 / 3.) * u.d + 2) * u.d - 2 / 3.;
    return E + val;
}
#define TENLOG2OVERLOG10 3.0103
#define ph_log(x) (-TENLOG2OVERLOG10 * fast_log2((x)))

/* ---------------- one read in the column iterator (consensus_pileup.h:41-75) ---------------- */
typedef struct cread {
    struct cread *next, *eofn, *eofl;
    int *nm;              /* client data of nm_init: [i] = poly-run length << 24 | local edit cost */
    int eof, qual, start, base, ref_skip, padding, base4;
    hpos_t pos;
    int nth, is_rev, seq_off;
    int cig_ind, cig_op, cig_len, first_del;
    orec_t b;
} cread_t;

/* qual byte at index i.  The reference reads b_qual[seq_offset+1] (and b_qual[seq_offset] for reads without SEQ) without a
 * bounds check: exactly one past the array that is the first aux byte of the BAM record, further out it is undefined
 * (whatever follows in memory) and reads as 0 here. */
static inline int qual_at(const orec_t *b, int i)
{
    if (i >= 0 && i < b->l_qseq) return b->qual[i];
    return i == b->l_qseq && b->l_aux > 0 ? b->aux[0] : 0;
}

static int take_op(cread_t *p)
{
    if (p->cig_ind >= (int)p->b.n_cigar) return 0;
    p->cig_op = cig_op(p->b.cigar[p->cig_ind]);
    p->cig_len = cig_len(p->b.cigar[p->cig_ind]);
    p->cig_ind++;
    return 1;
}

/* consensus_pileup.c:69-286: state of read p at column (pos, nth); *ins = inserted bases this read still has here.
 * 1 fetched, 0 ran off the read, -1 bad CIGAR op. */
static int next_base(cread_t *p, hpos_t pos, int nth, int *ins)
{
    const orec_t *b = &p->b;
    int op = p->cig_op;
    if (p->start > 0) p->start--;
    if (p->first_del && op != C_P) p->first_del = 0;
    *ins = 0;

    while (p->pos < pos) {                       /* up to the reference column */
        p->nth = 0;
        if (p->cig_len == 0) {
            if (!take_op(p)) { p->eof = 1; return 0; }
            op = p->cig_op;
        }
        const int aligned = op == C_M || op == C_EQ || op == C_X;
        if (aligned && p->cig_len <= pos - p->pos) {
            p->seq_off += p->cig_len; p->pos += p->cig_len; p->cig_len = 0;
        } else if (aligned) {
            p->seq_off++; p->pos++; p->cig_len--;
        } else if (op == C_D || op == C_N) {
            p->pos++; p->cig_len--;
        } else if (op == C_I || op == C_S) {
            p->seq_off += p->cig_len; p->cig_len = 0;
        } else if (op == C_P || op == C_H) {
            p->cig_len = 0;
        } else {
            fprintf(stderr, "Unhandled cigar_op %d\n", op);
            return -1;
        }
    }
    while (p->nth < nth) {                       /* then along the insertion */
        if (p->cig_len == 0) {
            if (!take_op(p)) { p->eof = 1; return 0; }
            op = p->cig_op;
        }
        if (op == C_I) { p->seq_off++; p->cig_len--; p->nth++; }
        else if (op == C_P) { p->cig_len--; p->nth++; }
        else if (op == C_H) p->cig_len = 0;
        else if (op == C_M || op == C_EQ || op == C_X || op == C_S || op == C_D || op == C_N) break;
        else { fprintf(stderr, "Unhandled cigar_op %d\n", op); return -1; }
    }

    p->ref_skip = 0;
    if (p->nth < nth && op != C_I) {             /* a pad opposite another read's insertion */
        p->base = '*'; p->base4 = 16; p->padding = 1;
        if (p->seq_off < b->l_qseq) { int q = qual_at(b, p->seq_off + 1); if (q < p->qual) p->qual = q; }
        else p->qual = 0;
    } else {
        p->padding = 0;
        if (op == C_D || op == C_P) {
            p->base = '*'; p->base4 = 16;
            int q = p->seq_off + 1 < b->l_qseq ? b->qual[p->seq_off + 1] : qual_at(b, p->seq_off);
            if (q < p->qual) p->qual = q;
        } else if (op == C_N) {
            p->base = '.'; p->base4 = 0; p->qual = 0;
            p->eof = p->eof ? 2 : 3;
            p->ref_skip = 1;
        } else if (p->seq_off < b->l_qseq) {
            p->qual = b->qual[p->seq_off];
            p->base4 = rec_seqi(b->seq, p->seq_off);
            p->base = "NACMGRSVTWYHKDBN"[p->base4];
        } else {
            p->base = 'N'; p->base4 = 15; p->qual = 0xff;
        }
    }
    if (p->eof && p->base != '.') { p->start = 1; p->ref_skip = 1; p->eof = 0; }   /* out of a ref skip again */
    if (p->start && p->cig_op == C_D) p->first_del = 1;

    if (p->cig_len == 0) {                       /* peek: is an insertion next? */
        if (take_op(p)) {
            op = p->cig_op;
            if (op == C_N) { p->eof = 3; p->ref_skip = 1; }
        } else p->eof = 1;
    }
    if (op == C_P || op == C_I) *ins = p->cig_len;
    else if (op == C_S)
        p->eof = (p->cig_ind == (int)b->n_cigar || (p->cig_ind + 1 == (int)b->n_cigar && cig_op(b->cigar[p->cig_ind]) == C_H)) ? 1 : 0;
    else if (op == C_H) p->eof = 1;
    return 1;
}

/* ---------------- job context (bam_consensus.c:263-293) ---------------- */
typedef struct {
    copts_t *o;
    ostr_t row, seq, qual;
    hpos_t last_pos;
    int last_tid;
    const char *ref; hpos_t ref_len; int ref_tid;
    int has_iter, iter_tid; hpos_t iter_beg, iter_end;
    oreader_t *rd;
} cctx_t;

/* bam_consensus.c:2024-2052 */
static hpos_t update_ref(cctx_t *c, int tid)
{
    copts_t *o = c->o;
    if (!o->ref_fn) return 0;
    if (tid == c->ref_tid && c->ref) return c->ref_len;
    c->ref = NULL; c->ref_tid = tid;
    if (tid < 0 || tid >= o->h->n_ref) return -1;
    c->ref = fa_fetch(o->fa, o->h->name[tid], &c->ref_len);
    return c->ref ? c->ref_len : -1;
}

/* bam_consensus.c:2083-2103 */
static int fetch_read(cctx_t *c, orec_t *b)
{
    copts_t *o = c->o;
    for (;;) {
        int r = rd_next(c->rd, b);
        if (r < 0) return r;
        if (o->incl_flags && !(b->flag & o->incl_flags)) continue;
        if (o->excl_flags && (b->flag & o->excl_flags)) continue;
        if (b->mapq < o->min_mqual) continue;
        return r;
    }
}

/* ---------------- per-read preparation (bam_consensus.c:943-973, 1012-1206) ---------------- */
static void homopoly_qual_fix(orec_t *b)
{
    static double ph2err[256];
    if (!ph2err[0]) for (int i = 0; i < 256; i++) ph2err[i] = pow(10, i / -10.0);
    for (int i = 0; i < b->l_qseq; i++) {
        const int s = i, base = rec_seqi(b->seq, i);
        while (i + 1 < b->l_qseq && rec_seqi(b->seq, i + 1) == base) i++;
        for (int j = s, k = i; j < k; j++, k--) {
            double e = ph2err[b->qual[j]] + ph2err[b->qual[k]];
            b->qual[j] = b->qual[k] = (uint8_t)(-fast_log2(e / 2) * 3.0104 + .49);
        }
    }
}

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

static int read_init(cctx_t *c, cread_t *p)
{
    copts_t *o = c->o;
    if (!o->use_mqual) return 1;
    orec_t *b = &p->b;
    const int qlen = b->l_qseq;
    if (qlen <= 0) return 0;
    int *nm = (int *)calloc((size_t)qlen, sizeof(int));
    if (!nm) return -1;
    p->nm = nm;
    const double poly_adj = o->homopoly_fix ? o->homopoly_fix : 1;
    int i;

    if (o->adj_qual) {
        const uint8_t *qual = b->qual, *seq = b->seq;
        const int qhalo = 8, qhalop = 2;
        int qmin = qual[0], qminp = qual[0];
        int base = rec_seqi(seq, 0), polyl = 0, polyr = 0;
        for (i = 1; i < qlen; i++) {
            if (rec_seqi(seq, i) != base) break;
            if (i < qhalop && qminp > qual[i]) qminp = qual[i];
        }
        for (i = 0; i < qlen && i < qhalo; i++) if (qmin > qual[i]) qmin = qual[i];
        for (; i < qlen - qhalo; i++) {
            if (o->homopoly_fix && rec_seqi(seq, i) != base) {
                polyl = i; base = rec_seqi(seq, i); qminp = qual[i];
                int j;
                for (j = i + 1; j < qlen; j++) {
                    if (rec_seqi(seq, j) != base) break;
                    if (i < qhalop && qminp > qual[j]) qminp = qual[j];
                }
                polyr = j - 1;
            } else polyr = polyl;
            const int pl = polyr - polyl;
            int t = o->mode == MODE_BAYES_116 ? (qual[i] + 5 * qmin) / 4 : (int)(qual[i] / 3 + (qminp - pl * 2) * poly_adj);
            nm[i] += t < qual[i] ? qual[i] - t : 0;
            qminp = qual[i];
            for (int k = IMAX(polyl, i - qhalop); k <= IMIN(polyr, i + qhalop); k++) if (qminp > qual[k]) qminp = qual[k];
            if (qmin > qual[i + qhalo]) qmin = qual[i + qhalo];
            else if (qmin <= qual[i - qhalo]) {
                qmin = 99;
                for (int j = i - qhalo + 1; j <= i + qhalo; j++) if (qmin > qual[j]) qmin = qual[j];
            }
        }
        for (; i < qlen; i++) {
            int t = o->mode == MODE_BAYES_116 ? (qual[i] + 5 * qmin) / 4 : (int)(qual[i] / 3 + qminp * poly_adj);
            nm[i] += t < qual[i] ? qual[i] - t : 0;
        }
    }
    if (o->homopoly_fix) homopoly_qual_fix(b);

    for (i = 0; i < qlen; i++) {                 /* run length (minus one, capped) into the top byte */
        const int base = rec_seqi(b->seq, i);
        int j;
        for (j = i + 1; j < qlen; j++) if (rec_seqi(b->seq, j) != base) break;
        int poly = j - i - 1; if (poly > 100) poly = 100;
        for (int k = i; k < j; k++) nm[k] = (IMAX(poly, nm[k] >> 24) << 24) | (nm[k] & ((1 << 24) - 1));
        i = j - 1;
    }

    const int halo = o->nm_halo;
    const uint8_t *md = rec_aux_get(b, "MD");
    if (!md || *md != 'Z') return 1;
    md++;
    const uint32_t *cig = b->cigar; const int ncig = (int)b->n_cigar;
    if (cig_op(cig[0]) == C_S || (cig_op(cig[0]) == C_H && ncig > 1 && cig_op(cig[1]) == C_S)) {
        for (i = 0; i < halo && i < qlen; i++) nm[i] += o->sc_cost;
        for (; i < halo * 2 && i < qlen; i++) nm[i] += o->sc_cost >> 1;
    }
    if (cig_op(cig[ncig - 1]) == C_S || (cig_op(cig[ncig - 1]) == C_H && ncig > 1 && cig_op(cig[ncig - 2]) == C_S)) {
        for (i = qlen - 1; i >= qlen - halo && i >= 0; i--) nm[i] += o->sc_cost;
        for (; i >= qlen - halo * 2 && i >= 0; i--) nm[i] += o->sc_cost >> 1;
    }
    int pos = 0;                                  /* counts matched bases only: substitutions do not advance it */
    while (*md) {
        if (isdigit(*md)) { char *e; pos += (int)strtol((const char *)md, &e, 10); md = (const uint8_t *)e; continue; }
        if (*md == '^') { while (*++md && !isdigit(*md)) continue; continue; }
        for (i = pos - halo * 2 >= 0 ? pos - halo * 2 : 0; i < pos - halo && i < qlen; i++) nm[i] += 5;
        for (; i < pos + halo && i < qlen; i++) nm[i] += 10;
        for (; i < pos + halo * 2 && i < qlen; i++) nm[i] += 5;
        md++;
    }
    return 1;
}

/* bam_consensus.c:978-1000 (argument already reduced to the query index seq_offset+1) */
static double nm_local(const cread_t *p, hpos_t qi)
{
    if (!p->nm) return 0;
    if (qi < 0) return p->nm[0] & ((1 << 24) - 1);
    if (qi >= p->b.l_qseq) return p->nm[p->b.l_qseq - 1] & ((1 << 24) - 1);
    return (p->nm[qi] & ((1 << 24) - 1)) / 10.0;
}
static int poly_len(const cread_t *p, hpos_t qi)
{
    if (!p->nm) return 0;
    return qi >= 0 && qi < p->b.l_qseq ? p->nm[qi] >> 24 : 0;
}

/* ---------------- the callers ---------------- */
typedef struct { int call, het_call, het_logodd, phred, depth; } cons_t;

/* genotype j of the 15: alleles (A[j], B[j]) over ACGT* */
static const int GA[15] = { 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4 };
static const int GB[15] = { 0, 1, 2, 3, 4, 1, 2, 3, 4, 2, 3, 4, 3, 4, 4 };

/* bam_consensus.c:1258-1793 */
static void consensus_gap5(int use_mq, int td, const cread_t *plp, const copts_t *o, cons_t *cons, const cprobs_t *cp)
{
    const double min_e_exp = DBL_MIN_EXP * log(2) + 1;
    double S[15] = { 0 };
    int counts[6] = { 0 }, depth = 0;
    static const int L[32] = { 5, 0, 1, 5, 2, 5, 5, 5, 3, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4 };

    for (const cread_t *p = plp; p; p = p->next) {
        if (p->qual < o->min_qual) continue;
        if (p->ref_skip) continue;
        const orec_t *b = &p->b;
        uint8_t qual = (uint8_t)p->qual;
        const int q0 = b->l_qseq > 0 ? b->qual[0] : (b->l_aux > 0 ? b->aux[0] : 0);
        if (qual == 255 || (qual == 0 && q0 == 255)) qual = (uint8_t)o->default_qual;
        const int base = L[p->base4];

        if (use_mq) {
            double mqual = b->mapq;
            if (o->nm_adjust) {
                mqual /= (nm_local(p, p->seq_off + 1) + 1);
                mqual *= 1 + 2 * (0.5 - (td > 30 ? 30 : td) / 60.0);
            }
            mqual *= o->scale_mqual;
            if (mqual < o->low_mqual) mqual = o->low_mqual;
            if (mqual > o->high_mqual) mqual = o->high_mqual;
            const double P = q2p[qual > 100 ? 100 : qual], M = mqual_pow_1m[(int)mqual];
            qual = (uint8_t)ph_log(P + .75 * M - P * M);
        }
        if (qual < 1) qual = 1;
        const double poly = poly_len(p, p->seq_off + 1);
        const double q2d = qual - (poly - 2) * cp->poly_mul;
        const int qual2 = (int)(1 > q2d ? 1 : q2d);

        const double xx = cp->pxx[qual];
        const double MM = cp->pMM[qual] - xx, xM = cp->pxM[qual] - xx;
        const double oo = cp->poo[qual2] - xx, oM = cp->poM[qual2] - xx, ox = cp->pox[qual2] - xx;
        const double uu = cp->puu[qual2] - xx, um = cp->pum[qual2] - xx, mm = cp->pmm[qual2] - xx;
        counts[base]++;
        for (int j = 0; j < 15; j++) {           /* exactly one addend per genotype per read */
            const int a = GA[j], c = GB[j];
            double v;
            if (base < 4) {
                const int ha = a == base, hc = c == base;
                if (c == 4) v = a == 4 ? oo : (ha ? oM : ox);
                else if (ha && hc) v = MM;
                else if (ha || hc) v = xM;
                else continue;                   /* xx - xx = nothing added */
            } else if (base == 4) v = c == 4 ? (a == 4 ? mm : um) : uu;
            else v = c == 4 ? (a == 4 ? oo : oM) : MM;
            S[j] += v;
        }
        depth++;
    }

    static const int map_sing[15] = { 0, 5, 5, 5, 5, 1, 5, 5, 5, 2, 5, 5, 3, 5, 4 };
    static const int map_het[15] = { 0, 1, 2, 3, 4, 6, 7, 8, 9, 12, 13, 14, 18, 19, 24 };
    double shift = -DBL_MAX, max = -DBL_MAX, max_het = -DBL_MAX, norm[15], tot1 = 0, tot2 = 0;
    int call = 0, het_call = 0;
    for (int j = 0; j < 15; j++) {
        S[j] += cp->lprior15[j];
        if (shift < S[j]) shift = S[j];
        if (GA[j] != GB[j]) { if (max_het < S[j]) { max_het = S[j]; het_call = j; } continue; }
        if (max < S[j]) { max = S[j]; call = j; }
    }
    for (int j = 0; j < 15; j++) {
        S[j] -= shift;
        const double e = fast_exp(S[j]);
        S[j] = S[j] > min_e_exp ? e : DBL_MIN;
        norm[j] = 0;
    }
    for (int j = 0; j < 15; j++) {
        norm[j] += tot1; norm[14 - j] += tot2;
        tot1 += S[j]; tot2 += S[14 - j];
    }
    if (!depth || depth == counts[5]) { cons->call = 4; cons->het_call = 0; cons->het_logodd = 0; cons->phred = 0; cons->depth = 0; return; }
    cons->depth = depth;
    if (norm[call] == 0) norm[call] = DBL_MIN;
    int ph;
    if (S[call] == 1 && norm[call] < .01) ph = (int)(ph_log(norm[call]) + .5);
    else ph = (int)(ph_log(1 - S[call] / (norm[call] + S[call])) + .5);
    cons->call = map_sing[call];
    cons->phred = ph < 0 ? 0 : ph;
    if (norm[het_call] == 0) norm[het_call] = DBL_MIN;
    ph = (int)(TENLOG2OVERLOG10 * (fast_log2(S[het_call]) - fast_log2(norm[het_call])) + .5);
    cons->het_call = map_het[het_call];
    cons->het_logodd = ph;
}

/* bam_consensus.c:1799-1880 */
static void consensus_gap5m(int use_mq, int depth, const cread_t *plp, const copts_t *o, cons_t *cons)
{
    if (o->mode != MODE_MIXED) { consensus_gap5(use_mq, depth, plp, o, cons, o->mode == MODE_PRECISE ? &cp_precise : &cp_recall); return; }
    cons_t P, R;
    consensus_gap5(use_mq, depth, plp, o, &P, &cp_precise);
    consensus_gap5(use_mq, depth, plp, o, &R, &cp_recall);
    *cons = P;
    if (P.phred > 0 && R.phred > 0 && P.call == R.call) cons->phred += IMIN(20, R.phred);
    else if (P.het_logodd >= 0 && R.het_logodd >= 0 && P.het_call == R.het_call) cons->het_logodd += IMIN(20, R.het_logodd);
    else if (P.het_logodd >= 0) { int q2 = IMAX(R.phred, R.het_logodd); cons->het_logodd = IMAX(1, (cons->het_logodd - q2 / 2)); }
    else if (R.het_logodd >= 70) {
        int q1 = P.phred, q2 = R.het_logodd;
        *cons = R;
        double a = (q2 - q1 * 2) / 2, bb = 1 + q2 / (q1 + 1.0), m = a > bb ? a : bb;
        cons->het_logodd = (int)(15 < m ? 15 : m);
    } else if (R.het_logodd >= 0) {
        int q1 = P.phred, q2 = R.het_logodd;
        *cons = R;
        double v = q2 - 0.3 * q1;
        cons->het_logodd = (int)((1 > v ? 1 : v) + 5 * (P.het_call == R.het_call));
        cons->phred = 0;
    } else {
        R.phred = R.phred / 2;
        if (R.phred > P.phred) *cons = R;
        cons->phred = IMAX(10, cons->phred);
    }
}

/* bam_consensus.c:1907-2014 */
static int consensus_simple(const cread_t *plp, const copts_t *o, int *qual)
{
    static const int wA[16] = { 0, 8, 0, 4, 0, 4, 0, 2, 0, 4, 0, 2, 0, 2, 0, 1 };
    static const int wC[16] = { 0, 0, 8, 4, 0, 0, 4, 2, 0, 0, 4, 2, 0, 0, 2, 1 };
    static const int wG[16] = { 0, 0, 0, 0, 8, 4, 4, 1, 0, 0, 0, 0, 4, 2, 2, 1 };
    static const int wT[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 8, 4, 4, 2, 8, 2, 2, 1 };
    uint64_t score[5] = { 0 };                   /* A C G T * */
    int tot_depth = 0;
    for (const cread_t *p = plp; p; p = p->next) {
        const int q = p->qual;
        if (q < o->min_qual) continue;
        const int w = o->use_qual ? q : 1, b = p->base4;
        if (b < 16) {
            score[0] += (uint64_t)(int64_t)(wA[b] * w); score[1] += (uint64_t)(int64_t)(wC[b] * w);
            score[2] += (uint64_t)(int64_t)(wG[b] * w); score[3] += (uint64_t)(int64_t)(wT[b] * w);
        } else score[4] += (uint64_t)(int64_t)(8 * w);
        tot_depth++;
    }
    uint64_t tscore = 0, score1 = 0, score2 = 0;
    int call1 = 15, call2 = 15;
    for (int i = 0; i < 5; i++) tscore += score[i];
    for (int i = 0; i < 5; i++) {
        const int c = 1 << i;
        if (score1 < score[i]) { score2 = score1; call2 = call1; score1 = score[i]; call1 = c; }
        else if (score2 < score[i]) { score2 = score[i]; call2 = c; }
    }
    uint64_t used_score = score1;
    int used_base = call1;
    if (score2 >= o->het_fract * score1 && o->ambig) { used_base |= call2; used_score += score2; }
    if (tot_depth < o->min_depth || used_score < o->call_fract * tscore) used_base = call1 == 16 ? 16 : 0;
    if (qual) *qual = used_base ? (int)(100.0 * used_score / tscore) : 0;
    return "NACMGRSVTWYHKDBN*ac?g???t???????"[used_base];
}

/* bam_consensus.c:2139-2183 */
static void consensus_base(const copts_t *o, const cread_t *p, int depth, int *base, int *qual)
{
    int cb, cq;
    if (o->mode == MODE_SIMPLE) { cb = consensus_simple(p, o, &cq); *base = cb; *qual = cq; return; }
    cons_t cons;
    consensus_gap5m(o->use_mqual, depth, p, o, &cons);
    if (cons.depth < o->min_depth && cons.call != 4) { cb = 'N'; cq = 0; }
    else if (cons.het_logodd > 0 && o->ambig) { cb = "AMRWaMCSYcRSGKgWYKTtacgt*"[cons.het_call]; cq = cons.het_logodd; }
    else { cb = "ACGT*"[cons.call]; cq = cons.phred; }
    if (cq < o->cons_cutoff && cb != '*' && cons.het_call % 5 != 4 && cons.het_call / 5 != 4) { cb = 'N'; cq = 0; }
    *base = cb; *qual = cq;
}

/* bam_consensus.c:2107-2131 */
static void empty_rows(cctx_t *c, int tid, hpos_t start, hpos_t end)
{
    const char *rseq = NULL;
    if (c->o->ref_fn && update_ref(c, tid) > 0) rseq = c->ref;
    for (hpos_t i = start; i < end; i++)
        fprintf(c->o->out, "%s\t%lld\t0\t0\t%c\t0\t*\t*\n", c->o->h->name[tid], (long long)(i + 1), rseq ? rseq[i] : 'N');
}

/* bam_consensus.c:2191-2317 */
static int column_pileup(cctx_t *c, const cread_t *p, int depth, hpos_t pos, int nth)
{
    copts_t *o = c->o;
    const int tid = p->b.tid;
    if (!o->show_ins && nth) return 0;
    if (c->has_iter && (c->iter_beg >= pos || c->iter_end < pos)) return 0;
    if (o->all_bases) {
        if (tid != c->last_tid && c->last_tid >= -1) {
            if (c->last_tid >= 0) {
                hpos_t len = o->h->len[c->last_tid];
                if (c->has_iter && c->iter_end < len) len = c->iter_end;
                empty_rows(c, c->last_tid, c->last_pos, len);
            }
            c->last_pos = c->has_iter ? c->iter_beg : 0;
        }
        if (!c->has_iter && tid > c->last_tid && o->all_bases > 1)
            while (++c->last_tid < tid) empty_rows(c, c->last_tid, 0, o->h->len[c->last_tid]);
        if (c->last_pos >= 0 && pos > c->last_pos + 1) empty_rows(c, tid, c->last_pos, pos - 1);
        else if (c->last_pos < 0) empty_rows(c, tid, c->has_iter ? c->iter_beg : 0, pos - 1);
    }
    int cb, cq;
    consensus_base(o, p, depth, &cb, &cq);
    if (!o->show_del && cb == '*') return 0;
    ostr_t *ks = &c->row;
    os_clear(ks);
    os_puts(ks, o->h->name[tid]); os_putc(ks, '\t'); os_putll(ks, pos); os_putc(ks, '\t'); os_putll(ks, nth); os_putc(ks, '\t');
    os_putll(ks, depth); os_putc(ks, '\t'); os_putc(ks, cb); os_putc(ks, '\t'); os_putll(ks, cq); os_putc(ks, '\t');
    os_reserve(ks, (size_t)depth * 2 + 3);
    char *cp = ks->s + ks->l, *qp = cp + depth + 1;
    for (; p; p = p->next) {
        *cp++ = p->is_rev ? (p->base == '*' ? '#' : (char)tolower(p->base)) : (char)p->base;
        *qp++ = (char)(IMIN(p->qual, 93) + '!');
    }
    *cp = '\t'; *qp = '\n';
    ks->l += (size_t)depth * 2 + 2;
    fwrite(ks->s, 1, ks->l, o->out);
    c->last_pos = pos; c->last_tid = tid;
    return 0;
}

/* -f dump (not a samtools format): one row per column with what get_next_base left in every pileup_t */
static int column_dump(cctx_t *c, const cread_t *p, int depth, hpos_t pos, int nth)
{
    fprintf(c->o->out, "%s\t%lld\t%d\t%d", c->o->h->name[p->b.tid], (long long)pos, nth, depth);
    for (; p; p = p->next)
        fprintf(c->o->out, "\t%c,%d,%d,%d,%d,%d,%d", p->base, p->qual, p->base4, p->ref_skip, p->is_rev, p->seq_off, p->padding);
    fputc('\n', c->o->out);
    return 0;
}

/* bam_consensus.c:2054-2075 */
static void dump_fastq(const copts_t *o, const char *name, const ostr_t *seq, const ostr_t *qual)
{
    if (!seq->l) return;
    fprintf(o->out, "%c%s\n", ">@"[o->fmt == FMT_FASTQ], name);
    for (size_t i = 0; i < seq->l; i += (size_t)o->line_len) { size_t n = seq->l - i; if (n > (size_t)o->line_len) n = (size_t)o->line_len; fprintf(o->out, "%.*s\n", (int)n, seq->s + i); }
    if (o->fmt != FMT_FASTQ) return;
    fprintf(o->out, "+\n");
    for (size_t i = 0; i < seq->l; i += (size_t)o->line_len) { size_t n = seq->l - i; if (n > (size_t)o->line_len) n = (size_t)o->line_len; fprintf(o->out, "%.*s\n", (int)n, qual->s + i); }
}

static void fill_flat(cctx_t *c, hpos_t from, hpos_t n)
{
    for (hpos_t i = 0; i < n; i++) {
        os_putc(&c->seq, c->ref ? c->ref[from + i] : 'N');
        os_putc(&c->qual, (c->ref ? c->o->ref_qual : 0) + '!');
    }
}

/* bam_consensus.c:2323-2455 */
static int column_fasta(cctx_t *c, const cread_t *p, int depth, hpos_t pos, int nth)
{
    copts_t *o = c->o;
    const int tid = p->b.tid;
    if (!o->show_ins && nth) return 0;
    if (c->has_iter && (c->iter_beg >= pos || c->iter_end < pos)) return 0;
    while (tid != c->last_tid) {
        if (c->last_tid != -1) {
            if (o->all_bases) {
                hpos_t N;
                if (c->has_iter) { if (c->last_pos < c->iter_beg - 1) c->last_pos = c->iter_beg - 1; N = c->iter_end; }
                else N = HPOS_MAX;
                if (N > o->h->len[c->last_tid]) N = o->h->len[c->last_tid];
                N -= c->last_pos;
                if (N > 0) {
                    if (c->ref && update_ref(c, c->last_tid) < 0) return -1;
                    fill_flat(c, c->last_pos, N);
                }
            }
            dump_fastq(o, o->h->name[c->last_tid], &c->seq, &c->qual);
        }
        if (update_ref(c, tid) < 0) return -1;
        os_clear(&c->seq); os_clear(&c->qual);
        if (!c->has_iter && o->all_bases > 1 && ++c->last_tid < tid) { c->last_pos = 0; continue; }
        c->last_tid = tid;
        c->last_pos = o->all_bases ? (c->has_iter ? c->iter_beg : 0) : pos - 1;
    }
    int cb, cq;
    consensus_base(o, p, depth, &cb, &cq);
    if (!o->show_del && cb == '*') { c->last_pos = pos; c->last_tid = tid; return 0; }
    if (o->mark_ins && nth && cb != '*') { os_putc(&c->seq, '_'); os_putc(&c->qual, '_'); }
    if (pos > c->last_pos && (c->last_pos > 0 || o->all_bases)) {
        if (update_ref(c, tid) < 0) return -1;
        fill_flat(c, c->last_pos, pos - (c->last_pos + 1));
    }
    if ((nth && o->show_ins && cb != '*') || cb != '*' || (pos > c->last_pos && o->show_del)) {
        os_putc(&c->seq, cb);
        os_putc(&c->qual, IMIN(cq, '~' - '!') + '!');
    }
    c->last_pos = pos; c->last_tid = tid;
    return 0;
}

/* ---------------- the column loop (consensus_pileup.c:301-608) ---------------- */
static int column_loop(cctx_t *c)
{
    copts_t *o = c->o;
    const int with_init = o->mode != MODE_SIMPLE;
    cread_t *head = NULL, *tail = NULL, *pool = NULL, *p, *last;
    cread_t *fresh = (cread_t *)calloc(1, sizeof(*fresh));
    int ret = -1, nth = 0, r, last_ref = -1;
    hpos_t col = 0;
    do {
        hpos_t pos;
        r = fetch_read(c, &fresh->b);
        if (r < -1) { fprintf(stderr, "pileup_loop() seq_fetch failure.\n"); goto done; }
        const orec_t *b = &fresh->b;
        if (r >= 0) {
            if (b->flag & F_UNMAP) continue;
            if (b->tid == -1) continue;
            pos = b->tid == last_ref ? b->pos + 1 : HPOS_MAX;
        } else pos = HPOS_MAX;
        if (col > pos) { fprintf(stderr, "BAM/SAM file is not sorted by position. Aborting\n"); goto done; }

        while (col < pos && head) {
            cread_t *dead = NULL, *dead_tail = NULL;
            int depth = 0, most_ins = 0, ins = 0;
            for (p = head, last = NULL; p; p = p->next) {
                if (!next_base(p, col, nth, &ins)) p->eof = 1;
                if (p->eof == 1) {
                    if (dead_tail) dead_tail->eofn = p; else dead = p;
                    dead_tail = p; p->eofl = last; p->eofn = NULL;
                } else last = p;
                if (most_ins < ins) most_ins = ins;
                depth++;
            }
            tail = last ? last : head;
            int v = o->fmt == FMT_DUMP ? column_dump(c, head, depth, col, nth) : o->fmt == FMT_PILEUP ? column_pileup(c, head, depth, col, nth) : column_fasta(c, head, depth, col, nth);
            for (p = dead; p; p = p->eofn) {
                if (p->eofl) p->eofl->next = p->next; else head = p->next;
                p->next = pool; pool = p;
                if (with_init) { free(p->nm); p->nm = NULL; }
            }
            if (v != 0) goto done;
            if (most_ins) nth++; else { nth = 0; col++; }
        }
        col = pos;
        if (r >= 0 && b->tid != last_ref) { last_ref = b->tid; pos = b->pos + 1; nth = 0; col = pos; }

        if (r >= 0) {
            p = fresh;
            p->next = p->eofn = p->eofl = NULL; p->nm = NULL;
            p->start = 2; p->eof = 0;
            p->pos = pos - 1; p->cig_ind = 0; p->cig_len = 0; p->cig_op = -1; p->seq_off = -1; p->first_del = 0;
            p->is_rev = rec_is_rev(&p->b);
            int keep = with_init ? read_init(c, p) : 1;
            if (keep == -1) { p->next = pool; pool = p; fresh = NULL; goto done; }
            if (keep == 1) { if (head) tail->next = p; else head = p; tail = p; }
            else { p->next = pool; pool = p; }
            if (pool) { fresh = pool; pool = pool->next; }
            else fresh = (cread_t *)calloc(1, sizeof(*fresh));
        }
    } while (r >= 0);
    ret = 0;
done:
    if (fresh) { rec_free(&fresh->b); free(fresh); }
    for (p = pool; p; p = last) { last = p->next; free(p->nm); rec_free(&p->b); free(p); }
    for (p = head; p; p = last) { last = p->next; free(p->nm); rec_free(&p->b); free(p); }
    return ret;
}

/* ---------------- --regions-file: contigs in khash bucket order, intervals sorted by (beg, end) ---------------- */
typedef struct { char *name; int n, m; hpos_t *beg, *end; } cbed_chr_t;
typedef struct { int n_buckets, size, n_occupied, upper; int *slot; int n_chr; cbed_chr_t *chr; } cbed_t;   /* slot[b] = chr index or -1 */

static uint32_t fnv1a(const char *s) { uint32_t h = 2166136261u; for (; *s; ++s) h = (h ^ (uint8_t)*s) * 16777619u; return h; }

static void cbed_resize(cbed_t *t, int want)
{
    int nb = want - 1; nb |= nb >> 1; nb |= nb >> 2; nb |= nb >> 4; nb |= nb >> 8; nb |= nb >> 16; nb++;
    if (nb < 4) nb = 4;
    if (t->size >= (int)(nb * 0.77 + 0.5)) return;
    int *ns = (int *)malloc(sizeof(int) * (size_t)nb);
    for (int i = 0; i < nb; i++) ns[i] = -1;
    /* khash rehashes in place with a kick-out chain; with no deletions the resulting placement equals inserting the old
     * buckets in ascending order -- except that a kicked-out key continues the chain.  Emulate the chain literally. */
    int old_n = t->n_buckets;
    int *old = t->slot;                              /* old[j] = key or -1 */
    int *moved = (int *)calloc((size_t)(old_n > 0 ? old_n : 1), sizeof(int));
    const uint32_t mask = (uint32_t)nb - 1;
    for (int j = 0; j < old_n; j++) {
        if (old[j] < 0 || moved[j]) continue;
        int key = old[j];
        moved[j] = 1;
        for (;;) {
            uint32_t i = fnv1a(t->chr[key].name) & mask, step = 0;
            while (ns[i] >= 0) i = (i + (++step)) & mask;
            ns[i] = key;
            if ((int)i < old_n && old[i] >= 0 && !moved[i]) { key = old[i]; moved[i] = 1; }
            else break;
        }
    }
    free(moved); free(old);
    t->slot = ns; t->n_buckets = nb; t->n_occupied = t->size; t->upper = (int)(nb * 0.77 + 0.5);
}

static int cbed_get(cbed_t *t, const char *name)
{
    for (int i = 0; i < t->n_chr; i++) if (strcmp(t->chr[i].name, name) == 0) return i;
    if (t->n_occupied >= t->upper) cbed_resize(t, t->n_buckets > (t->size << 1) ? t->n_buckets - 1 : t->n_buckets + 1);
    t->chr = (cbed_chr_t *)realloc(t->chr, sizeof(cbed_chr_t) * (size_t)(t->n_chr + 1));
    memset(&t->chr[t->n_chr], 0, sizeof(cbed_chr_t));
    t->chr[t->n_chr].name = strdup(name);
    const uint32_t mask = (uint32_t)t->n_buckets - 1;
    uint32_t i = fnv1a(name) & mask, step = 0;
    while (t->slot[i] >= 0) i = (i + (++step)) & mask;
    t->slot[i] = t->n_chr;
    t->size++; t->n_occupied++;
    return t->n_chr++;
}

static cbed_t *cbed_read(const char *fn)
{
    gzFile fp = gzopen(fn, "r");
    if (!fp) return NULL;
    cbed_t *t = (cbed_t *)calloc(1, sizeof(*t));
    char line[65536];
    while (gzgets(fp, line, sizeof(line))) {
        char *ref = line;
        while (*ref && isspace((unsigned char)*ref)) ref++;
        if (!*ref || *ref == '#') continue;
        char *e = ref; while (*e && !isspace((unsigned char)*e)) e++;
        unsigned long long beg = 0, end = 0; int num = 0;
        if (*e) { *e = 0; num = sscanf(e + 1, "%llu %llu", &beg, &end); }
        if (num == 1) end = beg--;
        if (num < 1 || end < beg) {
            if (!strcmp(ref, "browser") || !strcmp(ref, "track")) continue;
            fprintf(stderr, "[bed_read] Parse error reading \"%s\"\n", fn);
            gzclose(fp); return NULL;
        }
        const int ci = cbed_get(t, ref);
        cbed_chr_t *c = &t->chr[ci];
        if (c->n == c->m) { c->m = c->m ? c->m << 1 : 4; c->beg = (hpos_t *)realloc(c->beg, sizeof(hpos_t) * (size_t)c->m); c->end = (hpos_t *)realloc(c->end, sizeof(hpos_t) * (size_t)c->m); }
        c->beg[c->n] = (hpos_t)beg; c->end[c->n++] = (hpos_t)end;
    }
    gzclose(fp);
    for (int k = 0; k < t->n_chr; k++) {             /* total order, so any sort gives the reference's result */
        cbed_chr_t *c = &t->chr[k];
        for (int i = 1; i < c->n; i++)
            for (int j = i; j > 0 && (c->beg[j] < c->beg[j - 1] || (c->beg[j] == c->beg[j - 1] && c->end[j] < c->end[j - 1])); j--) {
                hpos_t x = c->beg[j]; c->beg[j] = c->beg[j - 1]; c->beg[j - 1] = x;
                x = c->end[j]; c->end[j] = c->end[j - 1]; c->end[j - 1] = x;
            }
    }
    return t;
}

/* ---------------- serial driver (bam_consensus.c:2898-3075) ---------------- */
static int run_serial(copts_t *o, const char *fn)
{
    cctx_t c; memset(&c, 0, sizeof(c));
    c.o = o; c.last_tid = -1; c.last_pos = -1; c.ref_tid = -1;
    cbed_t *bed = NULL;
    int n_iv = 0, iv = 0, *iv_tid = NULL; hpos_t *iv_beg = NULL, *iv_end = NULL;
    int ret = -1;

    if (o->bed_fn) {
        bed = cbed_read(o->bed_fn);
        if (!bed) { fprintf(stderr, "samtools consensus: Could not read file \"%s\"\n", o->bed_fn); return -1; }
        for (int bkt = 0; bkt < bed->n_buckets; bkt++) {
            if (bed->slot[bkt] < 0) continue;
            const cbed_chr_t *ch = &bed->chr[bed->slot[bkt]];
            int tid = hdr_name2tid(o->h, ch->name);
            if (tid < 0) { fprintf(stderr, "[W::fill_reglist_tid] Region '%s' specifies an unknown reference name\n", ch->name); continue; }
            for (int k = 0; k < ch->n; k++) {
                iv_tid = (int *)realloc(iv_tid, sizeof(int) * (size_t)(n_iv + 1));
                iv_beg = (hpos_t *)realloc(iv_beg, sizeof(hpos_t) * (size_t)(n_iv + 1));
                iv_end = (hpos_t *)realloc(iv_end, sizeof(hpos_t) * (size_t)(n_iv + 1));
                iv_tid[n_iv] = tid; iv_beg[n_iv] = ch->beg[k]; iv_end[n_iv++] = ch->end[k];
            }
        }
        if (n_iv <= 0) return -1;
    } else if (o->reg) {
        int t; hpos_t bb, ee;
        if (parse_region(o->h, o->reg, &t, &bb, &ee) < 0) { fprintf(stderr, "samtools consensus: Failed to parse region \"%s\"\n", o->reg); return -1; }
        c.has_iter = 1; c.iter_tid = t; c.iter_beg = bb; c.iter_end = ee;
    }

    do {
        if (bed) {
            os_clear(&c.row); os_clear(&c.seq); os_clear(&c.qual);
            c.last_tid = -1; c.last_pos = -1; c.ref_tid = -1; c.has_iter = 0;
            if (iv >= n_iv) break;
            int chr = iv_tid[iv]; hpos_t start = iv_beg[iv], end = iv_end[iv]; iv++;
            if (start > end || start > o->h->len[chr]) {
                fprintf(stderr, "[consensus] Warning: Invalid region \"%s:%lld-%lld\"\n", o->h->name[chr], (long long)start, (long long)end);
                continue;
            }
            if (start < 0) start = 0;
            if (end > o->h->len[chr]) end = o->h->len[chr];
            c.has_iter = 1; c.iter_tid = chr; c.iter_beg = start; c.iter_end = end;
            c.last_pos = start;
        }
        c.rd = rd_open(fn);
        if (!c.rd) goto err;
        (void)rd_header(c.rd);
        if (c.has_iter) rd_set_region(c.rd, c.iter_tid, c.iter_beg, c.iter_end);
        int lr = column_loop(&c);
        rd_close(c.rd); c.rd = NULL;
        if (lr < 0) goto err;

        if (o->fmt == FMT_DUMP) {
        } else if (o->fmt == FMT_PILEUP) {
            if (o->all_bases) {
                int tid = c.has_iter ? c.iter_tid : c.last_tid;
                hpos_t len = tid >= 0 && tid < o->h->n_ref ? o->h->len[tid] : 0, pos = c.last_pos;
                if (c.has_iter) { if (c.iter_end < len) len = c.iter_end; if (c.iter_beg > pos) pos = c.iter_beg; }
                if (tid >= 0) empty_rows(&c, tid, pos, len);
            }
            while (!c.has_iter && o->all_bases > 1 && ++c.last_tid < o->h->n_ref) empty_rows(&c, c.last_tid, 0, (int)o->h->len[c.last_tid]);
        } else {
            for (;;) {
                if (o->all_bases) {
                    int tid = c.has_iter ? c.iter_tid : c.last_tid;
                    hpos_t len = tid >= 0 && tid < o->h->n_ref ? o->h->len[tid] : 0, pos = c.last_pos;
                    if (c.has_iter) { if (c.iter_end < len) len = c.iter_end; if (c.iter_beg > pos) pos = c.iter_beg; c.last_tid = c.iter_tid; }
                    if (pos < len) {
                        if (update_ref(&c, c.last_tid) < 0) goto err;
                        fill_flat(&c, pos, len - pos);
                    }
                }
                if (c.last_tid >= 0) {
                    char name[1024];
                    int tid = c.has_iter ? c.iter_tid : c.last_tid;
                    int len = (int)o->h->len[tid];
                    hpos_t e = c.iter_end < len ? c.iter_end : len;
                    if (c.has_iter && (c.iter_beg > 0 || c.iter_end < len)) snprintf(name, sizeof(name), "%s:%lld-%lld", o->h->name[c.last_tid], (long long)c.iter_beg + 1, (long long)e);
                    else snprintf(name, sizeof(name), "%s", o->h->name[c.last_tid]);
                    dump_fastq(o, name, &c.seq, &c.qual);
                }
                if (!c.has_iter && o->all_bases > 1 && ++c.last_tid < o->h->n_ref) { c.last_pos = 0; os_clear(&c.seq); os_clear(&c.qual); continue; }
                break;
            }
        }
    } while (iv < n_iv);
    ret = 0;
err:
    free(c.row.s); free(c.seq.s); free(c.qual.s);
    free(iv_tid); free(iv_beg); free(iv_end);
    return ret;
}

/* bam_consensus.c:674-738 (named tables other than :flat are not restated) */
#include "o_qcal_tables.inc"      /* the platform tables as data (scripts/gen_qcal_tables.py; bam_consensus.c:446-662) */

/* bam_consensus.c:664-670 */
static int set_qcal(qcal_t *q, int id)
{
    if (id < 0 || id >= 6) return -1;
    memcpy(q->smap, QCAL_TABLES[id][0], sizeof q->smap); memcpy(q->umap, QCAL_TABLES[id][1], sizeof q->umap); memcpy(q->omap, QCAL_TABLES[id][2], sizeof q->omap);
    return 0;
}

static int load_qcal(qcal_t *q, const char *fn)
{
    for (int id = 1; id < 6; id++) if (fn[0] == ':' && strcmp(fn + 1, QCAL_NAMES[id]) == 0) return set_qcal(q, id);      /* :672-686 */
    for (int i = 0; i < 101; i++) q->smap[i] = q->umap[i] = q->omap[i] = i;
    if (strcmp(fn, ":flat") == 0) return 0;
    if (fn[0] == ':') { fprintf(stderr, "oracle consensus: calibration table %s is not restated\n", fn); return -1; }
    FILE *fp = fopen(fn, "r");
    if (!fp) return -1;
    char line[1024];
    int max = 0, last_qual = 0;
    while (fgets(line, sizeof(line), fp)) {
        int v, s, u, ov;
        if (*line == '#') continue;
        if (sscanf(line, "QUAL %d %d %d %d", &v, &s, &u, &ov) != 4) { fclose(fp); return -1; }
        while (v > last_qual && last_qual < 100) {
            q->smap[last_qual + 1] = q->smap[last_qual]; q->umap[last_qual + 1] = q->umap[last_qual]; q->omap[last_qual + 1] = q->omap[last_qual];
            last_qual++;
        }
        if (v >= 0 && v < 100) { q->smap[v] = s; q->umap[v] = u; q->omap[v] = ov; }
        if (v < max) { fprintf(stderr, "Qual calibration file is not in ascending order\n"); fclose(fp); return -1; }
        max = v;
    }
    for (int i = max + 1; i < 101; i++) { q->smap[i] = q->smap[max]; q->umap[i] = q->umap[max]; q->omap[i] = q->omap[max]; }
    fclose(fp);
    return 0;
}

/* bam_consensus.c:3149-3593 */
int o_main_consensus(int argc, char *argv[])
{
    copts_t o; memset(&o, 0, sizeof(o));
    o.mode = MODE_RECALL; o.adj_qual = 1; o.use_mqual = 1; o.scale_mqual = 1.00; o.nm_adjust = 1; o.nm_halo = 50; o.sc_cost = 60;
    o.low_mqual = 1; o.high_mqual = 60; o.min_depth = 1; o.call_fract = 0.75; o.het_fract = 0.5; o.fmt = FMT_FASTA; o.cons_cutoff = 10;
    o.line_len = 70; o.default_qual = 10; o.show_ins = 1; o.excl_flags = F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP;
    o.P_het = 1e-3; o.P_indel = 2e-4; o.het_scale = 1.0; o.homopoly_redux = 0.01; o.out = stdout;
    set_qcal(&o.qcal, 0);                   /* bam_consensus.c:3196 */

    static const struct option lopts[] = {
        { "use-qual", no_argument, NULL, 'q' }, { "no-use-qual", no_argument, NULL, 'q' + 1000 }, { "adj-qual", no_argument, NULL, 'q' + 100 },
        { "no-adj-qual", no_argument, NULL, 'q' + 101 }, { "use-MQ", no_argument, NULL, 'm' + 1000 }, { "no-use-MQ", no_argument, NULL, 'm' + 1001 },
        { "adj-MQ", no_argument, NULL, 'm' + 100 }, { "no-adj-MQ", no_argument, NULL, 'm' + 101 }, { "NM-halo", required_argument, NULL, 'h' + 100 },
        { "SC-cost", required_argument, NULL, 'h' + 101 }, { "scale-MQ", required_argument, NULL, 14 }, { "low-MQ", required_argument, NULL, 9 },
        { "high-MQ", required_argument, NULL, 10 }, { "min-depth", required_argument, NULL, 'd' }, { "call-fract", required_argument, NULL, 'c' },
        { "het-fract", required_argument, NULL, 'H' }, { "region", required_argument, NULL, 'r' }, { "regions-file", required_argument, NULL, 'r' + 1000 },
        { "format", required_argument, NULL, 'f' }, { "cutoff", required_argument, NULL, 'C' }, { "ambig", no_argument, NULL, 'A' },
        { "line-len", required_argument, NULL, 'l' }, { "default-qual", required_argument, NULL, 1 }, { "het-only", no_argument, NULL, 6 },
        { "show-del", required_argument, NULL, 7 }, { "show-ins", required_argument, NULL, 8 }, { "mark-ins", no_argument, NULL, 18 },
        { "output", required_argument, NULL, 'o' }, { "incl-flags", required_argument, NULL, 11 }, { "rf", required_argument, NULL, 11 },
        { "excl-flags", required_argument, NULL, 12 }, { "ff", required_argument, NULL, 12 }, { "min-MQ", required_argument, NULL, 13 },
        { "min-BQ", required_argument, NULL, 16 }, { "P-het", required_argument, NULL, 15 }, { "P-indel", required_argument, NULL, 17 },
        { "het-scale", required_argument, NULL, 19 }, { "mode", required_argument, NULL, 'm' }, { "homopoly-fix", no_argument, NULL, 'p' },
        { "homopoly-score", required_argument, NULL, 'p' + 100 }, { "homopoly-redux", required_argument, NULL, 'p' + 200 },
        { "qual-calibration", required_argument, NULL, 't' }, { "config", required_argument, NULL, 'X' }, { "ref-qual", required_argument, NULL, 20 },
        { "block-size", required_argument, NULL, 'Z' }, { "reference", required_argument, NULL, 'T' }, { "threads", required_argument, NULL, '@' },
        { NULL, 0, NULL, 0 } };
    int c;
    optind = 1;
    while ((c = getopt_long(argc, argv, "@:qd:c:H:r:5f:C:aAl:o:m:pt:X:T:Z:", lopts, NULL)) >= 0) {
        switch (c) {
        case 'a': o.all_bases++; break;
        case 'q': o.use_qual = 1; break;
        case 'q' + 1000: o.use_qual = 0; break;
        case 'm' + 1000: o.use_mqual = 1; break;
        case 'm' + 1001: o.use_mqual = 0; break;
        case 14: o.scale_mqual = atof(optarg); break;
        case 9: o.low_mqual = atoi(optarg); break;
        case 10: o.high_mqual = atoi(optarg); break;
        case 'd': o.min_depth = atoi(optarg); break;
        case 'c': o.call_fract = atof(optarg); break;
        case 'H': o.het_fract = atof(optarg); break;
        case 'r':
            if (o.bed_fn) { fprintf(stderr, "samtools consensus: option -r and --regions-file are incompatible\n"); return 1; }
            o.reg = optarg; break;
        case 'r' + 1000:
            if (o.reg) { fprintf(stderr, "samtools consensus: option -r and --regions-file are incompatible\n"); return 1; }
            o.bed_fn = optarg; break;
        case 'C': o.cons_cutoff = atoi(optarg); break;
        case 'A': o.ambig = 1; break;
        case 'p': o.homopoly_fix = 0.5; break;
        case 'p' + 100: o.homopoly_fix = atof(optarg); break;
        case 'p' + 200: o.homopoly_redux = atof(optarg); break;
        case 1: o.default_qual = atoi(optarg); break;
        case 6: break;
        case 7: o.show_del = (*optarg == 'y' || *optarg == 'Y'); break;
        case 8: o.show_ins = (*optarg == 'y' || *optarg == 'Y'); break;
        case 18: o.mark_ins = 1; break;
        case 13: o.min_mqual = atoi(optarg); break;
        case 16: o.min_qual = atoi(optarg); break;
        case 15: o.P_het = atof(optarg); break;
        case 17: o.P_indel = atof(optarg); break;
        case 19: o.het_scale = atof(optarg); break;
        case 'q' + 100: o.adj_qual = 1; break;
        case 'q' + 101: o.adj_qual = 0; break;
        case 'm' + 100: o.nm_adjust = 1; break;
        case 'm' + 101: o.nm_adjust = 0; break;
        case 'h' + 100: o.nm_halo = atoi(optarg); break;
        case 'h' + 101: o.sc_cost = atoi(optarg); break;
        case 'Z': case '@': break;
        case 'm':
            if (!strcasecmp(optarg, "simple")) o.mode = MODE_SIMPLE;
            else if (!strcasecmp(optarg, "bayesian_m")) o.mode = MODE_MIXED;
            else if (!strcasecmp(optarg, "bayesian_p")) o.mode = MODE_PRECISE;
            else if (!strcasecmp(optarg, "bayesian_r") || !strcasecmp(optarg, "bayesian")) o.mode = MODE_RECALL;
            else if (!strcasecmp(optarg, "bayesian_116")) o.mode = MODE_BAYES_116;
            else { fprintf(stderr, "Unknown mode %s\n", optarg); return 1; }
            break;
        case 'l': if ((o.line_len = atoi(optarg)) <= 0) o.line_len = INT_MAX; break;
        case 'f':
            if (!strcasecmp(optarg, "fasta")) o.fmt = FMT_FASTA;
            else if (!strcasecmp(optarg, "fastq")) o.fmt = FMT_FASTQ;
            else if (!strcasecmp(optarg, "pileup")) o.fmt = FMT_PILEUP;
            else if (!strcasecmp(optarg, "dump")) o.fmt = FMT_DUMP;
            else { fprintf(stderr, "Unknown format %s\n", optarg); return 1; }
            break;
        case 'o': if (!(o.out = fopen(optarg, "w"))) { perror(optarg); return 1; } break;
        case 'X':      /* bam_consensus.c:3366-3421 */
            if (strcasecmp(optarg, "hifi") == 0) {
                set_qcal(&o.qcal, 1); o.mode = MODE_RECALL; o.homopoly_fix = 0.3; o.homopoly_redux = 0.01; o.low_mqual = 5; o.scale_mqual = 1.5; o.het_scale = 0.37;
            } else if (strcasecmp(optarg, "hiseq") == 0) {
                o.mode = MODE_RECALL; set_qcal(&o.qcal, 2); o.homopoly_redux = 0.01;
            } else if (strcasecmp(optarg, "r10.4_sup") == 0) {
                set_qcal(&o.qcal, 3); o.mode = MODE_RECALL; o.homopoly_fix = 0.3; o.homopoly_redux = 0.01; o.low_mqual = 5; o.scale_mqual = 1.5; o.het_scale = 0.37;
            } else if (strcasecmp(optarg, "r10.4_dup") == 0) {
                set_qcal(&o.qcal, 4); o.mode = MODE_RECALL; o.homopoly_fix = 0.3; o.homopoly_redux = 0.01; o.low_mqual = 5; o.scale_mqual = 1.5; o.het_scale = 0.37;
            } else if (strcasecmp(optarg, "ultima") == 0) {
                o.mode = MODE_RECALL; set_qcal(&o.qcal, 5); o.homopoly_fix = 0.3; o.homopoly_redux = 0.01; o.het_scale = 0.37; o.scale_mqual = 2; o.low_mqual = 10;
            } else { fprintf(stderr, "Unrecognised configuration name: \"%s\"\n", optarg); return 1; }
            break;
        case 11: if ((o.incl_flags = str2flag(optarg)) < 0) { fprintf(stderr, "samtools consensus: could not parse --rf %s\n", optarg); return 1; } break;
        case 12: if ((o.excl_flags = str2flag(optarg)) < 0) { fprintf(stderr, "samtools consensus: could not parse --ff %s\n", optarg); return 1; } break;
        case 't': if (load_qcal(&o.qcal, optarg) < 0) { fprintf(stderr, "samtools consensus: failed to load quality calibration '%s'\n", optarg); return 1; } break;
        case 'T': o.ref_fn = optarg; break;
        case 20: o.ref_qual = atoi(optarg); break;
        default: fprintf(stderr, "Usage: oracle_samtools consensus [options] <in.bam>\n"); return 1;
        }
    }
    static_tables();
    if (o.mode != MODE_SIMPLE) {
        if (o.mode == MODE_PRECISE) cons_init(o.P_het, o.P_indel, 0.3 * o.het_scale, o.homopoly_redux, &o.qcal, MODE_PRECISE, &cp_precise);
        if (o.mode == MODE_MIXED) cons_init(pow(o.P_het, 0.7), pow(o.P_indel, 0.7), 0.3 * o.het_scale, o.homopoly_redux, &o.qcal, MODE_PRECISE, &cp_precise);
        cons_init(o.P_het, o.P_indel, o.het_scale, o.mode == MODE_RECALL ? o.homopoly_redux : 0.01, &o.qcal, MODE_RECALL, &cp_recall);
    }
    if (argc != optind + 1) { fprintf(stderr, "Usage: oracle_samtools consensus [options] <in.bam>\n"); return argc == optind ? 0 : 1; }
    const char *fn = argv[optind];
    oreader_t *r0 = rd_open(fn);
    if (!r0) { fprintf(stderr, "samtools consensus: Cannot open input file \"%s\"\n", fn); return 1; }
    o.h = rd_header(r0);
    if (!o.h) { fprintf(stderr, "Failed to read header for \"%s\"\n", fn); return 1; }
    if (o.ref_fn && !(o.fa = fa_load(o.ref_fn))) { fprintf(stderr, "Failed to load fai for %s\n", o.ref_fn); return 1; }
    int ret = run_serial(&o, fn) < 0 ? 1 : 0;
    rd_close(r0);
    if (o.out != stdout) ret |= fclose(o.out) != 0; else ret |= fflush(stdout) != 0;
    if (ret) fprintf(stderr, "samtools consensus: failed\n");
    return ret;
}
