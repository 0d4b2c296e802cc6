/*
 * oracle/o_plp.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of the HTSlib 1.23.1 pileup engine (sam.c: bam_plp_push,
 * bam_plp64_next, bam_plp64_auto, resolve_cigar2, overlap_push,
 * tweak_overlap_quality, cigar_iref2iseq_set/next, bam_plp_insertion,
 * bam_mplp64_auto).  That file is NOT in /root/reference (un-vendored
 * dependency, pinned HTSlib 1.23.1: NEWS.md:9); behaviour restated from
 * SURVEY.md Appendix A.1-A.3 and pinned by the reference goldens
 * test/mpileup/expected/{mp_*,13,23,24,47,78,mp1_*}.out, test/dat/mpileup.out.{1,5}.
 * Reference call sites: bam_plcmd.c:581-607, bam_plbuf.c:40-66.
 */
#include "o_plp.h"
#include <assert.h>

/* ------------------------------------------------------------------ */
/* list node + memory pool (lbnode_t / mempool_t)                      */
typedef struct { int k; hpos_t x, y, end; } cstate_t;
typedef struct lbnode {
    orec_t b;
    hpos_t beg, end;
    cstate_t s;
    struct lbnode *next;
} lbnode_t;

typedef struct { int cnt, n, max; lbnode_t **buf; } mempool_t;

static lbnode_t *mp_alloc(mempool_t *mp)
{
    ++mp->cnt;
    if (mp->n == 0) return (lbnode_t *)calloc(1, sizeof(lbnode_t));
    return mp->buf[--mp->n];
}
static void mp_free(mempool_t *mp, lbnode_t *p)
{
    --mp->cnt; p->next = 0;
    if (mp->n == mp->max) {
        mp->max = mp->max ? mp->max << 1 : 256;
        mp->buf = (lbnode_t **)realloc(mp->buf, sizeof(lbnode_t *) * (size_t)mp->max);
    }
    mp->buf[mp->n++] = p;
}

/* ------------------------------------------------------------------ */
/* qname -> node hash for overlap detection (khash olap_hash stand-in) */
typedef struct oent { char *key; lbnode_t *val; struct oent *next; } oent_t;
typedef struct { oent_t **tab; size_t nb; size_t n; } ohash_t;

/* khash.h __ac_X31_hash_string / __ac_Wang_hash (32-bit) */
static uint32_t x31_hash(const char *s)
{
    uint32_t h = (uint32_t)(unsigned char)*s;
    if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)(unsigned char)*s;
    return h;
}
static uint32_t wang_hash(uint32_t key)
{
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}
static ohash_t *oh_init(void)
{
    ohash_t *h = (ohash_t *)calloc(1, sizeof(*h));
    h->nb = 1024; h->tab = (oent_t **)calloc(h->nb, sizeof(oent_t *));
    return h;
}
static void oh_grow(ohash_t *h)
{
    size_t nb = h->nb * 4;
    oent_t **t = (oent_t **)calloc(nb, sizeof(oent_t *));
    for (size_t i = 0; i < h->nb; ++i)
        for (oent_t *e = h->tab[i], *nx; e; e = nx) {
            nx = e->next;
            size_t j = x31_hash(e->key) & (nb - 1);
            e->next = t[j]; t[j] = e;
        }
    free(h->tab); h->tab = t; h->nb = nb;
}
static oent_t **oh_find(ohash_t *h, const char *key)
{
    oent_t **pp = &h->tab[x31_hash(key) & (h->nb - 1)];
    while (*pp && strcmp((*pp)->key, key)) pp = &(*pp)->next;
    return pp;
}
static void oh_put(ohash_t *h, const char *key, lbnode_t *val)
{
    if (h->n > h->nb) oh_grow(h);
    oent_t *e = (oent_t *)malloc(sizeof(*e));
    size_t j = x31_hash(key) & (h->nb - 1);
    e->key = strdup(key); e->val = val; e->next = h->tab[j]; h->tab[j] = e; h->n++;
}
static void oh_del(ohash_t *h, oent_t **pp)
{
    oent_t *e = *pp; *pp = e->next; free(e->key); free(e); h->n--;
}
static void oh_destroy(ohash_t *h)
{
    if (!h) return;
    for (size_t i = 0; i < h->nb; ++i)
        for (oent_t *e = h->tab[i], *nx; e; e = nx) { nx = e->next; free(e->key); free(e); }
    free(h->tab); free(h);
}

/* ------------------------------------------------------------------ */
struct oplp {
    mempool_t mp;
    lbnode_t *head, *tail;
    int32_t tid, max_tid;
    hpos_t pos, max_pos;
    int is_eof, max_plp, error, maxcnt;
    opileup1_t *plp;
    orec_t b;
    oplp_auto_f func;
    void *data;
    ohash_t *overlaps;
    int (*construct)(void *data, const orec_t *b, void *cd);   /* bam_plp_constructor hook (cd is not modelled: NULL) */
};

/* ---- cigar_iref2iseq_set / _next (Appendix A.3.1) ---- */
typedef struct { const uint32_t *cig, *cig_max; hpos_t icig, iseq, iref; } cwalk_t;

static int iref2iseq_set(cwalk_t *w, hpos_t pos)
{
    if (pos < 0) return -1;
    w->icig = 0; w->iseq = 0; w->iref = 0;
    while (w->cig < w->cig_max) {
        int op = cig_op(*w->cig); hpos_t n = cig_len(*w->cig);
        if (op == C_S) { w->cig++; w->iseq += n; w->icig = 0; continue; }
        if (op == C_H || op == C_P) { w->cig++; w->icig = 0; continue; }
        if (op == C_M || op == C_EQ || op == C_X) {
            pos -= n;
            if (pos < 0) { w->icig = n + pos; w->iseq += w->icig; w->iref += w->icig; return 0; }
            w->cig++; w->iseq += n; w->icig = 0; w->iref += n;
            continue;
        }
        if (op == C_I) { w->cig++; w->iseq += n; w->icig = 0; continue; }
        if (op == C_D || op == C_N) {
            pos -= n; if (pos < 0) pos = 0;
            w->cig++; w->icig = 0; w->iref += n;
            continue;
        }
        return -2;
    }
    w->iseq = -1;
    return -1;
}
static int iref2iseq_next(cwalk_t *w)
{
    while (w->cig < w->cig_max) {
        int op = cig_op(*w->cig); hpos_t n = cig_len(*w->cig);
        if (op == C_M || op == C_EQ || op == C_X) {
            if (w->icig >= n - 1) { w->icig = -1; w->cig++; continue; }
            w->iseq++; w->icig++; w->iref++;
            return 0;
        }
        if (op == C_D || op == C_N) { w->cig++; w->iref += n; w->icig = -1; continue; }
        if (op == C_I) { w->cig++; w->iseq += n; w->icig = -1; continue; }
        if (op == C_S) { w->cig++; w->iseq += n; w->icig = -1; continue; }
        if (op == C_H || op == C_P) { w->cig++; w->icig = -1; continue; }
        return -2;
    }
    w->iseq = -1; w->iref = -1;
    return -1;
}

/* tweak_overlap_quality (Appendix A.3 / A.3.1) */
static int tweak_overlap_quality(orec_t *a, orec_t *b)
{
    cwalk_t wa = { a->cigar, a->cigar + a->n_cigar, 0, 0, 0 };
    cwalk_t wb = { b->cigar, b->cigar + b->n_cigar, 0, 0, 0 };
    uint8_t *a_qual = a->qual, *b_qual = b->qual;
    hpos_t iref = b->pos;
    int a_ret = iref2iseq_set(&wa, iref - a->pos);
    if (a_ret < 0) return a_ret < -1 ? -1 : 0;
    int b_ret = iref2iseq_set(&wb, iref - b->pos);
    if (b_ret < 0) return b_ret < -1 ? -1 : 0;

    int amul, bmul;
    if (wang_hash(x31_hash(a->qname)) & 1) { amul = 1; bmul = 0; }
    else { amul = 0; bmul = 1; }

    int err = 0;
    for (;;) {
        while (a_ret >= 0 && wa.iref >= 0 && wa.iref < iref - a->pos) a_ret = iref2iseq_next(&wa);
        if (a_ret < 0) { err = a_ret < -1 ? -1 : 0; break; }
        if (iref < wa.iref + a->pos) iref = wa.iref + a->pos;

        while (b_ret >= 0 && wb.iref >= 0 && wb.iref < iref - b->pos) b_ret = iref2iseq_next(&wb);
        if (b_ret < 0) { err = b_ret < -1 ? -1 : 0; break; }
        if (iref < wb.iref + b->pos) iref = wb.iref + b->pos;

        iref++;

        if (wa.iref + a->pos != wb.iref + b->pos) {
            if (wa.iref + a->pos < wb.iref + b->pos
                && wb.cig > b->cigar && cig_op(*(wb.cig - 1)) == C_D) {
                do {
                    a_qual[wa.iseq] = amul ? (uint8_t)(a_qual[wa.iseq] * 0.8) : 0;
                    a_ret = iref2iseq_next(&wa);
                    if (a_ret < 0) return -(a_ret < -1);
                } while (wa.iref + a->pos < wb.iref + b->pos);
            } else if (wa.cig > a->cigar && cig_op(*(wa.cig - 1)) == C_D) {
                do {
                    b_qual[wb.iseq] = bmul ? (uint8_t)(b_qual[wb.iseq] * 0.8) : 0;
                    b_ret = iref2iseq_next(&wb);
                    if (b_ret < 0) return -(b_ret < -1);
                } while (wb.iref + b->pos < wa.iref + a->pos);
            } else {
                continue;
            }
        }

        if (wa.iseq > a->l_qseq || wb.iseq > b->l_qseq) return -1;

        if (rec_seqi(a->seq, wa.iseq) == rec_seqi(b->seq, wb.iseq)) {
            int qual = a_qual[wa.iseq] + b_qual[wb.iseq];
            a_qual[wa.iseq] = (uint8_t)(amul * (qual > 200 ? 200 : qual));
            b_qual[wb.iseq] = (uint8_t)(bmul * (qual > 200 ? 200 : qual));
        } else {
            if (a_qual[wa.iseq] > b_qual[wb.iseq]) {
                a_qual[wa.iseq] = (uint8_t)(0.8 * a_qual[wa.iseq]);
                b_qual[wb.iseq] = 0;
            } else if (a_qual[wa.iseq] < b_qual[wb.iseq]) {
                b_qual[wb.iseq] = (uint8_t)(0.8 * b_qual[wb.iseq]);
                a_qual[wa.iseq] = 0;
            } else {
                a_qual[wa.iseq] = (uint8_t)(amul * 0.8 * a_qual[wa.iseq]);
                b_qual[wb.iseq] = (uint8_t)(bmul * 0.8 * b_qual[wb.iseq]);
            }
        }
    }
    return err;
}

static void overlap_remove(oplp_t *iter, const orec_t *b)
{
    if (!iter->overlaps || !b) return;
    oent_t **pp = oh_find(iter->overlaps, b->qname);
    if (*pp) oh_del(iter->overlaps, pp);
}

static int overlap_push(oplp_t *iter, lbnode_t *node)
{
    if (!iter->overlaps) return 0;
    if ((node->b.flag & F_MUNMAP) || !(node->b.flag & F_PROPER_PAIR)) return 0;
    if ((node->b.mtid >= 0 && node->b.tid != node->b.mtid)
        || (llabs(node->b.isize) >= 2 * (long long)node->b.l_qseq && node->b.mpos >= node->end))
        return 0;
    oent_t **pp = oh_find(iter->overlaps, node->b.qname);
    if (!*pp) {
        if (node->b.mpos >= node->b.pos || ((node->b.flag & F_PAIRED) && node->b.mpos == -1))
            oh_put(iter->overlaps, node->b.qname, node);
    } else {
        lbnode_t *a = (*pp)->val;
        int err = tweak_overlap_quality(&a->b, &node->b);
        oh_del(iter->overlaps, pp);
        return err;
    }
    return 0;
}

/* ---- resolve_cigar2 (Appendix A.2) ---- */
static int is_refop(int op) { return op == C_M || op == C_D || op == C_N || op == C_EQ || op == C_X; }
static int is_mop(int op) { return op == C_M || op == C_EQ || op == C_X; }

static int resolve_cigar2(opileup1_t *p, hpos_t pos, cstate_t *s)
{
    orec_t *b = p->b;
    const uint32_t *cigar = b->cigar;
    int n_cigar = (int)b->n_cigar, k;
    if (s->k == -1) {
        p->qpos = 0;
        if (n_cigar == 1) {
            if (is_mop(cig_op(cigar[0]))) s->k = 0, s->x = b->pos, s->y = 0;
        } else {
            for (k = 0, s->x = b->pos, s->y = 0; k < n_cigar; ++k) {
                int op = cig_op(cigar[k]); int l = (int)cig_len(cigar[k]);
                if (is_refop(op)) break;
                else if (op == C_I || op == C_S) s->y += l;
            }
            assert(k < n_cigar);
            s->k = k;
        }
    } else {
        int op, l = (int)cig_len(cigar[s->k]);
        if (pos - s->x >= l) {
            assert(s->k < n_cigar);
            if (is_mop(cig_op(cigar[s->k]))) s->y += l;
            s->x += l;
            for (k = s->k + 1; k < n_cigar; ++k) {
                op = cig_op(cigar[k]); l = (int)cig_len(cigar[k]);
                if (is_refop(op)) break;
                else if (op == C_I || op == C_S) s->y += l;
            }
            s->k = k;
            assert(s->k < n_cigar);
        }
    }
    {
        int op = cig_op(cigar[s->k]), l = (int)cig_len(cigar[s->k]);
        p->is_del = p->indel = p->is_refskip = 0;
        if (s->x + l - 1 == pos && s->k + 1 < n_cigar) {
            int op2 = cig_op(cigar[s->k + 1]);
            int l2 = (int)cig_len(cigar[s->k + 1]);
            if (op2 == C_D && op != C_D) {
                p->indel = -l2;
                for (k = s->k + 2; k < n_cigar; ++k) {
                    op2 = cig_op(cigar[k]); l2 = (int)cig_len(cigar[k]);
                    if (op2 == C_D) p->indel -= l2; else break;
                }
            } else if (op2 == C_I) {
                p->indel = l2;
                for (k = s->k + 2; k < n_cigar; ++k) {
                    op2 = cig_op(cigar[k]); l2 = (int)cig_len(cigar[k]);
                    if (op2 == C_I) p->indel += l2;
                    else if (op2 != C_P) break;
                }
            } else if (op2 == C_P && s->k + 2 < n_cigar) {
                int l3 = 0;
                for (k = s->k + 2; k < n_cigar; ++k) {
                    op2 = cig_op(cigar[k]); l2 = (int)cig_len(cigar[k]);
                    if (op2 == C_I) l3 += l2;
                    else if (is_refop(op2)) break;
                }
                if (l3 > 0) p->indel = l3;
            }
        }
        if (is_mop(op)) {
            p->qpos = (int32_t)(s->y + (pos - s->x));
        } else if (op == C_D || op == C_N) {
            p->is_del = 1; p->qpos = (int32_t)s->y;
            p->is_refskip = (op == C_N);
        }
        p->is_head = (pos == b->pos); p->is_tail = (pos == s->end);
    }
    p->cigar_ind = s->k;
    return 1;
}

/* bam_plp_insertion (Appendix A.2, last paragraph) */
int oplp_insertion(const opileup1_t *p, ostr_t *ins, int *del_len) { return oplp_insertion_mod(p, NULL, ins, del_len); }

int oplp_insertion_mod(const opileup1_t *p, const omods_t *m, ostr_t *ins, int *del_len)
{
    int j, k, indel;
    os_clear(ins);
    if (del_len) *del_len = 0;
    if (p->indel <= 0) return 0;
    const uint32_t *cigar = p->b->cigar;
    int n_cigar = (int)p->b->n_cigar;
    /* measure: sum of I and P following cigar_ind */
    indel = 0; k = p->cigar_ind + 1;
    while (k < n_cigar) {
        int op = cig_op(cigar[k]);
        if (op == C_P || op == C_I) indel += (int)cig_len(cigar[k]); else break;
        k++;
    }
    /* produce sequence */
    k = p->cigar_ind + 1; j = 1;
    while (k < n_cigar) {
        int op = cig_op(cigar[k]); int l = (int)cig_len(cigar[k]);
        if (op == C_P) {
            for (int c = 0; c < l; ++c) os_putc(ins, '*');
        } else if (op == C_I) {
            for (int c = 0; c < l; ++c, ++j) {
                int qi = p->qpos + j - p->is_del;
                os_putc(ins, qi < p->b->l_qseq ? nt16_str[rec_seqi(p->b->seq, qi)] : 'N');
                if (m) omods_put(m, qi, ins);
            }
        } else break;
        k++;
    }
    if (k < n_cigar && cig_op(cigar[k]) == C_D && del_len) *del_len = (int)cig_len(cigar[k]);
    return indel;
}

/* ---- bam_plp_* ---- */
oplp_t *oplp_init(oplp_auto_f func, void *data)
{
    oplp_t *iter = (oplp_t *)calloc(1, sizeof(*iter));
    iter->head = iter->tail = mp_alloc(&iter->mp);
    iter->max_tid = -1; iter->max_pos = -1;
    iter->maxcnt = 8000;
    iter->func = func; iter->data = data;
    return iter;
}

void oplp_destroy(oplp_t *iter)
{
    if (!iter) return;
    lbnode_t *p, *pnext;
    for (p = iter->head; p; p = pnext) { pnext = p->next; rec_free(&p->b); free(p); }
    for (int i = 0; i < iter->mp.n; ++i) { rec_free(&iter->mp.buf[i]->b); free(iter->mp.buf[i]); }
    free(iter->mp.buf);
    oh_destroy(iter->overlaps);
    rec_free(&iter->b);
    free(iter->plp);
    free(iter);
}

void oplp_set_maxcnt(oplp_t *iter, int maxcnt) { iter->maxcnt = maxcnt; }
int oplp_init_overlaps(oplp_t *iter) { iter->overlaps = oh_init(); return 0; }

/* Appendix A.1 */
int oplp_push(oplp_t *iter, const orec_t *b)
{
    if (iter->error) return -1;
    if (b) {
        if (b->tid < 0) { overlap_remove(iter, b); return 0; }
        if (b->flag & F_UNMAP) { overlap_remove(iter, b); return 0; }
        if (iter->tid == b->tid && iter->pos == b->pos && iter->mp.cnt > iter->maxcnt) {
            overlap_remove(iter, b);
            return 0;
        }
        rec_copy(&iter->tail->b, b);
        iter->tail->beg = b->pos;
        iter->tail->end = b->pos + rec_rlen(b);
        iter->tail->s.k = -1; iter->tail->s.x = iter->tail->s.y = 0;
        iter->tail->s.end = iter->tail->end - 1;
        if (b->tid < iter->max_tid) {
            fprintf(stderr, "[E::bam_plp_push] The input is not sorted (chromosomes out of order)\n");
            iter->error = 1; return -1;
        }
        if (b->tid == iter->max_tid && iter->tail->beg < iter->max_pos) {
            fprintf(stderr, "[E::bam_plp_push] The input is not sorted (reads out of order)\n");
            iter->error = 1; return -1;
        }
        iter->max_tid = b->tid; iter->max_pos = iter->tail->beg;
        if (iter->tail->end > iter->pos || iter->tail->b.tid > iter->tid) {
            lbnode_t *next = mp_alloc(&iter->mp);
            if (iter->construct && iter->construct(iter->data, &iter->tail->b, NULL) < 0) { mp_free(&iter->mp, next); iter->error = 1; return -1; }
            if (overlap_push(iter, iter->tail) < 0) {
                mp_free(&iter->mp, next);
                iter->error = 1; return -1;
            }
            iter->tail->next = next;
            iter->tail = next;
        }
    } else iter->is_eof = 1;
    return 0;
}

/* Appendix A.2 */
const opileup1_t *oplp_next(oplp_t *iter, int *_tid, hpos_t *_pos, int *_n_plp)
{
    if (iter->error) { *_n_plp = -1; return NULL; }
    *_n_plp = 0;
    if (iter->is_eof && iter->head == iter->tail) return NULL;
    while (iter->is_eof || iter->max_tid > iter->tid || (iter->max_tid == iter->tid && iter->max_pos > iter->pos)) {
        int n_plp = 0;
        lbnode_t **pptr = &iter->head;
        while (*pptr != iter->tail) {
            lbnode_t *p = *pptr;
            if (p->b.tid < iter->tid || (p->b.tid == iter->tid && p->end <= iter->pos)) {
                overlap_remove(iter, &p->b);
                *pptr = p->next; mp_free(&iter->mp, p);
            } else {
                if (p->b.tid == iter->tid && p->beg <= iter->pos) {
                    if (n_plp == iter->max_plp) {
                        iter->max_plp = iter->max_plp ? iter->max_plp << 1 : 256;
                        iter->plp = (opileup1_t *)realloc(iter->plp, sizeof(opileup1_t) * (size_t)iter->max_plp);
                    }
                    iter->plp[n_plp].b = &p->b;
                    if (resolve_cigar2(iter->plp + n_plp, iter->pos, &p->s)) ++n_plp;
                }
                pptr = &(*pptr)->next;
            }
        }
        *_n_plp = n_plp; *_tid = iter->tid; *_pos = iter->pos;
        if (iter->head != iter->tail) {
            if (iter->tid > iter->head->b.tid) {
                fprintf(stderr, "[E::bam_plp64_next] Unsorted input. Pileup aborts\n");
                iter->error = 1; *_n_plp = -1;
                return NULL;
            }
        }
        if (iter->tid < iter->head->b.tid) {
            iter->tid = iter->head->b.tid; iter->pos = iter->head->beg;
        } else if (iter->pos < iter->head->beg) {
            iter->pos = iter->head->beg;
        } else ++iter->pos;
        if (n_plp) return iter->plp;
        if (iter->is_eof && iter->head == iter->tail) break;
    }
    return NULL;
}

const opileup1_t *oplp_auto(oplp_t *iter, int *_tid, hpos_t *_pos, int *_n_plp)
{
    const opileup1_t *plp;
    if (iter->func == 0 || iter->error) { *_n_plp = -1; return 0; }
    if ((plp = oplp_next(iter, _tid, _pos, _n_plp)) != 0) return plp;
    *_n_plp = 0;
    if (iter->is_eof) return 0;
    int ret;
    while ((ret = iter->func(iter->data, &iter->b)) >= 0) {
        if (oplp_push(iter, &iter->b) < 0) { *_n_plp = -1; return 0; }
        if ((plp = oplp_next(iter, _tid, _pos, _n_plp)) != 0) return plp;
    }
    if (ret < -1) { iter->error = ret; *_n_plp = -1; return 0; }
    if (oplp_push(iter, 0) < 0) { *_n_plp = -1; return 0; }
    if ((plp = oplp_next(iter, _tid, _pos, _n_plp)) != 0) return plp;
    return 0;
}

/* ---- bam_mplp_* ---- */
struct omplp {
    int n;
    int32_t min_tid, *tid;
    hpos_t min_pos, *pos;
    oplp_t **iter;
    int *n_plp;
    const opileup1_t **plp;
};

omplp_t *omplp_init(int n, oplp_auto_f func, void **data)
{
    omplp_t *iter = (omplp_t *)calloc(1, sizeof(*iter));
    iter->pos = (hpos_t *)calloc((size_t)n, sizeof(hpos_t));
    iter->tid = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    iter->n_plp = (int *)calloc((size_t)n, sizeof(int));
    iter->plp = (const opileup1_t **)calloc((size_t)n, sizeof(void *));
    iter->iter = (oplp_t **)calloc((size_t)n, sizeof(void *));
    iter->n = n;
    iter->min_pos = HPOS_MAX;
    iter->min_tid = -1;   /* (uint32_t)-1 */
    for (int i = 0; i < n; ++i) {
        iter->iter[i] = oplp_init(func, data[i]);
        iter->pos[i] = iter->min_pos;
        iter->tid[i] = iter->min_tid;
    }
    return iter;
}
void omplp_destroy(omplp_t *iter)
{
    if (!iter) return;
    for (int i = 0; i < iter->n; ++i) oplp_destroy(iter->iter[i]);
    free(iter->iter); free(iter->pos); free(iter->tid); free(iter->n_plp); free(iter->plp);
    free(iter);
}
void omplp_set_maxcnt(omplp_t *iter, int maxcnt)
{
    for (int i = 0; i < iter->n; ++i) iter->iter[i]->maxcnt = maxcnt;
}
void oplp_constructor(oplp_t *iter, int (*func)(void *data, const orec_t *b, void *cd)) { iter->construct = func; }
void omplp_constructor(omplp_t *iter, int (*func)(void *data, const orec_t *b, void *cd))
{
    for (int i = 0; i < iter->n; ++i) iter->iter[i]->construct = func;
}
int omplp_init_overlaps(omplp_t *iter)
{
    for (int i = 0; i < iter->n; ++i) oplp_init_overlaps(iter->iter[i]);
    return 0;
}
int omplp_auto(omplp_t *iter, int *_tid, hpos_t *_pos, int *n_plp, const opileup1_t **plp)
{
    int i, ret = 0;
    hpos_t new_min_pos = HPOS_MAX;
    uint32_t new_min_tid = (uint32_t)-1;
    for (i = 0; i < iter->n; ++i) {
        if (iter->pos[i] == iter->min_pos && iter->tid[i] == iter->min_tid) {
            int tid; hpos_t pos;
            iter->plp[i] = oplp_auto(iter->iter[i], &tid, &pos, &iter->n_plp[i]);
            if (iter->iter[i]->error) return -1;
            if (iter->plp[i]) { iter->tid[i] = tid; iter->pos[i] = pos; }
            else { iter->tid[i] = 0; iter->pos[i] = 0; }
        }
        if (iter->plp[i]) {
            if ((uint32_t)iter->tid[i] < new_min_tid) {
                new_min_tid = (uint32_t)iter->tid[i];
                new_min_pos = iter->pos[i];
            } else if ((uint32_t)iter->tid[i] == new_min_tid && iter->pos[i] < new_min_pos) {
                new_min_pos = iter->pos[i];
            }
        }
    }
    iter->min_pos = new_min_pos;
    iter->min_tid = (int32_t)new_min_tid;
    if (new_min_pos == HPOS_MAX) return 0;
    *_tid = (int)new_min_tid; *_pos = new_min_pos;
    for (i = 0; i < iter->n; ++i) {
        if (iter->pos[i] == iter->min_pos && iter->tid[i] == iter->min_tid) {
            n_plp[i] = iter->n_plp[i]; plp[i] = iter->plp[i];
            ++ret;
        } else { n_plp[i] = 0; plp[i] = 0; }
    }
    return ret;
}
