/*
 * oracle/o_mpileup.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of samtools bam_plcmd.c (reference file:line cited per
 * function): pileup_seq :54-169, print_empty_pileup :372-398, mplp_func
 * :400-461, mpileup :470-934, bam_mpileup option table :1075-1272, plus
 * sample.c:79-122 (sample counting for the mandatory stderr line).
 * Pinned by test/mpileup/mpileup.reg goldens (see tests/test_oracle_goldens.py).
 * -M/--output-mods: o_mods.c restates HTSlib's MM/ML parser (pinned on mp2.out / mp2-noins.out).  Not supported (documented):
 * CRAM input.
 */
#include "o_plp.h"
#include <getopt.h>
#include <errno.h>
#include <limits.h>
#include <math.h>
#include <ctype.h>

#define MPLP_NO_ORPHAN  (1<<3)
#define MPLP_REALN      (1<<4)
#define MPLP_REDO_BAQ   (1<<6)
#define MPLP_ILLUMINA13 (1<<7)
#define MPLP_IGNORE_RG  (1<<8)
#define MPLP_SMART_OVERLAPS (1<<10)
#define MPLP_PRINT_MAPQ_CHAR (1<<11)
#define MPLP_PRINT_QPOS  (1<<12)
#define MPLP_PRINT_QNAME (1<<13)
#define MPLP_PRINT_FLAG  (1<<14)
#define MPLP_PRINT_RNAME (1<<15)
#define MPLP_PRINT_POS   (1<<16)
#define MPLP_PRINT_MAPQ  (1<<17)
#define MPLP_PRINT_CIGAR (1<<18)
#define MPLP_PRINT_RNEXT (1<<19)
#define MPLP_PRINT_PNEXT (1<<20)
#define MPLP_PRINT_TLEN  (1<<21)
#define MPLP_PRINT_SEQ   (1<<22)
#define MPLP_PRINT_QUAL  (1<<23)
#define MPLP_PRINT_RLEN  (1<<24)
#define MPLP_PRINT_MODS  (1<<25)
#define MPLP_PRINT_QPOS5 (1<<26)
#define MPLP_PRINT_LAST  (1<<27)
#define MPLP_MAX_DEPTH 8000

typedef struct {
    int min_mq, flag, min_baseQ, capQ_thres, max_depth, all, rev_del;
    int rflag_require, rflag_filter;
    char *reg, *fai_fname, *output_fname;
    ofasta_t *fai;
    obed_t *bed;
    char **rg_excl; int n_rg_excl;
    char **auxlist; int n_aux;
    char sep, empty, no_ins, no_ins_mods, no_del, no_ends;
} mplp_conf_t;

typedef struct {
    oreader_t *fp;
    int has_iter;
    ohdr_t *h;
    const mplp_conf_t *conf;
} mplp_aux_t;

/* mplp_get_ref (bam_plcmd.c:289-352): whole-contig fetch; caching is not observable */
static int mplp_get_ref(mplp_aux_t *ma, int tid, const char **ref, hpos_t *ref_len)
{
    if (!ma->conf->fai || tid < 0 || tid >= ma->h->n_ref) { *ref = NULL; return 0; }
    *ref = fa_fetch(ma->conf->fai, ma->h->name[tid], ref_len);
    if (!*ref) { *ref_len = 0; return 0; }
    return 1;
}

/* HTSlib realn.c sam_cap_mapq (mpileup -C, calmd -C).  No reference golden uses -C; tests/test_cap_mapq_vectors.py pins the clip term, the
 * square root, the drop rule and the sign of the M term on vectors derived by hand from doc/samtools-mpileup.1:228-242 */
int o_cap_mapq(orec_t *b, const char *ref, hpos_t ref_len, int thres)
{
    uint8_t *seq = b->seq, *qual = b->qual;
    int i, y, mm, q, len, clip_l, clip_q;
    hpos_t x;
    double t;
    if (thres < 0) thres = 40;
    mm = q = len = clip_l = clip_q = 0;
    for (i = y = 0, x = b->pos; i < (int)b->n_cigar; ++i) {
        int j, l = (int)cig_len(b->cigar[i]), op = cig_op(b->cigar[i]);
        if (op == C_M || op == C_EQ || op == C_X) {
            for (j = 0; j < l; ++j) {
                int c1, c2, z = y + j;
                if (x + j >= ref_len || ref[x + j] == '\0') break;
                if (z >= b->l_qseq) continue;      /* a record without SEQ: HTSlib reads behind the record here (undefined); a base that is not there is no mismatch */
                c1 = rec_seqi(seq, z); c2 = nt16_table[(unsigned char)ref[x + j]];
                if (c2 != 15 && c1 != 15 && qual[z] >= 13) {
                    ++len;
                    if (c1 && c1 != c2 && qual[z] >= 13) { ++mm; q += qual[z] > 33 ? 33 : qual[z]; }
                }
            }
            if (j < l) break;
            x += l; y += l; len += l;
        } else if (op == C_D) {
            for (j = 0; j < l; ++j) if (x + j >= ref_len || ref[x + j] == '\0') break;
            if (j < l) break;
            x += l;
        } else if (op == C_S) {
            for (j = 0; j < l; ++j) if (y + j < b->l_qseq) clip_q += qual[y + j];
            clip_l += l; y += l;
        } else if (op == C_H) { clip_q += 13 * l; clip_l += l; }
        else if (op == C_I) y += l;
        else if (op == C_N) x += l;
    }
    for (i = 0, t = 1; i < mm; ++i) t *= (double)len / (i + 1);
    t = q - 4.343 * log(t) + clip_q / 5.;
    if (t > thres) return -1;
    if (t < 0) t = 0;
    t = sqrt((thres - t) / thres) * thres;
    return (int)(t + .499);
}

/* bam_plcmd.c:400-461 */
static int mplp_func(void *data, orec_t *b)
{
    const char *ref = NULL;
    mplp_aux_t *ma = (mplp_aux_t *)data;
    int ret, skip = 0;
    hpos_t ref_len = 0;
    int has_ref = 0, last_tid = -1;
    do {
        ret = rd_next(ma->fp, b);
        if (ret < 0) break;
        if (b->tid < 0 || (b->flag & F_UNMAP)) { skip = 1; continue; }
        if (ma->conf->rflag_require && !(ma->conf->rflag_require & b->flag)) { skip = 1; continue; }
        if (ma->conf->rflag_filter && (ma->conf->rflag_filter & b->flag)) { skip = 1; continue; }
        if (ma->conf->bed && ma->conf->all == 0) {
            skip = !bed_olap(ma->conf->bed, ma->h->name[b->tid], b->pos, rec_endpos(b));
            if (skip) continue;
        }
        if (ma->conf->rg_excl) {
            const uint8_t *rg = rec_aux_get(b, "RG");
            skip = 0;
            if (rg && (*rg == 'Z'))
                for (int i = 0; i < ma->conf->n_rg_excl; ++i)
                    if (!strcmp(ma->conf->rg_excl[i], (const char *)(rg + 1))) { skip = 1; break; }
            if (skip) continue;
        }
        if (ma->conf->flag & MPLP_ILLUMINA13) {
            for (int i = 0; i < b->l_qseq; ++i) b->qual[i] = b->qual[i] > 31 ? b->qual[i] - 31 : 0;
        }
        if (ma->conf->fai && b->tid >= 0) {
            if (!has_ref || last_tid != b->tid) {
                has_ref = mplp_get_ref(ma, b->tid, &ref, &ref_len);
                last_tid = b->tid;
            }
            if (has_ref && ref_len <= b->pos) {
                fprintf(stderr, "[%s] Skipping because %lld is outside of %lld [ref:%d]\n",
                        __func__, (long long)b->pos, (long long)ref_len, b->tid);
                skip = 1;
                continue;
            }
        } else has_ref = 0;
        skip = 0;
        if (has_ref && (ma->conf->flag & MPLP_REALN))
            o_prob_realn(b, ref, ref_len, (ma->conf->flag & MPLP_REDO_BAQ) ? 7 : 3);
        if (has_ref && ma->conf->capQ_thres > 10) {
            int q = o_cap_mapq(b, ref, ref_len, ma->conf->capQ_thres);
            if (q < 0) skip = 1;
            else if (b->mapq > q) b->mapq = (uint8_t)q;
        }
        if (b->mapq < ma->conf->min_mq) skip = 1;
        else if ((ma->conf->flag & MPLP_NO_ORPHAN) && (b->flag & F_PAIRED) && !(b->flag & F_PROPER_PAIR)) skip = 1;
    } while (skip);
    return ret;
}

static inline int tolower_c(int c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }
static inline int toupper_c(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

/* bam_plcmd.c:54-169; m = the read's base modifications with --output-mods (o_mods.c), else NULL */
static int pileup_seq(ostr_t *ks_seq, const opileup1_t *p, hpos_t pos, hpos_t ref_len, const char *ref,
                      ostr_t *ks_mod, int rev_del, int no_ins, int no_ins_mods, int no_del, int no_ends, const omods_t *m)
{
    int j;
    no_ins_mods |= no_ins;
    if (!no_ends && p->is_head) {
        os_putc(ks_seq, '^');
        os_putc(ks_seq, p->b->mapq > 93 ? 126 : p->b->mapq + 33);
    }
    if (!p->is_del) {
        const char seq_nt_str_lc[] = ",acmgrsvtwyhkdbn";
        const char seq_nt_str_uc[] = ".ACMGRSVTWYHKDBN";
        int c = p->qpos < p->b->l_qseq ? rec_seqi(p->b->seq, p->qpos) : 15;
        if (ref) {
            int rb = pos < ref_len ? nt16_table[(uint8_t)ref[pos]] : 15;
            if (c == rb) c = 0;
        }
        c = rec_is_rev(p->b) ? seq_nt_str_lc[c] : seq_nt_str_uc[c];
        os_putc(ks_seq, c);
        if (m) omods_put(m, p->qpos, ks_seq);
    } else {
        os_putc(ks_seq, p->is_refskip ? (rec_is_rev(p->b) ? '<' : '>')
                                      : ((rec_is_rev(p->b) && rev_del) ? '#' : '*'));
    }
    int del_len = -p->indel;
    if (p->indel > 0) {
        int len = oplp_insertion_mod(p, m && !no_ins_mods ? m : NULL, ks_mod, &del_len);
        if (len < 0) return -1;
        if (no_ins < 2) { os_putc(ks_seq, '+'); os_putll(ks_seq, len); }
        if (!no_ins) {
            int in_mod = 0;
            if (rec_is_rev(p->b)) {
                char pad = rev_del ? '#' : '*';
                for (j = 0; j < (int)ks_mod->l; j++) {
                    if (ks_mod->s[j] == '[') in_mod = 1;
                    else if (ks_mod->s[j] == ']') in_mod = 0;
                    os_putc(ks_seq, ks_mod->s[j] != '*' ? (in_mod ? ks_mod->s[j] : tolower_c(ks_mod->s[j])) : pad);
                }
            } else {
                for (j = 0; j < (int)ks_mod->l; j++) {
                    if (ks_mod->s[j] == '[') in_mod = 1;
                    if (ks_mod->s[j] == ']') in_mod = 0;
                    os_putc(ks_seq, in_mod ? ks_mod->s[j] : toupper_c(ks_mod->s[j]));
                }
            }
        }
    }
    if (del_len > 0) {
        if (no_del < 2) os_putll(ks_seq, -del_len);
        if (!no_del) {
            for (j = 1; j <= del_len; ++j) {
                int c = (ref && (int)pos + j < ref_len) ? ref[pos + j] : 'N';
                os_putc(ks_seq, rec_is_rev(p->b) ? tolower_c(c) : toupper_c(c));
            }
        }
    }
    if (!no_ends && p->is_tail) os_putc(ks_seq, '$');
    return 0;
}

/* bam_plcmd.c:372-398 */
static void print_empty_pileup(ostr_t *out, const mplp_conf_t *conf, const char *tname,
                               hpos_t pos, int n, const char *ref, hpos_t ref_len)
{
    os_puts(out, tname); os_putc(out, '\t');
    os_putll(out, pos + 1); os_putc(out, '\t');
    os_putc(out, (ref && pos < ref_len) ? ref[pos] : 'N');
    for (int i = 0; i < n; ++i) {
        os_putsn(out, "\t0\t*\t*", 6);
        int flag_value = MPLP_PRINT_MAPQ_CHAR;
        while (flag_value < MPLP_PRINT_LAST) {
            if (flag_value != MPLP_PRINT_MODS && (conf->flag & flag_value)) os_putsn(out, "\t*", 2);
            flag_value <<= 1;
        }
        for (int t = 0; t < conf->n_aux; ++t) os_putsn(out, "\t*", 2);
    }
    os_putc(out, '\n');
}

/* sample.c:79-122 bam_smpl_add: counts distinct SM values (or file names) */
typedef struct { char **rg; int n_rg; char **sm; int n_sm; } smpl_t;
static int strlist_find(char **l, int n, const char *s) { for (int i = 0; i < n; ++i) if (!strcmp(l[i], s)) return i; return -1; }
static void smpl_add_pair(smpl_t *s, const char *key, const char *val)
{
    if (strlist_find(s->rg, s->n_rg, key) >= 0) return;
    s->rg = (char **)realloc(s->rg, sizeof(char *) * (size_t)(s->n_rg + 1)); s->rg[s->n_rg++] = strdup(key);
    if (strlist_find(s->sm, s->n_sm, val) < 0) {
        s->sm = (char **)realloc(s->sm, sizeof(char *) * (size_t)(s->n_sm + 1)); s->sm[s->n_sm++] = strdup(val);
    }
}
static void smpl_add(smpl_t *sm, const char *fn, const char *txt)
{
    if (!txt) { smpl_add_pair(sm, fn, fn); return; }
    char *copy = strdup(txt);
    char *p = copy, *q, *r;
    int n = 0; char *first_sm = NULL;
    ostr_t buf = { 0, 0, NULL };
    while ((q = strstr(p, "@RG")) != 0) {
        p = q + 3;
        r = q = 0;
        if ((q = strstr(p, "\tID:")) != 0) q += 4;
        if ((r = strstr(p, "\tSM:")) != 0) r += 4;
        if (r && q) {
            char *u, *v; int oq, orr;
            for (u = q; *u && *u != '\t' && *u != '\n'; ++u);
            for (v = r; *v && *v != '\t' && *v != '\n'; ++v);
            oq = *u; orr = *v; *u = *v = '\0';
            os_clear(&buf); os_puts(&buf, fn); os_putc(&buf, '/'); os_puts(&buf, q);
            smpl_add_pair(sm, buf.s, r);
            if (!first_sm) first_sm = strdup(r);
            *u = (char)oq; *v = (char)orr;
        } else break;
        p = q > r ? q : r;
        ++n;
    }
    if (n == 0) smpl_add_pair(sm, fn, fn);
    else if (n == 1 && first_sm) smpl_add_pair(sm, fn, first_sm);
    free(first_sm); free(buf.s); free(copy);
}

/* HTSlib kputd (kstring.c), which bam_plcmd.c:840 calls for 'f' / 'd' aux values.  Restated from the published source (HTSlib is
 * not in the reference tree): zero -> "0" / "-0"; outside [0.0001, 999999] -> "%g"; inside, the value times 10^10 is truncated
 * to an integer, half a unit of the sixth significant digit added, the decimal digits written right to left into a small
 * buffer, cut to six significant digits, the point inserted and trailing zeros culled.  Unlike "%g" this rounds half UP on the
 * truncated decimal expansion: 123456.5 -> "123457", 12345.25 -> "12345.3". */
static void put_double(ostr_t *s, double d);
void o_put_double(ostr_t *s, double d) { put_double(s, d); }
static void put_double(ostr_t *s, double d)
{
    char buf[21], *cp = buf + 20, *ep;
    if (d == 0) { os_puts(s, signbit(d) ? "-0" : "0"); return; }
    if (d < 0) { os_putc(s, '-'); d = -d; }
    if (!(d >= 0.0001 && d <= 999999)) { char b[64]; snprintf(b, sizeof b, "%g", d); os_puts(s, b); return; }
    uint64_t i = (uint64_t)(d * 10000000000LL);
    if (d < .0001) i += 0;
    else if (d < 0.001) i += 5;
    else if (d < 0.01) i += 50;
    else if (d < 0.1) i += 500;
    else if (d < 1) i += 5000;
    else if (d < 10) i += 50000;
    else if (d < 100) i += 500000;
    else if (d < 1000) i += 5000000;
    else if (d < 10000) i += 50000000;
    else if (d < 100000) i += 500000000;
    else i += 5000000000LL;
    do { *--cp = (char)('0' + i % 10); i /= 10; } while (i >= 1);
    buf[20] = 0;
    int p = (int)(buf + 20 - cp);
    if (p <= 10) {                      /* d < 1 */
        cp[6] = 0; ep = cp + 5;         /* six digits */
        while (p < 10) { *--cp = '0'; p++; }
        *--cp = '.';
        *--cp = '0';
    } else {
        char *xp = --cp;
        while (p > 10) { xp[0] = xp[1]; p--; xp++; }
        xp[0] = '.';
        cp[7] = 0; ep = cp + 6;
        if (cp[6] == '.') cp[6] = 0;
    }
    while (*ep == '0' && ep > cp) ep--;            /* cull trailing zeros */
    {
        char *z = ep + 1;
        while (ep > cp) {
            if (*ep == '.') { if (z[-1] == '.') z[-1] = 0; else z[0] = 0; break; }
            ep--;
        }
    }
    os_puts(s, cp);
}

static long long aux2i(const uint8_t *t)
{
    switch (*t) {
    case 'c': return (int8_t)t[1];
    case 'C': return t[1];
    case 's': { int16_t v; memcpy(&v, t + 1, 2); return v; }
    case 'S': { uint16_t v; memcpy(&v, t + 1, 2); return v; }
    case 'i': { int32_t v; memcpy(&v, t + 1, 4); return v; }
    case 'I': { uint32_t v; memcpy(&v, t + 1, 4); return v; }
    }
    return 0;
}

/* bam_plcmd.c:470-934 */
static int mpileup(mplp_conf_t *conf, int nfn, char **fn)
{
    mplp_aux_t **data;
    int i, tid = 0, *n_plp, tid0 = 0, max_depth;
    hpos_t pos = 0, beg0 = 0, end0 = HPOS_MAX, ref_len = 0;
    const opileup1_t **plp;
    omplp_t *iter;
    ohdr_t *h = NULL;
    const char *ref = NULL;
    FILE *pileup_fp = NULL;
    smpl_t sm; memset(&sm, 0, sizeof sm);
    ostr_t buf = { 0, 0, NULL };

    data = (mplp_aux_t **)calloc((size_t)nfn, sizeof(mplp_aux_t *));
    plp = (const opileup1_t **)calloc((size_t)nfn, sizeof(void *));
    n_plp = (int *)calloc((size_t)nfn, sizeof(int));
    if (nfn == 0) { fprintf(stderr, "[%s] no input file/data given\n", __func__); exit(EXIT_FAILURE); }

    for (i = 0; i < nfn; ++i) {
        data[i] = (mplp_aux_t *)calloc(1, sizeof(mplp_aux_t));
        data[i]->fp = rd_open(fn[i]);
        if (!data[i]->fp) {
            fprintf(stderr, "[%s] failed to open %s: %s\n", __func__, fn[i], strerror(errno));
            exit(EXIT_FAILURE);
        }
        data[i]->conf = conf;
        ohdr_t *h_tmp = rd_header(data[i]->fp);
        smpl_add(&sm, fn[i], (conf->flag & MPLP_IGNORE_RG) ? 0 : h_tmp->text);
        if (conf->reg) {
            int rtid; hpos_t rbeg, rend;
            if (parse_region(h_tmp, conf->reg, &rtid, &rbeg, &rend) < 0) {
                fprintf(stderr, "[E::%s] fail to parse region '%s' with %s\n", __func__, conf->reg, fn[i]);
                exit(EXIT_FAILURE);
            }
            rd_set_region(data[i]->fp, rtid, rbeg, rend);
            data[i]->has_iter = 1;
            if (i == 0) beg0 = rbeg, end0 = rend, tid0 = rtid;
        }
        if (i == 0) h = data[i]->h = h_tmp;
        else data[i]->h = h;
    }
    fprintf(stderr, "[%s] %d samples in %d input files\n", __func__, sm.n_sm, nfn);

    pileup_fp = conf->output_fname ? fopen(conf->output_fname, "w") : stdout;
    if (pileup_fp == NULL) {
        fprintf(stderr, "[%s] failed to write to %s: %s\n", __func__, conf->output_fname, strerror(errno));
        exit(EXIT_FAILURE);
    }

    iter = omplp_init(nfn, mplp_func, (void **)data);
    if (conf->flag & MPLP_SMART_OVERLAPS) omplp_init_overlaps(iter);
    if (!conf->max_depth) {
        max_depth = INT_MAX;
        fprintf(stderr, "[%s] Max depth set to maximum value (%d)\n", __func__, INT_MAX);
    } else {
        max_depth = conf->max_depth;
        if (max_depth * nfn > 1 << 20)
            fprintf(stderr, "[%s] Combined max depth is above 1M. Potential memory hog!\n", __func__);
    }
    omplp_set_maxcnt(iter, max_depth);

    int ret, last_tid = -1, got_ref = 0;
    hpos_t last_pos = -1;
    int one_seq = 0;
    ostr_t ks_seq = { 0, 0, NULL }, ks_mod = { 0, 0, NULL }, ks_qual = { 0, 0, NULL };

    while ((ret = omplp_auto(iter, &tid, &pos, n_plp, plp)) > 0) {
        one_seq = 1;
        if (conf->reg && (pos < beg0 || pos >= end0)) continue;
        if (conf->all) {
            while (tid > last_tid) {
                if (last_tid >= 0 && !conf->reg) {
                    while (++last_pos < h->len[last_tid]) {
                        if (conf->bed && bed_olap(conf->bed, h->name[last_tid], last_pos, last_pos + 1) == 0) continue;
                        print_empty_pileup(&buf, conf, h->name[last_tid], last_pos, nfn, ref, ref_len);
                        fwrite(buf.s, 1, buf.l, pileup_fp);
                        os_clear(&buf);
                    }
                }
                last_tid++;
                got_ref = 0;
                last_pos = -1;
                if (conf->all < 2) break;
                if (tid > last_tid) got_ref = mplp_get_ref(data[0], last_tid, &ref, &ref_len);
            }
        }
        if (!got_ref || last_tid != tid) {
            got_ref = mplp_get_ref(data[0], tid, &ref, &ref_len);
            last_tid = tid;
        }
        if (conf->all) {
            while (++last_pos < pos) {
                if (conf->reg && last_pos < beg0) continue;
                if (conf->bed && bed_olap(conf->bed, h->name[tid], last_pos, last_pos + 1) == 0) continue;
                print_empty_pileup(&buf, conf, h->name[tid], last_pos, nfn, ref, ref_len);
                fwrite(buf.s, 1, buf.l, pileup_fp);
                os_clear(&buf);
            }
            last_pos = pos;
        }
        if (conf->bed && tid >= 0 && !bed_olap(conf->bed, h->name[tid], pos, pos + 1)) continue;

        os_puts(&buf, h->name[tid]); os_putc(&buf, '\t');
        os_putll(&buf, pos + 1); os_putc(&buf, '\t');
        os_putc(&buf, (ref && pos < ref_len) ? ref[pos] : 'N');

        for (i = 0; i < nfn; ++i) {
            int j, cnt;
            os_clear(&ks_seq); os_clear(&ks_qual); os_clear(&ks_mod);
            for (j = cnt = 0; j < n_plp[i]; ++j) {
                const opileup1_t *p = plp[i] + j;
                int c = p->qpos < p->b->l_qseq ? p->b->qual[p->qpos] : 0;
                if (c >= conf->min_baseQ) {
                    omods_t mods; const omods_t *mp = NULL;
                    if (conf->flag & MPLP_PRINT_MODS) { omods_parse(p->b, &mods); mp = &mods; }      /* (bam_plcmd.c:356-362 parses once per read; the result is the same) */
                    pileup_seq(&ks_seq, p, pos, ref_len, ref, &ks_mod, conf->rev_del, conf->no_ins, conf->no_ins_mods, conf->no_del, conf->no_ends, mp);
                    if (mp) omods_free(&mods);
                    os_putc(&ks_qual, c + 33 < 126 ? c + 33 : 126);
                    cnt++;
                }
            }
            os_putc(&buf, '\t'); os_putll(&buf, cnt); os_putc(&buf, '\t');
            if (n_plp[i] == 0) {
                os_putsn(&buf, "*\t*", 3);
                int flag_value = MPLP_PRINT_MAPQ_CHAR;
                while (flag_value < MPLP_PRINT_LAST) {
                    if (flag_value != MPLP_PRINT_MODS && (conf->flag & flag_value)) os_putsn(&buf, "\t*", 2);
                    flag_value <<= 1;
                }
                for (int t = 0; t < conf->n_aux; ++t) os_putsn(&buf, "\t*", 2);
            } else {
                if (ks_seq.l) os_putsn(&buf, ks_seq.s, ks_seq.l); else os_putc(&buf, '*');
                os_putc(&buf, '\t');
                if (ks_qual.l) os_putsn(&buf, ks_qual.s, ks_qual.l); else os_putc(&buf, '*');

                int flag_value = MPLP_PRINT_MAPQ_CHAR;
                while (flag_value < MPLP_PRINT_LAST) {
                    if (flag_value != MPLP_PRINT_MODS && (conf->flag & flag_value)) {
                        int n = 0;
                        os_putc(&buf, '\t');
                        for (j = 0; j < n_plp[i]; ++j) {
                            const opileup1_t *p = &plp[i][j];
                            int c = p->qpos < p->b->l_qseq ? p->b->qual[p->qpos] : 0;
                            if (c < conf->min_baseQ) continue;
                            if (n > 0 && flag_value != MPLP_PRINT_MAPQ_CHAR) os_putc(&buf, ',');
                            n++;
                            switch (flag_value) {
                            case MPLP_PRINT_MAPQ_CHAR:
                                c = p->b->mapq + 33; if (c > 126) c = 126;
                                os_putc(&buf, c);
                                break;
                            case MPLP_PRINT_QPOS: os_putll(&buf, p->qpos + 1); break;
                            case MPLP_PRINT_QPOS5: {
                                int pos5 = rec_is_rev(p->b) ? p->b->l_qseq - p->qpos + p->is_del : p->qpos + 1;
                                os_putll(&buf, pos5);
                                break;
                            }
                            case MPLP_PRINT_QNAME: os_puts(&buf, p->b->qname); break;
                            case MPLP_PRINT_FLAG: os_putll(&buf, p->b->flag); break;
                            case MPLP_PRINT_RNAME:
                                if (p->b->tid >= 0) os_puts(&buf, h->name[p->b->tid]); else os_putc(&buf, '*');
                                break;
                            case MPLP_PRINT_POS: os_putll(&buf, (long long)p->b->pos + 1); break;
                            case MPLP_PRINT_MAPQ: os_putll(&buf, p->b->mapq); break;
                            case MPLP_PRINT_RNEXT:
                                if (p->b->mtid >= 0) os_puts(&buf, h->name[p->b->mtid]); else os_putc(&buf, '*');
                                break;
                            case MPLP_PRINT_PNEXT: os_putll(&buf, (long long)p->b->mpos + 1); break;
                            case MPLP_PRINT_RLEN: os_putll(&buf, p->b->l_qseq); break;
                            }
                        }
                        if (!n) os_putc(&buf, '*');
                    }
                    flag_value <<= 1;
                }
                for (int t = 0; t < conf->n_aux; ++t) {
                    int n = 0;
                    os_putc(&buf, '\t');
                    for (j = 0; j < n_plp[i]; ++j) {
                        const opileup1_t *p = &plp[i][j];
                        int c = p->qpos < p->b->l_qseq ? p->b->qual[p->qpos] : 0;
                        if (c < conf->min_baseQ) continue;
                        if (n > 0) os_putc(&buf, conf->sep);
                        n++;
                        const uint8_t *tag_u = rec_aux_get(p->b, conf->auxlist[t]);
                        if (!tag_u) { os_putc(&buf, conf->empty); continue; }
                        int tag_supported = 0;
                        if (*tag_u == 'Z' || *tag_u == 'H') { os_puts(&buf, (const char *)tag_u + 1); tag_supported = 1; }
                        if (*tag_u == 'I' || *tag_u == 'i' || *tag_u == 'C' || *tag_u == 'c' || *tag_u == 'S' || *tag_u == 's') {
                            os_putll(&buf, aux2i(tag_u)); tag_supported = 1;
                        }
                        if (*tag_u == 'd' || *tag_u == 'f') {
                            double v; if (*tag_u == 'f') { float f; memcpy(&f, tag_u + 1, 4); v = f; } else memcpy(&v, tag_u + 1, 8);
                            put_double(&buf, v); tag_supported = 1;
                        }
                        if (*tag_u == 'A') { os_putc(&buf, tag_u[1]); tag_supported = 1; }
                        if (!tag_supported) os_putc(&buf, '*');
                    }
                    if (!n) os_putc(&buf, '*');
                }
            }
        }
        os_putc(&buf, '\n');
        if (buf.l != fwrite(buf.s, 1, buf.l, pileup_fp)) { fprintf(stderr, "Failed to write pileup data.\n"); goto fail; }
        os_clear(&buf);
    }

    if (ret < 0) {
        fflush(stdout);
        fprintf(stderr, "samtools mpileup: error reading from input file\n");
        ret = EXIT_FAILURE;
        goto fail;
    }

    if (conf->all) {
        if (last_tid < 0 && conf->reg && conf->all > 1) {
            last_tid = tid0;
            last_pos = beg0 - 1;
            mplp_get_ref(data[0], tid0, &ref, &ref_len);
        } else if (last_tid < 0 && !one_seq && conf->all > 1) {
            last_tid = 0;
        }
        while (last_tid >= 0 && last_tid < h->n_ref) {
            mplp_get_ref(data[0], last_tid, &ref, &ref_len);
            while (++last_pos < h->len[last_tid]) {
                if (last_pos >= end0) break;
                if (conf->bed && bed_olap(conf->bed, h->name[last_tid], last_pos, last_pos + 1) == 0) continue;
                print_empty_pileup(&buf, conf, h->name[last_tid], last_pos, nfn, ref, ref_len);
                fwrite(buf.s, 1, buf.l, pileup_fp);
                os_clear(&buf);
            }
            last_tid++;
            last_pos = -1;
            if (conf->all < 2 || conf->reg) break;
        }
    }

fail:
    free(ks_seq.s); free(ks_mod.s); free(ks_qual.s);
    if (pileup_fp && conf->output_fname) fclose(pileup_fp);
    free(buf.s);
    omplp_destroy(iter);
    for (i = 0; i < nfn; ++i) { rd_close(data[i]->fp); free(data[i]); }
    free(data); free(plp); free(n_plp);
    for (i = 0; i < sm.n_rg; ++i) free(sm.rg[i]);
    for (i = 0; i < sm.n_sm; ++i) free(sm.sm[i]);
    free(sm.rg); free(sm.sm);
    return ret;
}

/* bam_plcmd.c:944-999 read_file_list (plain list, one path per line) */
int o_read_file_list(const char *file_list, int *n, char ***argv)
{
    FILE *fh = fopen(file_list, "r");
    if (!fh) { fprintf(stderr, "%s: %s\n", file_list, strerror(errno)); return 1; }
    char buf[1024]; char **files = NULL; int nfiles = 0;
    while (fgets(buf, sizeof buf, fh)) {
        int len = (int)strlen(buf);
        while (len > 0 && isspace((unsigned char)buf[len - 1])) len--;
        if (!len) continue;
        buf[len] = 0;
        files = (char **)realloc(files, sizeof(char *) * (size_t)(nfiles + 1));
        files[nfiles++] = strdup(buf);
    }
    fclose(fh);
    if (!nfiles) { fprintf(stderr, "No files read from %s\n", file_list); return 1; }
    *argv = files; *n = nfiles;
    return 0;
}

/* bam_plcmd.c:240-287 build_auxlist */
static int build_auxlist(mplp_conf_t *conf, char *optstring)
{
    static const struct { const char *name; int supported; } colnames[12] = {
        { "QNAME", 1 }, { "FLAG", 1 }, { "RNAME", 1 }, { "POS", 1 }, { "MAPQ", 1 }, { "CIGAR", 0 },
        { "RNEXT", 1 }, { "PNEXT", 1 }, { "TLEN", 0 }, { "SEQ", 0 }, { "QUAL", 0 }, { "RLEN", 1 } };
    if (!optstring) return 0;
    char *save_p;
    for (char *tag = strtok_r(optstring, ",", &save_p); tag; tag = strtok_r(NULL, ",", &save_p)) {
        int f = MPLP_PRINT_QNAME, hit = 0;
        for (int i = 0; i < 12; i++, f <<= 1)
            if (colnames[i].supported && !strcmp(colnames[i].name, tag)) { conf->flag |= f; hit = 1; break; }
        if (hit) continue;
        if (strlen(tag) != 2) fprintf(stderr, "[%s] tag '%s' has more than two characters or not supported\n", __func__, tag);
        else {
            conf->auxlist = (char **)realloc(conf->auxlist, sizeof(char *) * (size_t)(conf->n_aux + 1));
            conf->auxlist[conf->n_aux++] = tag;
        }
    }
    return 0;
}

/* bam_plcmd.c:1075-1272 */
int o_main_mpileup(int argc, char *argv[])
{
    int c;
    const char *file_list = NULL;
    char **fn = NULL;
    int nfiles = 0, use_orphan = 0;
    mplp_conf_t mplp;
    memset(&mplp, 0, sizeof(mplp_conf_t));
    mplp.min_baseQ = 13;
    mplp.capQ_thres = 0;
    mplp.max_depth = MPLP_MAX_DEPTH;
    mplp.flag = MPLP_NO_ORPHAN | MPLP_REALN | MPLP_SMART_OVERLAPS;
    mplp.rflag_filter = F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP;
    mplp.sep = ','; mplp.empty = '*';

    static const struct option lopts[] = {
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 },
        { "incl-flags", required_argument, NULL, 1 }, { "excl-flags", required_argument, NULL, 2 },
        { "output", required_argument, NULL, 3 },
        { "output-QNAME", no_argument, NULL, 5 }, { "output-qname", no_argument, NULL, 5 },
        { "illumina1.3+", no_argument, NULL, '6' }, { "count-orphans", no_argument, NULL, 'A' },
        { "bam-list", required_argument, NULL, 'b' },
        { "no-BAQ", no_argument, NULL, 'B' }, { "no-baq", no_argument, NULL, 'B' },
        { "adjust-MQ", required_argument, NULL, 'C' }, { "adjust-mq", required_argument, NULL, 'C' },
        { "max-depth", required_argument, NULL, 'd' },
        { "redo-BAQ", no_argument, NULL, 'E' }, { "redo-baq", no_argument, NULL, 'E' },
        { "fasta-ref", required_argument, NULL, 'f' }, { "reference", required_argument, NULL, 'f' },
        { "exclude-RG", required_argument, NULL, 'G' }, { "exclude-rg", required_argument, NULL, 'G' },
        { "positions", required_argument, NULL, 'l' }, { "region", required_argument, NULL, 'r' },
        { "ignore-RG", no_argument, NULL, 'R' }, { "ignore-rg", no_argument, NULL, 'R' },
        { "min-MQ", required_argument, NULL, 'q' }, { "min-mq", required_argument, NULL, 'q' },
        { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "ignore-overlaps-removal", no_argument, NULL, 'x' }, { "disable-overlap-removal", no_argument, NULL, 'x' },
        { "output-mods", no_argument, NULL, 'M' },
        { "output-BP", no_argument, NULL, 'O' }, { "output-bp", no_argument, NULL, 'O' },
        { "output-BP-5", no_argument, NULL, 14 }, { "output-bp-5", no_argument, NULL, 14 },
        { "output-MQ", no_argument, NULL, 's' }, { "output-mq", no_argument, NULL, 's' },
        { "customized-index", no_argument, NULL, 'X' },
        { "reverse-del", no_argument, NULL, 6 }, { "output-extra", required_argument, NULL, 7 },
        { "output-sep", required_argument, NULL, 8 }, { "output-empty", required_argument, NULL, 9 },
        { "no-output-ins", no_argument, NULL, 10 }, { "no-output-ins-mods", no_argument, NULL, 11 },
        { "no-output-del", no_argument, NULL, 12 }, { "no-output-ends", no_argument, NULL, 13 },
        { NULL, 0, NULL, 0 } };

    optind = 1;
    while ((c = getopt_long(argc, argv, "Af:r:l:q:Q:RC:Bd:b:o:EG:6OsxXaM", lopts, NULL)) >= 0) {
        switch (c) {
        case 'x': mplp.flag &= ~MPLP_SMART_OVERLAPS; break;
        case 1:
            mplp.rflag_require = str2flag(optarg);
            if (mplp.rflag_require < 0) { fprintf(stderr, "Could not parse --rf %s\n", optarg); return 1; }
            break;
        case 2:
            mplp.rflag_filter = str2flag(optarg);
            if (mplp.rflag_filter < 0) { fprintf(stderr, "Could not parse --ff %s\n", optarg); return 1; }
            break;
        case 3: mplp.output_fname = optarg; break;
        case 5: mplp.flag |= MPLP_PRINT_QNAME; break;
        case 6: mplp.rev_del = 1; break;
        case 7: build_auxlist(&mplp, optarg); break;
        case 8: mplp.sep = optarg[0]; break;
        case 9: mplp.empty = optarg[0]; break;
        case 10: mplp.no_ins++; break;
        case 11: mplp.no_ins_mods = 1; break;
        case 12: mplp.no_del++; break;
        case 13: mplp.no_ends = 1; break;
        case 'f':
            mplp.fai = fa_load(optarg);
            if (mplp.fai == NULL) { fprintf(stderr, "[E::fai_load] failed to load %s\n", optarg); return 1; }
            mplp.fai_fname = optarg;
            break;
        case 'd': mplp.max_depth = atoi(optarg); break;
        case 'r': mplp.reg = strdup(optarg); break;
        case 'l':
            mplp.bed = bed_load(optarg);
            if (!mplp.bed) { fprintf(stderr, "samtools mpileup: Could not read file \"%s\"\n", optarg); return 1; }
            break;
        case 'B': mplp.flag &= ~MPLP_REALN; break;
        case 'X': fprintf(stderr, "oracle: -X not supported\n"); return 1;
        case 'E': mplp.flag |= MPLP_REDO_BAQ; break;
        case '6': mplp.flag |= MPLP_ILLUMINA13; break;
        case 'R': mplp.flag |= MPLP_IGNORE_RG; break;
        case 's': mplp.flag |= MPLP_PRINT_MAPQ_CHAR; break;
        case 'O': mplp.flag |= MPLP_PRINT_QPOS; break;
        case 14: mplp.flag |= MPLP_PRINT_QPOS5; break;
        case 'M': mplp.flag |= MPLP_PRINT_MODS; break;
        case 'C': mplp.capQ_thres = atoi(optarg); break;
        case 'q': mplp.min_mq = atoi(optarg); break;
        case 'Q': mplp.min_baseQ = atoi(optarg); break;
        case 'b': file_list = optarg; break;
        case 'o': mplp.output_fname = optarg; break;
        case 'A': use_orphan = 1; break;
        case 'G': {
            FILE *fp_rg; char buf[1024];
            mplp.rg_excl = (char **)calloc(1, sizeof(char *));
            if ((fp_rg = fopen(optarg, "r")) == NULL)
                fprintf(stderr, "[%s] Fail to open file %s. Continue anyway.\n", __func__, optarg);
            while (fp_rg && !feof(fp_rg) && fscanf(fp_rg, "%1023s", buf) > 0) {
                mplp.rg_excl = (char **)realloc(mplp.rg_excl, sizeof(char *) * (size_t)(mplp.n_rg_excl + 1));
                mplp.rg_excl[mplp.n_rg_excl++] = strdup(buf);
            }
            if (fp_rg) fclose(fp_rg);
            break;
        }
        case 'a': mplp.all++; break;
        default:
            fprintf(stderr, "Usage: samtools mpileup [options] in1.bam [in2.bam [...]]\n");
            return 1;
        }
    }
    if (!(mplp.flag & MPLP_REALN) && (mplp.flag & MPLP_REDO_BAQ)) {
        fprintf(stderr, "Error: The -B option cannot be combined with -E\n");
        return 1;
    }
    if (use_orphan) mplp.flag &= ~MPLP_NO_ORPHAN;
    if (argc == 1) { fprintf(stderr, "Usage: samtools mpileup [options] in1.bam [in2.bam [...]]\n"); return 1; }
    int ret;
    if (file_list) {
        if (o_read_file_list(file_list, &nfiles, &fn)) return 1;
        ret = mpileup(&mplp, nfiles, fn);
        for (c = 0; c < nfiles; c++) free(fn[c]);
        free(fn);
    } else {
        nfiles = argc - optind;
        ret = mpileup(&mplp, nfiles, argv + optind);
    }
    for (c = 0; c < mplp.n_rg_excl; ++c) free(mplp.rg_excl[c]);
    free(mplp.rg_excl);
    free(mplp.reg);
    if (mplp.fai) fa_free(mplp.fai);
    if (mplp.bed) bed_free(mplp.bed);
    free(mplp.auxlist);
    return ret;
}
