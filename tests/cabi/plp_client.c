/*
 * plp_client.c -- an EXTERNAL client of the drop-in pileup surface: plain C99, built by the test with
 *     gcc -std=c99 -DSTA_PLP_DROPIN -Iinclude tests/cabi/plp_client.c -Lsamtools_amd/lib -lsamtools_amd
 * It includes nothing but the public header include/samtools_amd_plp.h and uses only the unprefixed HTSlib names
 * (bam_mplp_init, bam_mplp64_auto, bam_plp_insertion_mod, bam_plbuf_push, ...), i.e. it is written the way the
 * reference's own small pileup clients are:
 *   - the pull loop is bedcov.c:303-335 (bam_mplp_init + read callback, bam_mplp_set_maxcnt, bam_mplp_auto loop),
 *   - the push loop is bam_plbuf.c:40-69 (bam_plbuf_init / bam_plbuf_push(b) ... bam_plbuf_push(NULL)),
 *   - inserted sequences come from bam_plp_insertion_mod with a NULL modification state (bam_plcmd.c:119).
 * The only thing it has to bring along is a record source, because HTSlib's sam_read1 is not in this tree: a minimal SAM
 * text parser that fills caller-owned bam1_t records (the 11 mandatory fields; aux fields are not needed by the iterator).
 *
 *   plp_client [-x] [-d maxcnt] [-p] in1.sam [in2.sam ...]
 * prints one line per column with every bam_pileup1_t field, in the format of the oracle's `plpdump`.
 *   plp_client -M | -N [-x] in.sam
 * prints `mpileup -Q0 --output-mods` lines instead (-N: with --no-output-ins-mods), the way bam_plcmd.c does it: a modification state per
 * read through the iterator's constructor hook (bam_plcmd.c:356-369: hts_base_mod_state_alloc + bam_parse_basemod), bam_mods_at_qpos behind
 * every base (:86-109), bam_plp_insertion_mod with the live state (:119).  For that the record source also keeps the MM:Z / ML:B:C fields.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "samtools_amd_plp.h"

typedef struct {
    FILE *fp;
    char **names; int n_names;          /* @SQ SN: values, in header order = tid */
    char *line; size_t cap;
    int have_line;
} src_t;

static int tid_of(const src_t *s, const char *name)
{
    int i;
    for (i = 0; i < s->n_names; ++i) if (strcmp(s->names[i], name) == 0) return i;
    return -1;
}

static int open_src(src_t *s, const char *path)
{
    memset(s, 0, sizeof *s);
    s->fp = fopen(path, "r");
    if (!s->fp) return -1;
    while (getline(&s->line, &s->cap, s->fp) > 0) {
        if (s->line[0] != '@') { s->have_line = 1; break; }
        if (strncmp(s->line, "@SQ", 3) == 0) {
            char *p = strstr(s->line, "\tSN:");
            if (p) {
                char *e;
                p += 4; e = p + strcspn(p, "\t\n");
                s->names = (char **)realloc(s->names, sizeof(char *) * (size_t)(s->n_names + 1));
                s->names[s->n_names] = (char *)malloc((size_t)(e - p) + 1);
                memcpy(s->names[s->n_names], p, (size_t)(e - p)); s->names[s->n_names][e - p] = 0;
                s->n_names++;
            }
        }
    }
    return 0;
}

static const char NT16[] = "=ACMGRSVTWYHKDBN";

/* bam_plp_auto_f: >= 0 ok, -1 EOF, < -1 error; fills the caller-owned record */
static int read_cb(void *data, bam1_t *b)
{
    src_t *s = (src_t *)data;
    char *f[11], *p, *mm = NULL, *ml = NULL;
    int i, n_cig = 0;
    size_t l_qn, pad, l_seq, need, l_aux = 0, n_ml = 0;
    uint8_t *d;
    if (s->have_line) s->have_line = 0;
    else if (getline(&s->line, &s->cap, s->fp) <= 0) return -1;
    p = s->line;
    for (i = 0; i < 11; ++i) {
        f[i] = p;
        p += strcspn(p, "\t\n");
        if (*p == 0 && i < 10) return -2;
        if (*p) *p++ = 0;
    }
    /* optional fields: only the two base-modification tags are kept (BAM aux encoding: MM Z string NUL, ML B C count values) */
    while (*p) {
        char *t = p;
        p += strcspn(p, "\t\n");
        if (*p) *p++ = 0;
        if ((strncmp(t, "MM:Z:", 5) == 0 || strncmp(t, "Mm:Z:", 5) == 0) && !mm) { mm = t + 5; l_aux += 3 + strlen(mm) + 1; }
        else if ((strncmp(t, "ML:B:C", 6) == 0 || strncmp(t, "Ml:B:C", 6) == 0) && !ml) {
            char *q;
            ml = t + 6;
            for (q = ml; *q; ++q) if (*q == ',') ++n_ml;
            l_aux += 3 + 1 + 4 + n_ml;
        }
    }
    if (strcmp(f[5], "*") != 0) for (p = f[5]; *p; ++p) if (*p < '0' || *p > '9') ++n_cig;
    l_qn = strlen(f[0]) + 1; pad = (4 - (l_qn & 3)) & 3;
    l_seq = strcmp(f[9], "*") == 0 ? 0 : strlen(f[9]);
    need = l_qn + pad + 4 * (size_t)n_cig + (l_seq + 1) / 2 + l_seq + l_aux;
    if (b->m_data < need) { b->data = (uint8_t *)realloc(b->data, need); b->m_data = (uint32_t)need; }
    b->l_data = (int)need;
    b->core.tid = strcmp(f[2], "*") == 0 ? -1 : tid_of(s, f[2]);
    b->core.pos = atoll(f[3]) - 1;
    b->core.bin = 0; b->core.qual = (uint8_t)atoi(f[4]); b->core.l_extranul = (uint8_t)pad;
    b->core.flag = (uint16_t)atoi(f[1]); b->core.l_qname = (uint16_t)(l_qn + pad); b->core.n_cigar = (uint32_t)n_cig;
    b->core.l_qseq = (int32_t)l_seq;
    b->core.mtid = strcmp(f[6], "=") == 0 ? b->core.tid : (strcmp(f[6], "*") == 0 ? -1 : tid_of(s, f[6]));
    b->core.mpos = atoll(f[7]) - 1; b->core.isize = atoll(f[8]);
    d = b->data;
    memcpy(d, f[0], l_qn); memset(d + l_qn, 0, pad); d += l_qn + pad;
    if (n_cig) {
        uint32_t *cig = (uint32_t *)d;
        for (p = f[5], i = 0; *p; ++i) {
            char *q; unsigned long len = strtoul(p, &q, 10);
            const char *ops = "MIDNSHP=XB", *o = strchr(ops, *q);
            if (!o) return -2;
            cig[i] = (uint32_t)(len << 4 | (unsigned long)(o - ops));
            p = q + 1;
        }
        d += 4 * (size_t)n_cig;
    }
    memset(d, 0, (l_seq + 1) / 2);
    for (i = 0; i < (int)l_seq; ++i) {
        const char *o = strchr(NT16, f[9][i] >= 'a' && f[9][i] <= 'z' ? f[9][i] - 32 : f[9][i]);
        d[i >> 1] |= (uint8_t)((o ? (int)(o - NT16) : 15) << ((~i & 1) << 2));
    }
    d += (l_seq + 1) / 2;
    if (strcmp(f[10], "*") == 0) memset(d, 0xff, l_seq);
    else { if (strlen(f[10]) != l_seq) return -2; for (i = 0; i < (int)l_seq; ++i) d[i] = (uint8_t)(f[10][i] - 33); }
    d += l_seq;
    if (mm) { size_t l = strlen(mm); d[0] = 'M'; d[1] = 'M'; d[2] = 'Z'; memcpy(d + 3, mm, l + 1); d += 3 + l + 1; }
    if (ml) {
        uint32_t n32 = (uint32_t)n_ml;
        d[0] = 'M'; d[1] = 'L'; d[2] = 'B'; d[3] = 'C'; memcpy(d + 4, &n32, 4); d += 8;
        for (p = ml; *p == ','; ) { char *q; *d++ = (uint8_t)strtoul(p + 1, &q, 10); p = q; }
    }
    return 0;
}

static kstring_t g_ins = { 0, 0, NULL };

static void print_entries(FILE *out, int n, const bam_pileup1_t *plp)
{
    int i;
    fprintf(out, "\t%d", n);
    for (i = 0; i < n; ++i) {
        const bam_pileup1_t *p = &plp[i];
        int q = p->qpos < p->b->core.l_qseq ? bam_get_qual(p->b)[p->qpos] : -1;
        int del_len = 0;
        int il = bam_plp_insertion_mod(p, NULL, &g_ins, &del_len);      /* bam_plcmd.c:119 with no modification state */
        fprintf(out, "\t%s,%d,%d,%d,%d%d%d%d,%d,%d,%s,%d", bam_get_qname(p->b), p->b->core.flag, p->qpos, p->indel, (int)p->is_del,
                (int)p->is_head, (int)p->is_tail, (int)p->is_refskip, p->cigar_ind, q, il > 0 ? g_ins.s : ".", del_len);
    }
}

static int plbuf_cb(uint32_t tid, hts_pos_t pos, int n, const bam_pileup1_t *pl, void *data)
{
    FILE *out = (FILE *)data;
    fprintf(out, "%u\t%lld", tid, (long long)pos);
    print_entries(out, n, pl);
    fputc('\n', out);
    return 0;
}

/* ---- mpileup --output-mods through the drop-in names ---- */
static int mod_enter(void *data, const bam1_t *b, bam_pileup_cd *cd)
{
    hts_base_mod_state *m = hts_base_mod_state_alloc();
    (void)data;
    cd->p = m;
    return m ? bam_parse_basemod(b, m) : -1;
}
static int mod_leave(void *data, const bam1_t *b, bam_pileup_cd *cd) { (void)data; (void)b; hts_base_mod_state_free((hts_base_mod_state *)cd->p); return 0; }

static void put_mods(FILE *out, const hts_base_mod *mod, int nm)
{
    int j;
    fputc('[', out);
    for (j = 0; j < nm && j < 256; ++j) {
        if (mod[j].modified_base < 0) fprintf(out, "%c(%d)", "+-"[mod[j].strand], -mod[j].modified_base);
        else fprintf(out, "%c%c", "+-"[mod[j].strand], mod[j].modified_base);
        if (mod[j].qual >= 0) fprintf(out, "%d", mod[j].qual);
    }
    fputc(']', out);
}

/* one column as `mpileup -Q0 --output-mods [--no-output-ins-mods]` prints it without a reference: name, position, N, depth, bases, qualities */
static int print_mods_line(FILE *out, const char *name, hts_pos_t pos, int n, const bam_pileup1_t *plp, int no_ins_mods)
{
    int i, j;
    fprintf(out, "%s\t%lld\tN\t%d\t", name, (long long)pos + 1, n);
    if (n == 0) { fputs("*\t*\n", out); return 0; }
    for (i = 0; i < n; ++i) {
        const bam_pileup1_t *p = &plp[i];
        hts_base_mod_state *m = (hts_base_mod_state *)p->cd.p;
        const int rev = (p->b->core.flag & 16) != 0;
        int del_len = -p->indel;
        if (p->is_head) { fputc('^', out); fputc(p->b->core.qual > 93 ? 126 : p->b->core.qual + 33, out); }
        if (!p->is_del) {
            hts_base_mod mod[256];
            int nm, c = p->qpos < p->b->core.l_qseq ? bam_seqi(bam_get_seq(p->b), p->qpos) : 15;
            fputc(rev ? ",acmgrsvtwyhkdbn"[c] : ".ACMGRSVTWYHKDBN"[c], out);
            if (m && (nm = bam_mods_at_qpos(p->b, p->qpos, m, mod, 256)) > 0) put_mods(out, mod, nm);
        } else fputc(p->is_refskip ? (rev ? '<' : '>') : '*', out);
        if (p->indel > 0) {
            int in_mod = 0, len = bam_plp_insertion_mod(p, m && !no_ins_mods ? m : NULL, &g_ins, &del_len);
            if (len < 0) return -1;
            fprintf(out, "+%d", len);
            for (j = 0; j < (int)g_ins.l; ++j) {
                const char ch = g_ins.s[j];
                if (ch == '[') in_mod = 1; else if (ch == ']') in_mod = 0;
                fputc(in_mod || ch == '[' || ch == ']' ? ch : (rev ? tolower((unsigned char)ch) : toupper((unsigned char)ch)), out);
            }
        }
        if (del_len > 0) { fprintf(out, "-%d", del_len); for (j = 0; j < del_len; ++j) fputc(rev ? 'n' : 'N', out); }
        if (p->is_tail) fputc('$', out);
    }
    fputc('\t', out);
    for (i = 0; i < n; ++i) {
        const bam_pileup1_t *p = &plp[i];
        int q = p->qpos < p->b->core.l_qseq ? bam_get_qual(p->b)[p->qpos] : 0;
        fputc(q + 33 > 126 ? 126 : q + 33, out);
    }
    fputc('\n', out);
    return 0;
}

/* constructor / destructor hooks (bam_plcmd.c:356-369, bedcov.c:71-75): count the reads that enter and leave */
static long g_ctor = 0, g_dtor = 0;
static int on_enter(void *data, const bam1_t *b, bam_pileup_cd *cd) { (void)data; (void)b; cd->i = ++g_ctor; return 0; }
static int on_leave(void *data, const bam1_t *b, bam_pileup_cd *cd) { (void)data; (void)b; if (cd->i > 0) ++g_dtor; return 0; }

int main(int argc, char **argv)
{
    int overlaps = 1, push = 0, maxcnt = 8000, a = 1, n, i, ret = 0, mods = 0;
    src_t *src; void **data;
    for (; a < argc && argv[a][0] == '-' && argv[a][1]; ++a) {
        if (!strcmp(argv[a], "-x")) overlaps = 0;
        else if (!strcmp(argv[a], "-p")) push = 1;
        else if (!strcmp(argv[a], "-M")) mods = 1;
        else if (!strcmp(argv[a], "-N")) mods = 2;
        else if (!strcmp(argv[a], "-d") && a + 1 < argc) maxcnt = atoi(argv[++a]);
        else return 2;
    }
    n = argc - a;
    if (n <= 0) { fprintf(stderr, "usage: plp_client [-x] [-d maxcnt] [-p] in.sam [...]\n"); return 2; }
    src = (src_t *)calloc((size_t)n, sizeof *src);
    data = (void **)calloc((size_t)n, sizeof *data);
    for (i = 0; i < n; ++i) {
        if (open_src(&src[i], argv[a + i]) < 0) { fprintf(stderr, "plp_client: cannot open %s\n", argv[a + i]); return 2; }
        data[i] = &src[i];
    }
    if (push) {
        bam_plbuf_t *buf = bam_plbuf_init(plbuf_cb, stdout);
        bam1_t b; int r;
        memset(&b, 0, sizeof b);
        if (!buf) { fprintf(stderr, "plp_client: bam_plbuf_init failed (no HIP device?)\n"); return 3; }
        while ((r = read_cb(&src[0], &b)) >= 0)
            if (bam_plbuf_push(&b, buf) < 0) { ret = 1; break; }
        if (r < -1) ret = 1;
        if (!ret && bam_plbuf_push(NULL, buf) < 0) ret = 1;
        bam_plbuf_destroy(buf);
        free(b.data);
    } else {
        bam_mplp_t it = bam_mplp_init(n, read_cb, data);
        int *n_plp = (int *)calloc((size_t)n, sizeof(int));
        const bam_pileup1_t **plp = (const bam_pileup1_t **)calloc((size_t)n, sizeof *plp);
        int tid = 0, r; hts_pos_t pos = 0;
        if (!it) { fprintf(stderr, "plp_client: bam_mplp_init failed (no HIP device?)\n"); return 3; }
        if (overlaps) bam_mplp_init_overlaps(it);
        bam_mplp_set_maxcnt(it, maxcnt);
        bam_mplp_constructor(it, mods ? mod_enter : on_enter);
        bam_mplp_destructor(it, mods ? mod_leave : on_leave);
        while ((r = bam_mplp64_auto(it, &tid, &pos, n_plp, plp)) > 0) {
            if (mods) {
                if (tid < 0 || tid >= src[0].n_names || print_mods_line(stdout, src[0].names[tid], pos, n_plp[0], plp[0], mods == 2) < 0) { ret = 1; break; }
                continue;
            }
            printf("%d\t%lld", tid, (long long)pos);
            for (i = 0; i < n; ++i) print_entries(stdout, n_plp[i], plp[i]);
            fputc('\n', stdout);
        }
        if (r < 0) { fprintf(stderr, "plp_client: error reading from input file\n"); ret = 1; }
        bam_mplp_destroy(it);
        if (!ret && g_ctor != g_dtor) { fprintf(stderr, "plp_client: %ld constructor calls but %ld destructor calls\n", g_ctor, g_dtor); ret = 4; }
        free(n_plp); free(plp);
    }
    free(g_ins.s);
    return ret;
}
