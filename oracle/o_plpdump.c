/*
 * oracle/o_plpdump.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Prints every pileup entry the restated HTSlib iterator (o_plp.c: bam_plp_* / bam_mplp_*,
 * SURVEY.md Appendix A.1-A.3) produces, in the format of `samtools-amd plpdump`, so that the
 * engine's bam_plp_* surface (include/samtools_amd_plp.h) can be diffed against it.
 *   plpdump [-x] [-d maxcnt] [-p] in1.sam [in2.sam ...]
 */
#include "o_plp.h"
#include <getopt.h>

typedef struct { oreader_t *rd; } src_t;

static int read_cb(void *data, orec_t *b)
{
    src_t *s = (src_t *)data;
    return rd_next(s->rd, b);
}

static ostr_t g_ins;

static void print_entries(FILE *out, int n, const opileup1_t *plp)
{
    fprintf(out, "\t%d", n);
    for (int i = 0; i < n; ++i) {
        const opileup1_t *p = &plp[i];
        int q = p->qpos < p->b->l_qseq ? p->b->qual[p->qpos] : -1;
        int del_len = 0;
        int il = oplp_insertion(p, &g_ins, &del_len);
        fprintf(out, "\t%s,%d,%d,%d,%d%d%d%d,%d,%d,%s,%d", p->b->qname, p->b->flag, p->qpos, p->indel, (int)p->is_del, (int)p->is_head,
                (int)p->is_tail, (int)p->is_refskip, p->cigar_ind, q, il > 0 ? g_ins.s : ".", del_len);
    }
}

int o_main_plpdump(int argc, char *argv[])
{
    int overlaps = 1, push = 0, maxcnt = 8000, c;
    optind = 1;
    while ((c = getopt(argc, argv, "xd:p")) >= 0) {
        if (c == 'x') overlaps = 0;
        else if (c == 'd') maxcnt = atoi(optarg);
        else if (c == 'p') push = 1;
        else return 1;
    }
    int n = argc - optind;
    if (n <= 0) { fprintf(stderr, "usage: oracle_samtools plpdump [-x] [-d maxcnt] [-p] in.sam [...]\n"); return 1; }
    src_t *src = (src_t *)calloc((size_t)n, sizeof(src_t));
    void **data = (void **)calloc((size_t)n, sizeof(void *));
    for (int i = 0; i < n; ++i) {
        src[i].rd = rd_open(argv[optind + i]);
        if (!src[i].rd) { fprintf(stderr, "[plpdump] failed to open %s\n", argv[optind + i]); return 1; }
        data[i] = &src[i];
    }
    FILE *out = stdout;
    int ret = 0;
    if (push) {
        oplp_t *it = oplp_init(NULL, NULL);
        oplp_set_maxcnt(it, 8000);
        orec_t b; memset(&b, 0, sizeof b);
        int r, tid, n_plp; hpos_t pos;
        const opileup1_t *plp;
        for (;;) {
            r = rd_next(src[0].rd, &b);
            if (r < -1) { ret = 1; break; }
            if (oplp_push(it, r >= 0 ? &b : NULL) < 0) { ret = 1; break; }
            while ((plp = oplp_next(it, &tid, &pos, &n_plp)) != 0) {
                fprintf(out, "%d\t%lld", tid, (long long)pos);
                print_entries(out, n_plp, plp);
                fputc('\n', out);
            }
            if (n_plp < 0) { ret = 1; break; }
            if (r < 0) break;
        }
        rec_free(&b);
        oplp_destroy(it);
    } else {
        omplp_t *it = omplp_init(n, read_cb, data);
        if (overlaps) omplp_init_overlaps(it);
        omplp_set_maxcnt(it, maxcnt);
        int *n_plp = (int *)calloc((size_t)n, sizeof(int));
        const opileup1_t **plp = (const opileup1_t **)calloc((size_t)n, sizeof(*plp));
        int tid, r; hpos_t pos;
        while ((r = omplp_auto(it, &tid, &pos, n_plp, plp)) > 0) {
            fprintf(out, "%d\t%lld", tid, (long long)pos);
            for (int i = 0; i < n; ++i) print_entries(out, n_plp[i], plp[i]);
            fputc('\n', out);
        }
        if (r < 0) { fprintf(stderr, "[plpdump] error reading from input file\n"); ret = 1; }
        omplp_destroy(it);
        free(n_plp); free(plp);
    }
    for (int i = 0; i < n; ++i) rd_close(src[i].rd);
    free(src); free(data); free(g_ins.s);
    return ret;
}
