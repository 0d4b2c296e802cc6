/*
 * oracle/o_depth.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of samtools bam2depth.c: zero_region :88-118, qlen_used
 * :124-159, incr_hist[_qual] :165-195, add_depth :209-477, fastdepth_core
 * :486-699, main_depth :732-1006.  Pinned by test/mpileup/depth.reg goldens
 * and test/large_pos/depth*.expected.out.
 */
#include "o_plp.h"
#include <getopt.h>
#include <errno.h>
#include <limits.h>

#define MIN(a,b) ((a)<(b)?(a):(b))
#define MAX(a,b) ((a)>(b)?(a):(b))

typedef struct {
    size_t size;
    int **hist;
    hpos_t *end_pos;
    hpos_t last_output;
    int last_ref;
    int nfiles;
    const char *ref;
    ostr_t ks;
    hpos_t beg, end;
    int tid;
} depth_hist;

typedef struct {
    int header, flag, incl_flag, require_flag, min_qual, min_mqual, min_len, skip_del, all_pos, remove_overlaps;
    FILE *out;
    char *reg;
    obed_t *bed;
} depth_opt;

/* bam2depth.c:88-118 */
static void zero_region(depth_opt *opt, depth_hist *dh, const char *name, hpos_t start, hpos_t end)
{
    hpos_t i;
    ostr_t *ks = &dh->ks;
    os_clear(ks); os_puts(ks, name); os_putc(ks, '\t');
    size_t cur_l = ks->l;
    if (dh->beg >= 0 && start < dh->beg) start = dh->beg;
    if (dh->end >= 0 && end > dh->end) end = dh->end;
    for (i = start; i < end; i++) {
        if (opt->bed && bed_olap(opt->bed, name, i, i + 1) == 0) continue;
        ks->l = cur_l;
        os_putll(ks, i + 1);
        for (int n = 0; n < dh->nfiles; n++) { os_putc(ks, '\t'); os_putc(ks, '0'); }
        os_putc(ks, '\n');
        fputs(ks->s, opt->out);
    }
    ks->l = cur_l;
}

/* bam2depth.c:124-159 */
static hpos_t qlen_used(orec_t *b)
{
    int n_cigar = (int)b->n_cigar;
    const uint32_t *cigar = b->cigar;
    hpos_t l;
    if (b->l_qseq) {
        l = b->l_qseq;
        int kl, kr;
        for (kl = 0; kl < n_cigar; kl++)
            if (cig_op(cigar[kl]) == C_S) l -= cig_len(cigar[kl]); else break;
        for (kr = n_cigar - 1; kr > kl; kr--)
            if (cig_op(cigar[kr]) == C_S) l -= cig_len(cigar[kr]); else break;
    } else {
        static const int query[16] = { 1,1,0,0, 0,0,0,1, 1,0,0,0, 0,0,0,0 };
        int k;
        for (k = 0, l = 0; k < n_cigar; k++)
            if (query[cig_op(cigar[k])]) l += cig_len(cigar[k]);
    }
    return l;
}

/* bam2depth.c:165-195 */
static inline void incr_hist_qual(int *hist, uint8_t *qual, int min_qual, int oplen)
{
    int k;
    if (!min_qual) { for (k = 0; k < oplen; k++) hist[k]++; return; }
    for (k = 0; k < oplen; k++) hist[k] += qual[k] >= min_qual;
}

static void flush_rows(depth_opt *opt, depth_hist *dh, hpos_t *pi, hpos_t limit, int bounded)
{
    /* shared body of bam2depth.c:219-245 and :291-316 */
    size_t hmask = dh->size - 1;
    size_t cur_l = dh->ks.l;
    int nf = dh->nfiles, n;
    hpos_t i;
    for (i = dh->last_output; bounded ? i < limit : nf; i++) {
        nf = 0;
        for (n = 0; n < dh->nfiles; n++) if (i < dh->end_pos[n]) nf++;
        if (!nf) break;
        if (opt->bed && bed_olap(opt->bed, dh->ref, i, i + 1) == 0) continue;
        dh->ks.l = cur_l;
        os_putll(&dh->ks, i + 1);
        for (n = 0; n < dh->nfiles; n++) {
            os_putc(&dh->ks, '\t');
            int d = i < dh->end_pos[n] ? dh->hist[n][(size_t)i & hmask] : 0;
            os_putll(&dh->ks, (unsigned)d);
        }
        os_putc(&dh->ks, '\n');
        fputs(dh->ks.s, opt->out);
    }
    dh->ks.l = cur_l;
    *pi = i;
}

/* bam2depth.c:209-477 */
static int add_depth(depth_opt *opt, depth_hist *dh, ohdr_t *h, orec_t *b, hpos_t overlap_clip, int file)
{
    hpos_t i;
    size_t hmask = dh->size - 1;
    int n;

    if (!b || b->tid != dh->last_ref) {
        if (dh->last_ref >= 0) {
            flush_rows(opt, dh, &i, 0, 0);
            if (opt->all_pos)
                zero_region(opt, dh, h->name[dh->last_ref], i, h->len[dh->last_ref]);
        }
        if (opt->all_pos > 1 && !opt->reg) {
            int lr = dh->last_ref < 0 ? 0 : dh->last_ref + 1;
            int rr = b ? b->tid : h->n_ref, r;
            for (r = lr; r < rr; r++) zero_region(opt, dh, h->name[r], 0, h->len[r]);
        }
        if (!b) {
            if (opt->all_pos && opt->reg && dh->last_ref < 0)
                zero_region(opt, dh, h->name[dh->tid], dh->beg, MIN(dh->end, h->len[dh->tid]));
            return 0;
        }
        for (n = 0; dh->end_pos && n < dh->nfiles; n++) dh->end_pos[n] = 0;
        dh->last_output = dh->beg >= 0 ? MAX(b->pos, dh->beg) : b->pos;
        dh->last_ref = b->tid;
        dh->ref = h->name[b->tid];
        os_clear(&dh->ks); os_puts(&dh->ks, dh->ref); os_putc(&dh->ks, '\t');
        if (opt->all_pos) zero_region(opt, dh, dh->ref, 0, b->pos);
        /* zero_region resets ks to "name\t" */
    } else {
        if (dh->last_output < b->pos) {
            flush_rows(opt, dh, &i, b->pos, 1);
            if (opt->all_pos && i < b->pos) zero_region(opt, dh, dh->ref, i, b->pos);
            dh->last_output = b->pos;
        }
    }

    hpos_t end_pos = rec_endpos(b);
    if (b->tid < dh->last_ref || (dh->last_ref == b->tid && end_pos < dh->last_output)) {
        fflush(stdout);
        fprintf(stderr, "samtools depth: Data is not position sorted\n");
        return -1;
    }

    if ((size_t)(end_pos + 1 - b->pos) >= dh->size) {
        size_t old_size = dh->size;
        size_t old_hmask = hmask;
        while ((size_t)(end_pos + 1 - b->pos) >= dh->size) dh->size = dh->size ? 2 * dh->size : 2048;
        hmask = dh->size - 1;
        if (!dh->hist) {
            dh->hist = (int **)calloc((size_t)dh->nfiles, sizeof(*dh->hist));
            dh->end_pos = (hpos_t *)calloc((size_t)dh->nfiles, sizeof(*dh->end_pos));
        }
        for (n = 0; n < dh->nfiles; n++) {
            int *hist = (int *)calloc(dh->size, sizeof(int));
            if (dh->hist[n])
                for (i = dh->last_output; i < dh->last_output + (hpos_t)old_size; i++)
                    hist[(size_t)i & hmask] = dh->hist[n][(size_t)i & old_hmask];
            free(dh->hist[n]);
            dh->hist[n] = hist;
        }
    }

    uint32_t *cig = b->cigar;
    int ncig = (int)b->n_cigar, j, k, spos = 0;
    hpos_t end = MAX(dh->end_pos[file], b->pos);
    for (i = end; i < end_pos; i++) dh->hist[file][(size_t)i & hmask] = 0;

    i = b->pos;
    uint8_t *qual = b->qual;
    int min_qual = opt->min_qual;
    for (j = 0; j < ncig; j++) {
        int op = cig_op(cig[j]);
        int oplen = (int)cig_len(cig[j]);
        switch (op) {
        case C_D: case C_N:
            if (op != C_D || opt->skip_del) {
                if (i + oplen >= dh->end_pos[file]) {
                    for (k = 0; k < oplen; k++, i++)
                        if (i >= dh->end_pos[file]) dh->hist[file][(size_t)i & hmask] = 0;
                } else i += oplen;
            } else {
                int *hist = dh->hist[file];
                k = 0;
                if (overlap_clip) {
                    if (i + oplen <= overlap_clip) { i += oplen; break; }
                    else if (i < overlap_clip) { k = (int)(overlap_clip - i); i = overlap_clip; }
                }
                if (spos < b->l_qseq)
                    for (; k < oplen; k++, i++) hist[(size_t)i & hmask] += qual[spos] >= min_qual;
                else
                    for (; k < oplen; k++, i++) hist[(size_t)i & hmask]++;
            }
            break;
        case C_M: case C_EQ: case C_X: {
            int *hist = dh->hist[file];
            if (overlap_clip) {
                if (i + oplen <= overlap_clip) { i += oplen; spos += oplen; break; }
                else if (i < overlap_clip) {
                    oplen -= (int)(overlap_clip - i);
                    spos += (int)(overlap_clip - i);
                    i = overlap_clip;
                }
            }
            int len = ((size_t)i & hmask) < ((size_t)(i + oplen) & hmask) ? oplen : (int)(dh->size - ((size_t)i & hmask));
            incr_hist_qual(&hist[(size_t)i & hmask], &qual[spos], min_qual, len);
            if (oplen > len) incr_hist_qual(hist, &qual[spos + len], min_qual, oplen - len);
            spos += oplen;
            i += oplen;
            break;
        }
        case C_I: case C_S: spos += oplen; break;
        case C_P: case C_H: break;
        default:
            fflush(stdout);
            fprintf(stderr, "samtools depth: Unsupported cigar op '%d'\n", op);
            return -1;
        }
    }
    if (dh->end >= 0 && end_pos > dh->end) end_pos = dh->end;
    if (dh->end_pos[file] < end_pos) dh->end_pos[file] = end_pos;
    return 0;
}

/* name -> end hash for -s (bam2depth.c:479-484) */
typedef struct nent { char *key; hpos_t val; struct nent *next; } nent_t;
typedef struct { nent_t **tab; size_t nb, n; } nhash_t;
static uint32_t nh_hash(const char *s) { uint32_t h = (uint32_t)(unsigned char)*s; if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)(unsigned char)*s; return h; }
static nhash_t *nh_init(void) { nhash_t *h = (nhash_t *)calloc(1, sizeof *h); h->nb = 1 << 16; h->tab = (nent_t **)calloc(h->nb, sizeof(nent_t *)); return h; }
static nent_t **nh_find(nhash_t *h, const char *k) { nent_t **pp = &h->tab[nh_hash(k) & (h->nb - 1)]; while (*pp && strcmp((*pp)->key, k)) pp = &(*pp)->next; return pp; }
static void nh_destroy(nhash_t *h)
{
    for (size_t i = 0; i < h->nb; ++i) for (nent_t *e = h->tab[i], *nx; e; e = nx) { nx = e->next; free(e->key); free(e); }
    free(h->tab); free(h);
}

static int read_filtered(depth_opt *opt, oreader_t *fp, orec_t *b)
{
    /* bam2depth.c:540-573 / :632-663 */
    for (;;) {
        int ret = rd_next(fp, b);
        if (ret < -1) return ret;
        if (ret == -1) return -1;
        if (b->tid < 0) continue;
        if (b->flag & opt->flag) continue;
        if (opt->incl_flag && (b->flag & opt->incl_flag) == 0) continue;
        if ((b->flag & opt->require_flag) != opt->require_flag) continue;
        if (b->mapq < opt->min_mqual) continue;
        if (opt->min_len && qlen_used(b) < opt->min_len) continue;
        return 0;
    }
}

/* bam2depth.c:486-699 */
static int fastdepth_core(depth_opt *opt, int nfiles, char **fn, oreader_t **fp, int has_itr,
                          int itr_tid, hpos_t itr_beg, hpos_t itr_end, ohdr_t **h)
{
    int ret = -1, err = 1, i;
    nhash_t **overlaps = NULL;
    depth_hist dh; memset(&dh, 0, sizeof dh);
    orec_t *b = (orec_t *)calloc((size_t)nfiles, sizeof(*b));
    int *finished = (int *)calloc((size_t)nfiles, sizeof(int)), to_go = nfiles;

    if (opt->remove_overlaps) {
        overlaps = (nhash_t **)calloc((size_t)nfiles, sizeof(*overlaps));
        for (i = 0; i < nfiles; i++) overlaps[i] = nh_init();
    }
    dh.nfiles = nfiles;
    dh.last_ref = -99;
    dh.last_output = has_itr ? itr_beg : 0;
    dh.beg = -1; dh.end = -1; dh.tid = 0;
    if (has_itr) { dh.tid = itr_tid; dh.beg = itr_beg; dh.end = itr_end; }

    if (opt->header) {
        fprintf(opt->out, "#CHROM\tPOS");
        for (i = 0; i < nfiles; i++) fprintf(opt->out, "\t%s", fn[i]);
        fputc('\n', opt->out);
    }
    for (i = 0; i < nfiles; i++) {
        ret = read_filtered(opt, fp[i], &b[i]);
        if (ret < -1) goto err;
        if (ret == -1) { to_go--; finished[i] = 1; }
    }
    while (to_go) {
        int best_tid = INT_MAX, best_file = 0;
        hpos_t best_pos = HPOS_MAX;
        for (i = 0; i < nfiles; i++) {
            if (finished[i]) continue;
            if (best_tid > b[i].tid) { best_tid = b[i].tid; best_pos = b[i].pos; best_file = i; }
            else if (best_tid == b[i].tid && best_pos > b[i].pos) { best_pos = b[i].pos; best_file = i; }
        }
        i = best_file;
        hpos_t clip = 0;
        if (overlaps && (b[i].flag & F_PAIRED) && !(b[i].flag & F_MUNMAP)) {
            nent_t **pp = nh_find(overlaps[i], b[i].qname);
            if (!*pp) {
                hpos_t endpos = rec_endpos(&b[i]);
                if (b[i].mpos == -1 || (b[i].tid == b[i].mtid && b[i].mpos <= endpos)) {
                    nent_t *e = (nent_t *)malloc(sizeof *e);
                    e->key = strdup(b[i].qname); e->val = endpos; e->next = NULL;
                    *pp = e;
                }
            } else {
                nent_t *e = *pp;
                clip = e->val;
                *pp = e->next; free(e->key); free(e);
            }
        }
        if ((ret = add_depth(opt, &dh, h[i], &b[i], clip, i)) < 0) { ret = -1; goto err; }
        ret = read_filtered(opt, fp[i], &b[i]);
        if (ret < -1) { ret = -1; goto err; }
        if (ret == -1) { to_go--; finished[i] = 1; }
    }
    ret = add_depth(opt, &dh, h[0], NULL, 0, 0);
    err = 0;
err:
    if (ret == 0 && err) ret = -1;
    for (i = 0; i < nfiles; i++) { rec_free(&b[i]); if (dh.hist && dh.hist[i]) free(dh.hist[i]); }
    free(b); free(finished); free(dh.ks.s); free(dh.hist); free(dh.end_pos);
    if (overlaps) { for (i = 0; i < nfiles; i++) nh_destroy(overlaps[i]); free(overlaps); }
    return ret;
}

int o_read_file_list(const char *file_list, int *n, char ***argv);

/* bam2depth.c:732-1006 */
int o_main_depth(int argc, char *argv[])
{
    int nfiles, i, tmp_flag, c;
    char *file_list = NULL, **fn = NULL;
    depth_opt opt;
    memset(&opt, 0, sizeof opt);
    opt.flag = F_UNMAP | F_SECONDARY | F_DUP | F_QCFAIL;
    opt.skip_del = 1;
    opt.out = stdout;

    static const struct option lopts[] = {
        { "min-MQ", required_argument, NULL, 'Q' }, { "min-mq", required_argument, NULL, 'Q' },
        { "min-BQ", required_argument, NULL, 'q' }, { "min-bq", required_argument, NULL, 'q' },
        { "excl-flags", required_argument, NULL, 'G' }, { "incl-flags", required_argument, NULL, 1 },
        { "require-flags", required_argument, NULL, 2 }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "@:q:Q:JHd:m:l:g:G:o:ar:Xf:b:s", lopts, NULL)) >= 0) {
        switch (c) {
        case 'a': opt.all_pos++; break;
        case 'b':
            opt.bed = bed_load(optarg);
            if (!opt.bed) { fprintf(stderr, "samtools depth: Could not read file \"%s\"\n", optarg); return 1; }
            break;
        case 'f': file_list = optarg; break;
        case 'd': case 'm': case '@': break;
        case 'g': tmp_flag = str2flag(optarg); if (tmp_flag < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.flag &= ~tmp_flag; break;
        case 'G': tmp_flag = str2flag(optarg); if (tmp_flag < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.flag |= tmp_flag; break;
        case 1: tmp_flag = str2flag(optarg); if (tmp_flag < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.incl_flag |= tmp_flag; break;
        case 2: tmp_flag = str2flag(optarg); if (tmp_flag < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.require_flag |= tmp_flag; break;
        case 'l': opt.min_len = atoi(optarg); break;
        case 'H': opt.header = 1; break;
        case 'q': opt.min_qual = atoi(optarg); break;
        case 'Q': opt.min_mqual = atoi(optarg); break;
        case 'J': opt.skip_del = 0; break;
        case 'o':
            if (opt.out != stdout) break;
            opt.out = fopen(optarg, "w");
            if (!opt.out) { fprintf(stderr, "samtools depth: Cannot open \"%s\" for writing.\n", optarg); return EXIT_FAILURE; }
            break;
        case 'r': opt.reg = optarg; break;
        case 's': opt.remove_overlaps = 1; break;
        case 'X': fprintf(stderr, "oracle: -X not supported\n"); return 1;
        default: fprintf(stderr, "Usage: samtools depth [options] in.bam [in.bam ...]\n"); return 1;
        }
    }
    if (argc < optind + 1 && !file_list) { fprintf(stderr, "Usage: samtools depth [options] in.bam [in.bam ...]\n"); return argc == optind ? 0 : 1; }
    if (file_list) {
        if (o_read_file_list(file_list, &nfiles, &fn)) return 1;
        argv = fn; argc = nfiles; optind = 0;
    } else nfiles = argc - optind;

    oreader_t **fp = (oreader_t **)malloc((size_t)nfiles * sizeof(*fp));
    ohdr_t **header = (ohdr_t **)malloc((size_t)nfiles * sizeof(*header));
    int has_itr = 0, itr_tid = 0; hpos_t itr_beg = 0, itr_end = 0;
    char **names = &argv[optind];
    for (i = 0; i < nfiles; i++) {
        fp[i] = rd_open(names[i]);
        if (!fp[i]) { fprintf(stderr, "samtools depth: Cannot open input file \"%s\": %s\n", names[i], strerror(errno)); return 1; }
        header[i] = rd_header(fp[i]);
        if (opt.reg) {
            int t; hpos_t bb, ee;
            if (parse_region(header[i], opt.reg, &t, &bb, &ee) < 0) {
                fprintf(stderr, "samtools depth: cannot parse region \"%s\"\n", opt.reg);
                return 1;
            }
            rd_set_region(fp[i], t, bb, ee);
            if (i == 0) { has_itr = 1; itr_tid = t; itr_beg = bb; itr_end = ee; }
        }
    }
    int ret = fastdepth_core(&opt, nfiles, names, fp, has_itr, itr_tid, itr_beg, itr_end, header) ? 1 : 0;
    for (i = 0; i < nfiles; i++) rd_close(fp[i]);
    free(header); free(fp);
    if (file_list) { for (i = 0; i < nfiles; i++) free(fn[i]); free(fn); }
    if (opt.bed) bed_free(opt.bed);
    if (opt.out != stdout) fclose(opt.out);
    return ret;
}
