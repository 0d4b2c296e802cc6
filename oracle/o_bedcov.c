/*
 * oracle/o_bedcov.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of `samtools bedcov` (bedcov.c:110-380): for every BED interval a fresh pileup
 * iterator over the reads overlapping it (sam_itr_queryi stand-in: rd_set_region), read filter
 * read_bam (bedcov.c:54-70), per-position sum of n_plp (minus deletions / ref skips with -j,
 * bedcov.c:316-333), optional depth-threshold and read-count columns, optional header (:81-108).
 * Pinned by test/bedcov/{bedcov,bedcov_j,bedcov_gG,bedcov_c}.expected and the -H cases of
 * test/test.pl:3817-3868.
 */
#include "o_plp.h"
#include <getopt.h>
#include <limits.h>
#include <ctype.h>

typedef struct { oreader_t *rd; int min_mapQ; uint32_t flags; int64_t rcnt; } baux_t;

static int read_bam(void *data, orec_t *b)
{
    baux_t *aux = (baux_t *)data;
    int ret;
    while (1) {
        ret = rd_next(aux->rd, b);
        if (ret < 0) break;
        if (b->flag & aux->flags) continue;
        if ((int)b->mapq < aux->min_mapQ) continue;
        break;
    }
    return ret;
}

static int incr_rcnt(void *data, const orec_t *b, void *cd) { ((baux_t *)data)->rcnt++; return 0; }

static void output_header(FILE *fp, const char *hdr, int fields, int n, char **fn, int depth, int rcount)
{
    static const char *bedcols[] = { "chrom", "chromStart", "chromEnd", "name", "score", "strand", "thickStart", "thickEnd",
                                     "itemRgb", "blockCount", "blockSizes", "blockStarts" };
    int i;
    if (hdr) fprintf(fp, "%s", hdr);
    else for (i = 0; i < fields; ++i) fprintf(fp, "%s%s", (i ? "\t" : "#"), (i < 12 ? bedcols[i] : "."));
    for (i = 0; i < n; ++i) fprintf(fp, "\t%s_cov", fn[i]);
    if (depth >= 0) for (i = 0; i < n; ++i) fprintf(fp, "\t%s_depth", fn[i]);
    if (rcount) for (i = 0; i < n; ++i) fprintf(fp, "\t%s_count", fn[i]);
    fprintf(fp, "\n");
}

int o_main_bedcov(int argc, char *argv[])
{
    int c, n, i, j, status = 0, min_mapQ = 0, skip_DN = 0, do_rcount = 0, tflags, min_depth = -1, max_depth = INT_MAX, print_header = 0, hdr = 0;
    uint32_t flags = F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP;
    static const struct option lopts[] = { { "min-MQ", required_argument, NULL, 'Q' }, { "min-mq", required_argument, NULL, 'Q' },
                                           { "max-depth", required_argument, NULL, 'd' + 1000 }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "Q:g:G:jd:Hc", lopts, NULL)) >= 0) {
        switch (c) {
        case 'Q': min_mapQ = atoi(optarg); break;
        case 'c': do_rcount = 1; break;
        case 'H': print_header = 1; break;
        case 'g': tflags = str2flag(optarg); if (tflags < 0 || tflags > ((F_SUPPLEMENTARY << 1) - 1)) return 1; flags &= ~(uint32_t)tflags; break;
        case 'G': tflags = str2flag(optarg); if (tflags < 0 || tflags > ((F_SUPPLEMENTARY << 1) - 1)) return 1; flags |= (uint32_t)tflags; break;
        case 'j': skip_DN = 1; break;
        case 'd': min_depth = atoi(optarg); break;
        case 'd' + 1000: max_depth = atoi(optarg); break;
        default: return 1;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, "Usage: oracle_samtools bedcov [options] <in.bed> <in1.bam> [...]\n"); return 1; }
    n = argc - optind - 1;
    char **fn = argv + optind + 1;
    if (!print_header) hdr = 1;
    baux_t *aux = (baux_t *)calloc((size_t)n, sizeof(baux_t));
    void **data = (void **)calloc((size_t)n, sizeof(void *));
    oreader_t *r0 = rd_open(fn[0]);
    if (!r0) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[0]); return 2; }
    ohdr_t *h0 = rd_header(r0);
    int64_t *cnt = (int64_t *)calloc((size_t)n, 8), *pcov = (int64_t *)calloc((size_t)n, 8);
    int *n_plp = (int *)calloc((size_t)n, sizeof(int));
    const opileup1_t **plp = (const opileup1_t **)calloc((size_t)n, sizeof(*plp));
    FILE *fp = fopen(argv[optind], "r");
    if (!fp) { fprintf(stderr, "samtools bedcov: can't open BED file '%s'\n", argv[optind]); return 2; }
    char *line = NULL; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, fp)) >= 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (len == 0) continue;
        if (line[0] == '#') {
            if (!hdr && !strncmp(line, "#chrom", 6)) { output_header(stdout, line, -1, n, fn, min_depth, do_rcount); hdr = 1; }
            continue;
        }
        if (strncmp(line, "track ", 6) == 0 || strncmp(line, "browser ", 8) == 0) continue;
        if (!hdr) {
            int fields = 0;
            for (char *t = line; *t; ++t) if (*t == '\t') fields++;
            output_header(stdout, NULL, fields + 1, n, fn, min_depth, do_rcount);
            hdr = 1;
        }
        char *p, *q;
        for (p = q = line; *p && !isspace((unsigned char)*p); ++p);
        if (*p == 0) goto bed_error;
        {
            char ch = *p; *p = 0;
            int tid = hdr_name2tid(h0, q);
            *p = ch;
            long long beg = 0, end = 0;
            if (tid < 0) goto bed_error;
            if (sscanf(p + 1, "%lld %lld", &beg, &end) < 2 || end < beg) goto bed_error;
            for (i = 0; i < n; ++i) {
                aux[i].rd = rd_open(fn[i]);
                if (!aux[i].rd) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[i]); return 2; }
                rd_set_region(aux[i].rd, tid, beg, end);
                aux[i].min_mapQ = min_mapQ; aux[i].flags = flags; aux[i].rcnt = 0;
                data[i] = &aux[i];
            }
            omplp_t *mplp = omplp_init(n, read_bam, data);
            omplp_set_maxcnt(mplp, min_depth > max_depth ? min_depth : max_depth);
            memset(cnt, 0, 8 * (size_t)n); memset(pcov, 0, 8 * (size_t)n);
            if (do_rcount) omplp_constructor(mplp, incr_rcnt);
            int ptid, ret; hpos_t pos;
            while ((ret = omplp_auto(mplp, &ptid, &pos, n_plp, plp)) > 0)
                if (pos >= beg && pos < end) {
                    for (i = 0; i < n; ++i) {
                        int m = 0;
                        if (skip_DN || min_depth >= 0)
                            for (j = 0; j < n_plp[i]; ++j) if (plp[i][j].is_del || plp[i][j].is_refskip) ++m;
                        int pd = n_plp[i] - m;
                        cnt[i] += pd;
                        if (min_depth >= 0 && pd >= min_depth) pcov[i]++;
                    }
                }
            if (ret < 0) { fprintf(stderr, "samtools bedcov: error reading from input file\n"); status = 2; omplp_destroy(mplp); break; }
            fputs(line, stdout);
            for (i = 0; i < n; ++i) printf("\t%lld", (long long)cnt[i]);
            if (min_depth >= 0) for (i = 0; i < n; ++i) printf("\t%lld", (long long)pcov[i]);
            if (do_rcount) for (i = 0; i < n; ++i) printf("\t%lld", (long long)aux[i].rcnt);
            putchar('\n');
            omplp_destroy(mplp);
            for (i = 0; i < n; ++i) rd_close(aux[i].rd);
        }
        continue;
bed_error:
        fprintf(stderr, "Errors in BED line '%s'\n", line);
        status = 2;
    }
    free(line); fclose(fp); rd_close(r0);
    free(cnt); free(pcov); free(n_plp); free(plp); free(aux); free(data);
    return status;
}
