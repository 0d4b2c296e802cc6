/*
 * oracle/o_coverage.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of `samtools coverage`, tabular mode (coverage.c:176-221 read_bam + print_tabular_line,
 * :572-700 the multi-pileup loop).  Histogram / plot modes (-m -A -D -w) are not restated.
 * Pinned by test/coverage/{1..5}.expected (test/test.pl:4143-4161).
 */
#include "o_plp.h"
#include <getopt.h>
#include <limits.h>

typedef struct {
    unsigned long long n_covered_bases, summed_coverage, summed_baseQ, summed_mapQ, quality_bases;
    unsigned int n_reads, n_selected_reads;
    int covered;
    hpos_t beg, end;
} cstats_t;

typedef struct { oreader_t *rd; ohdr_t *hdr; int min_mapQ, min_len, fail_flags, required_flags; cstats_t *stats; } caux_t;

static int cigar2qlen(const orec_t *b)
{
    int l = 0;
    for (uint32_t k = 0; k < b->n_cigar; ++k) {
        int op = cig_op(b->cigar[k]);
        if (op == C_M || op == C_I || op == C_S || op == C_EQ || op == C_X) l += (int)cig_len(b->cigar[k]);
    }
    return l;
}

static int read_bam(void *data, orec_t *b)
{
    caux_t *aux = (caux_t *)data;
    int nref = aux->hdr->n_ref, ret;
    while (1) {
        if ((ret = rd_next(aux->rd, b)) < 0) break;
        if (b->tid >= 0 && b->tid < nref) aux->stats[b->tid].n_reads++;
        if (aux->fail_flags && (b->flag & aux->fail_flags)) continue;
        if (aux->required_flags && !(b->flag & aux->required_flags)) continue;
        if (b->mapq < aux->min_mapQ) continue;
        if (aux->min_len && cigar2qlen(b) < aux->min_len) continue;
        if (b->tid >= 0 && b->tid < nref) { aux->stats[b->tid].n_selected_reads++; aux->stats[b->tid].summed_mapQ += b->mapq; }
        break;
    }
    return ret;
}

static void print_tabular_line(FILE *out, const ohdr_t *h, const cstats_t *stats, int tid, int *header)
{
    if (*header) { fputs("#rname\tstartpos\tendpos\tnumreads\tcovbases\tcoverage\tmeandepth\tmeanbaseq\tmeanmapq\n", out); *header = 0; }
    fputs(h->name[tid], out);
    double region_len = (double)stats[tid].end - stats[tid].beg;
    fprintf(out, "\t%lld\t%lld\t%u\t%llu\t%g\t%g\t%.3g\t%.3g\n", (long long)stats[tid].beg + 1, (long long)stats[tid].end,
            stats[tid].n_selected_reads, stats[tid].n_covered_bases, 100.0 * stats[tid].n_covered_bases / region_len,
            stats[tid].summed_coverage / region_len,
            stats[tid].quality_bases > 0 ? stats[tid].summed_baseQ / (double)stats[tid].quality_bases : 0,
            stats[tid].n_selected_reads > 0 ? stats[tid].summed_mapQ / (double)stats[tid].n_selected_reads : 0);
}

int o_main_coverage(int argc, char *argv[])
{
    int c, i, j, max_depth = 1000000, min_baseQ = 0, min_mapQ = 0, min_len = 0, mindepth = 1, print_header = 1, warn = 0;
    int fail_flags = F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP, required_flags = 0;
    const char *reg = NULL;
    static const struct option lopts[] = {
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 }, { "incl-flags", required_argument, NULL, 1 },
        { "excl-flags", required_argument, NULL, 2 }, { "min-read-len", required_argument, NULL, 'l' }, { "min-MQ", required_argument, NULL, 'q' },
        { "min-mq", required_argument, NULL, 'q' }, { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "no-header", no_argument, NULL, 'H' }, { "region", required_argument, NULL, 'r' }, { "depth", required_argument, NULL, 'd' },
        { "min-depth", required_argument, NULL, 3 }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "l:q:Q:Hr:d:", lopts, NULL)) >= 0) {
        switch (c) {
        case 1: if ((required_flags = str2flag(optarg)) < 0) return 1; break;
        case 2: if ((fail_flags = str2flag(optarg)) < 0) return 1; break;
        case 3: if ((i = atoi(optarg)) > 0) mindepth = i; break;
        case 'l': min_len = atoi(optarg); break;
        case 'q': min_mapQ = atoi(optarg); break;
        case 'Q': min_baseQ = atoi(optarg); break;
        case 'd': max_depth = atoi(optarg); break;
        case 'r': reg = optarg; break;
        case 'H': print_header = 0; break;
        default: return 1;
        }
    }
    if (optind == argc) { fprintf(stderr, "Usage: oracle_samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1; }
    int n = argc - optind;
    caux_t *aux = (caux_t *)calloc((size_t)n, sizeof(caux_t));
    void **data = (void **)calloc((size_t)n, sizeof(void *));
    int reg_tid = -1; hpos_t reg_beg = 0, reg_end = HPOS_MAX;
    for (i = 0; i < n; ++i) {
        aux[i].rd = rd_open(argv[optind + i]);
        if (!aux[i].rd) { fprintf(stderr, "samtools coverage: Could not open \"%s\"\n", argv[optind + i]); return 1; }
        aux[i].hdr = rd_header(aux[i].rd);
        aux[i].min_mapQ = min_mapQ; aux[i].min_len = min_len; aux[i].fail_flags = fail_flags; aux[i].required_flags = required_flags;
        if (reg) {
            int t; hpos_t b, e;
            if (parse_region(aux[i].hdr, reg, &t, &b, &e) < 0) { fprintf(stderr, "samtools coverage: Failed to parse region \"%s\"\n", reg); return 1; }
            rd_set_region(aux[i].rd, t, b, e);
            if (i == 0) { reg_tid = t; reg_beg = b; reg_end = e; }
        }
        data[i] = &aux[i];
    }
    ohdr_t *h = aux[0].hdr;
    int n_targets = h->n_ref;
    cstats_t *stats = (cstats_t *)calloc((size_t)(n_targets > 0 ? n_targets : 1), sizeof(cstats_t));
    if (reg) {
        cstats_t *s = stats + reg_tid;
        s->beg = reg_beg; s->end = reg_end;
        if (s->end == HPOS_MAX || s->end > h->len[reg_tid]) s->end = h->len[reg_tid];
    }
    for (i = 0; i < n; ++i) aux[i].stats = stats;
    omplp_t *mplp = omplp_init(n, read_bam, data);
    if (max_depth > 0) omplp_set_maxcnt(mplp, max_depth); else if (!max_depth) omplp_set_maxcnt(mplp, INT_MAX);
    int *n_plp = (int *)calloc((size_t)n, sizeof(int));
    const opileup1_t **plp = (const opileup1_t **)calloc((size_t)n, sizeof(*plp));
    int ret, tid = -1, old_tid = -1; hpos_t pos;
    while ((ret = omplp_auto(mplp, &tid, &pos, n_plp, plp)) > 0) {
        if (tid != old_tid) {
            if (old_tid >= 0) print_tabular_line(stdout, h, stats, old_tid, &print_header);
            stats[tid].covered = 1;
            if (!reg) stats[tid].end = h->len[tid];
            old_tid = tid;
        }
        if (pos < stats[tid].beg || pos >= stats[tid].end) continue;
        if (tid >= n_targets) continue;
        int count_base = 0;
        unsigned long long summed_baseQ = 0, quality_bases = 0, depth = 0;
        for (i = 0; i < n; ++i) {
            int depth_at_pos = n_plp[i];
            for (j = 0; j < n_plp[i]; ++j) {
                const opileup1_t *p = plp[i] + j;
                if (p->is_del || p->is_refskip) --depth_at_pos;
                else if (p->qpos < p->b->l_qseq) {
                    if (p->b->qual[p->qpos] < min_baseQ) --depth_at_pos;
                    else { summed_baseQ += p->b->qual[p->qpos]; ++quality_bases; }
                } else warn = 1;
            }
            if (depth_at_pos > 0) { count_base = 1; depth += (unsigned long long)depth_at_pos; }
        }
        if (count_base && depth >= (unsigned long long)mindepth) {
            stats[tid].summed_coverage += depth; stats[tid].summed_baseQ += summed_baseQ; stats[tid].quality_bases += quality_bases;
            stats[tid].n_covered_bases++;
        }
    }
    int status = 0;
    if (ret < 0) status = 1;
    else {
        if (tid == -1 && reg && *reg != '*') tid = reg_tid;
        if (tid < n_targets && tid >= 0) print_tabular_line(stdout, h, stats, tid, &print_header);
        if (!reg)
            for (i = 0; i < n_targets; ++i)
                if (!stats[i].covered) { stats[i].end = h->len[i]; print_tabular_line(stdout, h, stats, i, &print_header); }
        if (warn) fprintf(stderr, "samtools coverage: Warning:  Missing quality values in alignments.  Mean base quality calculated only on available values.\n");
    }
    omplp_destroy(mplp);
    for (i = 0; i < n; ++i) rd_close(aux[i].rd);
    free(n_plp); free(plp); free(stats); free(aux); free(data);
    return status;
}
