/*
 * oracle/o_coverage.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Restatement of `samtools coverage` (coverage.c:176-221 read_bam + print_tabular_line, :572-700 the multi-pileup loop,
 * :150-173 + :223-304 the histogram / depth plot of -m -A -D -w).
 * Pinned by test/coverage/{1..5}.expected (test/test.pl:4143-4161) for the tabular mode.  For the histogram modes the reference
 * holds no expected file; what it holds is the worked example of its manual (doc/samtools-coverage.1:147-178), which pins the title
 * line, the x axis (label positions, centring, readable_bps rounding) and the bin-width text for two regions
 * (tests/test_coverage.py); the bars themselves are PARITY UNPINNED (restated from coverage.c alone) -- their inputs, the per-bin
 * counts, are sums of the very per-column values the pinned tabular mode adds up.
 */
#include "o_plp.h"
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <sys/ioctl.h>

typedef struct {
    unsigned long long n_covered_bases, summed_coverage, summed_baseQ, summed_mapQ, quality_bases;
    unsigned int n_reads, n_selected_reads;
    int covered;
    hpos_t beg, end;
    int64_t bin_width;
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

/* coverage.c:163-173 */
static char *readable_bps(double base_pairs, char *buf)
{
    const char *units[] = { "", "K", "M", "G", "T" };
    int i = 0;
    while (base_pairs >= 1000 && i < 4) { base_pairs /= 1000; i++; }
    sprintf(buf, "%.*f%s", i, base_pairs, units[i]);
    return buf;
}

/* coverage.c:150-161 */
static char *center_text(char *text, char *buf, int width)
{
    int len = (int)strlen(text);
    int padding = (width - len) / 2, padding_ex = (width - len) % 2;
    if (padding >= 1) sprintf(buf, " %*s%*s", len + padding, text, padding - 1 + padding_ex, " ");
    else sprintf(buf, "%s", text);
    return buf;
}

/* coverage.c:223-304 */
static void print_hist(FILE *out, const ohdr_t *h, const cstats_t *stats, int tid, const uint32_t *hist, int hist_size, int full_utf, int plot_coverage)
{
    static const char *const B8[8] = { "\xE2\x96\x81", "\xE2\x96\x82", "\xE2\x96\x83", "\xE2\x96\x84", "\xE2\x96\x85", "\xE2\x96\x86", "\xE2\x96\x87", "\xE2\x96\x88" };
    static const char *const B2[2] = { ".", ":" };
    const char *const *BLOCK = full_utf ? B8 : B2;
    const char *VLINE = full_utf ? "\xE2\x94\x82" : "|";
    int i, col, n_rows = 10, nblk = full_utf ? 8 : 2;
    double region_len = (double)(stats[tid].end - stats[tid].beg);
    double *hd = (double *)calloc((size_t)(hist_size > 0 ? hist_size : 1), sizeof(double));
    double max_val = 0.0;
    for (i = 0; i < hist_size; ++i) {
        hd[i] = (plot_coverage ? 1 : 100) * hist[i] / (double)stats[tid].bin_width;
        if (hd[i] > max_val) max_val = hd[i];
    }
    char buf[64], buf2[64];
    fprintf(out, "%s (%sbp)\n", h->name[tid], readable_bps((double)h->len[tid], buf));
    double row = max_val / (double)n_rows;
    for (i = n_rows - 1; i >= 0; --i) {
        double cur = row * i;
        if (plot_coverage) fprintf(out, ">%8.1f ", i * row);
        else fprintf(out, ">%7.2f%% ", cur);
        fprintf(out, "%s", VLINE);
        for (col = 0; col < hist_size; ++col) {
            int d = round(nblk * (hd[col] - cur) / row) - 1;
            if (d < 0) fputc(' ', out);
            else { if (d >= nblk) d = nblk - 1; fprintf(out, "%s", BLOCK[d]); }
        }
        fprintf(out, "%s", VLINE);
        fputc(' ', out);
        switch (i) {
        case 9: fprintf(out, "Number of reads: %u", stats[tid].n_selected_reads); break;
        case 8: if (stats[tid].n_reads - stats[tid].n_selected_reads > 0) fprintf(out, "    (%i filtered)", stats[tid].n_reads - stats[tid].n_selected_reads); break;
        case 7: fprintf(out, "Covered bases:   %sbp", readable_bps((double)stats[tid].n_covered_bases, buf)); break;
        case 6: fprintf(out, "Percent covered: %.4g%%", 100.0 * stats[tid].n_covered_bases / region_len); break;
        case 5: fprintf(out, "Mean coverage:   %.3gx", stats[tid].summed_coverage / region_len); break;
        case 4: fprintf(out, "Mean baseQ:      %.3g", stats[tid].quality_bases > 0 ? stats[tid].summed_baseQ / (double)stats[tid].quality_bases : 0); break;
        case 3: fprintf(out, "Mean mapQ:       %.3g", stats[tid].summed_mapQ / (double)stats[tid].n_selected_reads); break;
        case 1: fprintf(out, "Histo bin width: %sbp", readable_bps((double)stats[tid].bin_width, buf)); break;
        case 0: if (plot_coverage) fprintf(out, "Histo max cov:   %.5g", max_val); else fprintf(out, "Histo max bin:   %.5g%%", max_val); break;
        }
        fputc('\n', out);
    }
    fprintf(out, "     %s", center_text(readable_bps((double)(stats[tid].beg + 1), buf), buf2, 10));
    int rest;
    for (rest = 10; rest < 10 * (hist_size / 10); rest += 10)
        fprintf(out, "%s", center_text(readable_bps((double)(stats[tid].beg + stats[tid].bin_width * rest), buf), buf2, 10));
    fprintf(out, "%*s%s", hist_size % 10, " ", center_text(readable_bps((double)stats[tid].end, buf), buf2, 10));
    fprintf(out, "\n");
    free(hd);
}

int o_main_coverage(int argc, char *argv[])
{
    int c, i, j, max_depth = 1000000, min_baseQ = 0, min_mapQ = 0, min_len = 0, mindepth = 1, print_header = 1, warn = 0;
    int fail_flags = F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP, required_flags = 0;
    const char *reg = NULL;
    int n_bins_opt = 50, full_width = 1, tabular = 1, histogram = 0, plot_cov = 0, full_utf = 1;
    static const struct option lopts[] = {
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 }, { "incl-flags", required_argument, NULL, 1 },
        { "excl-flags", required_argument, NULL, 2 }, { "min-read-len", required_argument, NULL, 'l' }, { "min-MQ", required_argument, NULL, 'q' },
        { "min-mq", required_argument, NULL, 'q' }, { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "no-header", no_argument, NULL, 'H' }, { "region", required_argument, NULL, 'r' }, { "depth", required_argument, NULL, 'd' },
        { "min-depth", required_argument, NULL, 3 }, { "histogram", no_argument, NULL, 'm' }, { "ascii", no_argument, NULL, 'A' },
        { "plot-depth", no_argument, NULL, 'D' }, { "n-bins", required_argument, NULL, 'w' }, { NULL, 0, NULL, 0 } };
    optind = 1;
    while ((c = getopt_long(argc, argv, "l:q:Q:Hr:d:mADw:o:", lopts, NULL)) >= 0) {
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
        case 'o': full_width = 0; if (strcmp(optarg, "-") != 0 && !freopen(optarg, "w", stdout)) { fprintf(stderr, "samtools coverage: Cannot open \"%s\" for writing.\n", optarg); return 1; } break;
        case 'w': n_bins_opt = atoi(optarg); full_width = 0; histogram = 1; tabular = 0; break;
        case 'm': histogram = 1; tabular = 0; break;
        case 'A': full_utf = 0; histogram = 1; tabular = 0; break;
        case 'D': histogram = 1; tabular = 0; plot_cov = 1; break;
        default: return 1;
        }
    }
    if (optind == argc) { fprintf(stderr, "Usage: oracle_samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1; }
    if (n_bins_opt <= 0 || full_width) {       /* coverage.c:427-451 */
        const char *env_columns = getenv("COLUMNS");
        int columns = 0;
        if (env_columns == NULL) { struct winsize w; if (ioctl(2, TIOCGWINSZ, &w) == 0) columns = w.ws_col; }
        else columns = atoi(env_columns);
        n_bins_opt = columns > 60 ? columns - 40 : 40;
    }
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
    int64_t n_bins = n_bins_opt;
    if (reg) {
        cstats_t *s = stats + reg_tid;
        s->beg = reg_beg; s->end = reg_end;
        if (s->end == HPOS_MAX || s->end > h->len[reg_tid]) s->end = h->len[reg_tid];
        if (n_bins_opt > s->end - s->beg) n_bins = s->end - s->beg;
        s->bin_width = (s->end - s->beg) / (n_bins > 0 ? n_bins : 1);
    }
    for (i = 0; i < n; ++i) aux[i].stats = stats;
    int64_t current_bin = 0;
    uint32_t *hist = (uint32_t *)calloc((size_t)n_bins_opt, sizeof(uint32_t));
    omplp_t *mplp = omplp_init(n, read_bam, data);
    if (max_depth > 0) omplp_set_maxcnt(mplp, max_depth); else if (!max_depth) omplp_set_maxcnt(mplp, INT_MAX);
    int *n_plp = (int *)calloc((size_t)n, sizeof(int));
    const opileup1_t **plp = (const opileup1_t **)calloc((size_t)n, sizeof(*plp));
    int ret, tid = -1, old_tid = -1; hpos_t pos;
    while ((ret = omplp_auto(mplp, &tid, &pos, n_plp, plp)) > 0) {
        if (tid != old_tid) {
            if (old_tid >= 0) {
                if (histogram) { print_hist(stdout, h, stats, old_tid, hist, (int)n_bins, full_utf, plot_cov); fputc('\n', stdout); }
                else if (tabular) print_tabular_line(stdout, h, stats, old_tid, &print_header);
                if (histogram) memset(hist, 0, (size_t)n_bins * sizeof(uint32_t));
            }
            stats[tid].covered = 1;
            if (!reg) stats[tid].end = h->len[tid];
            if (histogram) {
                n_bins = n_bins_opt > stats[tid].end - stats[tid].beg ? stats[tid].end - stats[tid].beg : n_bins_opt;
                stats[tid].bin_width = (stats[tid].end - stats[tid].beg) / n_bins;
            }
            old_tid = tid;
        }
        if (pos < stats[tid].beg || pos >= stats[tid].end) continue;
        if (tid >= n_targets) continue;
        if (histogram) current_bin = (pos - stats[tid].beg) / stats[tid].bin_width;
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
            if (current_bin < n_bins && plot_cov) hist[current_bin] += depth_at_pos;
        }
        if (count_base && depth >= (unsigned long long)mindepth) {
            stats[tid].summed_coverage += depth; stats[tid].summed_baseQ += summed_baseQ; stats[tid].quality_bases += quality_bases;
            stats[tid].n_covered_bases++;
            if (histogram && current_bin < n_bins && !plot_cov) ++hist[current_bin];
        }
    }
    int status = 0;
    if (ret < 0) status = 1;
    else {
        if (tid == -1 && reg && *reg != '*') tid = reg_tid;
        if (tid < n_targets && tid >= 0) {
            if (histogram) print_hist(stdout, h, stats, tid, hist, (int)n_bins, full_utf, plot_cov);
            else if (tabular) print_tabular_line(stdout, h, stats, tid, &print_header);
        }
        if (!reg && tabular)
            for (i = 0; i < n_targets; ++i)
                if (!stats[i].covered) { stats[i].end = h->len[i]; print_tabular_line(stdout, h, stats, i, &print_header); }
        if (warn) fprintf(stderr, "samtools coverage: Warning:  Missing quality values in alignments.  Mean base quality calculated only on available values.\n");
    }
    omplp_destroy(mplp);
    for (i = 0; i < n; ++i) rd_close(aux[i].rd);
    free(n_plp); free(plp); free(hist); free(stats); free(aux); free(data);
    return status;
}
