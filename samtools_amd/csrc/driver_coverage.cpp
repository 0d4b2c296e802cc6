// driver_coverage.cpp -- `samtools-amd coverage` (tabular mode): the reference's multi-pileup loop
// (coverage.c:572-700, read_bam :176-199, print_tabular_line :201-221) on the MI355X pileup iterator.
// As in driver_bedcov.cpp the HTSlib iterator names are the engine's (STA_PLP_DROPIN); host_io.h readers stand in
// for sam_open / sam_itr_querys.  Histogram and plot modes (-m -A -D -w) are terminal art and not provided.
#define STA_PLP_DROPIN
#include "../../include/samtools_amd.h"
#include "../../include/samtools_amd_plp.h"
#include "host_io.h"
#include "bam1_from_rec.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <vector>

using namespace sta;

namespace {

struct stats_aux_t {
    unsigned long long n_covered_bases = 0, summed_coverage = 0, summed_baseQ = 0, summed_mapQ = 0, quality_bases = 0;
    unsigned int n_reads = 0, n_selected_reads = 0;
    bool covered = false;
    int64_t beg = 0, end = 0;
};

struct bam_aux_t {
    std::unique_ptr<AlnReader> fp;
    Rec rec;
    int nref = 0, min_mapQ = 0, min_len = 0, fail_flags = 0, required_flags = 0;
    std::vector<stats_aux_t> *stats = nullptr;
};

int cigar2qlen(const Rec &r)
{
    int l = 0;
    for (uint32_t c : r.cigar) { int op = c & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) l += (int)(c >> 4); }
    return l;
}

// coverage.c:176-199
int read_bam(void *data, bam1_t *b)
{
    bam_aux_t *aux = (bam_aux_t *)data;
    for (;;) {
        int ret = aux->fp->next(aux->rec);
        if (ret == 0) return -1;
        if (ret < 0) return -2;
        const Rec &r = aux->rec;
        if (r.tid >= 0 && r.tid < aux->nref) (*aux->stats)[(size_t)r.tid].n_reads++;
        if (aux->fail_flags && (r.flag & aux->fail_flags)) continue;
        if (aux->required_flags && !(r.flag & aux->required_flags)) continue;
        if (r.mapq < aux->min_mapQ) continue;
        if (aux->min_len && cigar2qlen(r) < aux->min_len) continue;
        if (r.tid >= 0 && r.tid < aux->nref) { (*aux->stats)[(size_t)r.tid].n_selected_reads++; (*aux->stats)[(size_t)r.tid].summed_mapQ += r.mapq; }
        rec_to_bam1(r, b);
        return 0;
    }
}

// coverage.c:201-221
void print_tabular_line(FILE *out, const Header &h, const std::vector<stats_aux_t> &stats, int tid, bool *header)
{
    if (*header) { fputs("#rname\tstartpos\tendpos\tnumreads\tcovbases\tcoverage\tmeandepth\tmeanbaseq\tmeanmapq\n", out); *header = false; }
    const stats_aux_t &s = stats[(size_t)tid];
    fputs(h.names[(size_t)tid].c_str(), out);
    double region_len = (double)s.end - s.beg;
    fprintf(out, "\t%lld\t%lld\t%u\t%llu\t%g\t%g\t%.3g\t%.3g\n", (long long)s.beg + 1, (long long)s.end, s.n_selected_reads, s.n_covered_bases,
            100.0 * s.n_covered_bases / region_len, s.summed_coverage / region_len,
            s.quality_bases > 0 ? s.summed_baseQ / (double)s.quality_bases : 0,
            s.n_selected_reads > 0 ? s.summed_mapQ / (double)s.n_selected_reads : 0);
}

}  // namespace

extern "C" int sta_main_coverage_iter(int argc, char **argv)
{
    int c, i, max_depth = 1000000, opt_min_baseQ = 0, opt_min_mapQ = 0, opt_min_len = 0, mindepth = 1, print_value_warning = 0;
    int fail_flags = 4 | 256 | 512 | 1024, required_flags = 0;
    bool opt_print_header = true;
    const char *opt_reg = nullptr, *opt_output_file = nullptr;
    static const struct option lopts[] = {
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 }, { "incl-flags", required_argument, NULL, 1 },
        { "excl-flags", required_argument, NULL, 2 }, { "min-read-len", required_argument, NULL, 'l' }, { "min-MQ", required_argument, NULL, 'q' },
        { "min-mq", required_argument, NULL, 'q' }, { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "histogram", no_argument, NULL, 'm' }, { "ascii", no_argument, NULL, 'A' }, { "plot-depth", no_argument, NULL, 'D' },
        { "output", required_argument, NULL, 'o' }, { "no-header", no_argument, NULL, 'H' }, { "n-bins", required_argument, NULL, 'w' },
        { "region", required_argument, NULL, 'r' }, { "depth", required_argument, NULL, 'd' }, { "min-depth", required_argument, NULL, 3 },
        { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt_long(argc, argv, "Ao:l:q:Q:hHw:r:b:md:D", lopts, NULL)) >= 0) {
        switch (c) {
        case 1: if ((required_flags = str2flag(optarg)) < 0) { fprintf(stderr, "Could not parse --rf %s\n", optarg); return 1; } break;
        case 2: if ((fail_flags = str2flag(optarg)) < 0) { fprintf(stderr, "Could not parse --ff %s\n", optarg); return 1; } break;
        case 3: if ((i = atoi(optarg)) > 0) mindepth = i; break;
        case 'o': opt_output_file = optarg; break;
        case 'l': opt_min_len = atoi(optarg); break;
        case 'q': opt_min_mapQ = atoi(optarg); break;
        case 'Q': opt_min_baseQ = atoi(optarg); break;
        case 'd': max_depth = atoi(optarg); break;
        case 'r': opt_reg = optarg; break;
        case 'H': opt_print_header = false; break;
        case 'm': case 'A': case 'D': case 'w': case 'b':
            fprintf(stderr, "samtools coverage: option -%c (histogram / plot / file list) is not provided by the MI355X engine build\n", c); return 1;
        default: fprintf(stderr, "Usage: samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1;
        }
    }
    if (optind == argc) { fprintf(stderr, "Usage: samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1; }
    if (sta_device_count() < 1) { fprintf(stderr, "samtools coverage: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 1; }
    FILE *file_out = stdout;
    if (opt_output_file && strcmp(opt_output_file, "-") != 0) {
        file_out = fopen(opt_output_file, "w");
        if (!file_out) { fprintf(stderr, "samtools coverage: Cannot open \"%s\" for writing.\n", opt_output_file); return 1; }
    }
    const int n_bam_files = argc - optind;
    std::vector<bam_aux_t> data((size_t)n_bam_files);
    std::vector<void *> dptr((size_t)n_bam_files);
    std::vector<stats_aux_t> stats;
    int reg_tid = -1; int64_t reg_beg = 0, reg_end = INT64_MAX;
    for (i = 0; i < n_bam_files; ++i) {
        std::string err;
        data[(size_t)i].fp = AlnReader::open(argv[optind + i], &err);
        if (!data[(size_t)i].fp) { fprintf(stderr, "samtools coverage: Could not open \"%s\"\n", argv[optind + i]); return 1; }
        data[(size_t)i].min_mapQ = opt_min_mapQ; data[(size_t)i].min_len = opt_min_len;
        data[(size_t)i].fail_flags = fail_flags; data[(size_t)i].required_flags = required_flags;
        data[(size_t)i].nref = data[(size_t)i].fp->header().nref();
        if (opt_reg) {
            int t; int64_t b, e;
            if (!parse_region(data[(size_t)i].fp->header(), opt_reg, &t, &b, &e)) {
                fprintf(stderr, "samtools coverage: Failed to parse region \"%s\". Check the region format or region name presence in the file \"%s\"\n", opt_reg, argv[optind + i]);
                return 1;
            }
            data[(size_t)i].fp->set_region(t, b, e);
            if (i == 0) { reg_tid = t; reg_beg = b; reg_end = e; }
        }
        dptr[(size_t)i] = &data[(size_t)i];
    }
    const Header &h = data[0].fp->header();
    const int n_targets = h.nref();
    stats.assign((size_t)(n_targets > 0 ? n_targets : 1), stats_aux_t());
    if (opt_reg) {
        stats_aux_t &s = stats[(size_t)reg_tid];
        s.beg = reg_beg; s.end = reg_end;
        if (s.end == INT64_MAX || s.end > h.lens[(size_t)reg_tid]) s.end = h.lens[(size_t)reg_tid];
    }
    for (i = 0; i < n_bam_files; ++i) data[(size_t)i].stats = &stats;

    // the core multi-pileup loop (coverage.c:572-672)
    bam_mplp_t mplp = bam_mplp_init(n_bam_files, read_bam, dptr.data());
    if (max_depth > 0) bam_mplp_set_maxcnt(mplp, max_depth);
    else if (!max_depth) bam_mplp_set_maxcnt(mplp, INT_MAX);
    std::vector<int> n_plp((size_t)n_bam_files);
    std::vector<const bam_pileup1_t *> plp((size_t)n_bam_files);
    int ret, tid = -1, old_tid = -1; hts_pos_t pos = 0;
    while ((ret = bam_mplp64_auto(mplp, &tid, &pos, n_plp.data(), plp.data())) > 0) {
        if (tid != old_tid) {
            if (old_tid >= 0) print_tabular_line(file_out, h, stats, old_tid, &opt_print_header);
            stats[(size_t)tid].covered = true;
            if (!opt_reg) stats[(size_t)tid].end = h.lens[(size_t)tid];
            old_tid = tid;
        }
        if (pos < stats[(size_t)tid].beg || pos >= stats[(size_t)tid].end) continue;
        if (tid >= n_targets) continue;
        bool count_base = false;
        unsigned long long summed_baseQ = 0, quality_bases = 0, depth = 0;
        for (i = 0; i < n_bam_files; ++i) {
            int depth_at_pos = n_plp[(size_t)i];
            for (int j = 0; j < n_plp[(size_t)i]; ++j) {
                const bam_pileup1_t *p = plp[(size_t)i] + j;
                if (p->is_del || p->is_refskip) --depth_at_pos;
                else if (p->qpos < p->b->core.l_qseq) {
                    if (bam_get_qual(p->b)[p->qpos] < opt_min_baseQ) --depth_at_pos;
                    else { summed_baseQ += bam_get_qual(p->b)[p->qpos]; ++quality_bases; }
                } else print_value_warning = 1;
            }
            if (depth_at_pos > 0) { count_base = true; depth += (unsigned long long)depth_at_pos; }
        }
        if (count_base && depth >= (unsigned long long)mindepth) {
            stats_aux_t &s = stats[(size_t)tid];
            s.summed_coverage += depth; s.summed_baseQ += summed_baseQ; s.quality_bases += quality_bases;
            s.n_covered_bases++;
        }
    }
    int status = 0;
    if (ret < 0) status = 1;
    else {
        if (tid == -1 && opt_reg && *opt_reg != '*') tid = reg_tid;
        if (tid < n_targets && tid >= 0) print_tabular_line(file_out, h, stats, tid, &opt_print_header);
        if (!opt_reg)
            for (i = 0; i < n_targets; ++i)
                if (!stats[(size_t)i].covered) { stats[(size_t)i].end = h.lens[(size_t)i]; print_tabular_line(file_out, h, stats, i, &opt_print_header); }
        if (print_value_warning)
            fprintf(stderr, "samtools coverage: Warning:  Missing quality values in alignments.  Mean base quality calculated only on available values.\n");
    }
    bam_mplp_destroy(mplp);
    if (file_out != stdout) fclose(file_out);
    return status;
}
