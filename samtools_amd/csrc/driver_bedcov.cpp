// driver_bedcov.cpp -- `samtools-amd bedcov`: the reference's bedcov column loop (bedcov.c:297-360) running on
// the MI355X pileup iterator.  Written the way a samtools maintainer would switch bedcov.c over: the HTSlib iterator
// names below ARE the engine's (STA_PLP_DROPIN macros of include/samtools_amd_plp.h); only file access differs
// (host_io.h readers stand in for sam_open / sam_itr_queryi, which are HTSlib I/O and out of scope).
//   bedcov [-Q mapq] [-g flags] [-G flags] [-j] [-d depth] [--max-depth n] [-c] [-H] in.bed in1.bam [...]
#define STA_PLP_DROPIN
#include "../../include/samtools_amd.h"
#include "../../include/samtools_amd_plp.h"
#include "host_io.h"
#include "bam1_from_rec.h"
#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <vector>

using namespace sta;

namespace {

struct aux_t {
    std::unique_ptr<AlnReader> fp;
    Rec rec;
    int min_mapQ = 0;
    uint32_t flags = 0;
    int64_t rcnt = 0;
};

// bedcov.c:54-70
int read_bam(void *data, bam1_t *b)
{
    aux_t *aux = (aux_t *)data;
    for (;;) {
        int ret = aux->fp->next(aux->rec);
        if (ret == 0) return -1;
        if (ret < 0) return -2;
        if (aux->rec.flag & aux->flags) continue;
        if ((int)aux->rec.mapq < aux->min_mapQ) continue;
        rec_to_bam1(aux->rec, b);
        return 0;
    }
}

int incr_rcnt(void *data, const bam1_t *, bam_pileup_cd *) { ((aux_t *)data)->rcnt++; return 0; }

// bedcov.c:81-108
void output_header(FILE *fp, const char *hdr, int fields, int n, char **fn, int depth, int rcount)
{
    static const char *bedcols[] = { "chrom", "chromStart", "chromEnd", "name", "score", "strand", "thickStart", "thickEnd",
                                     "itemRgb", "blockCount", "blockSizes", "blockStarts" };
    if (hdr) fprintf(fp, "%s", hdr);
    else for (int i = 0; i < fields; ++i) fprintf(fp, "%s%s", (i ? "\t" : "#"), (i < 12 ? bedcols[i] : "."));
    for (int i = 0; i < n; ++i) fprintf(fp, "\t%s_cov", fn[i]);
    if (depth >= 0) for (int i = 0; i < n; ++i) fprintf(fp, "\t%s_depth", fn[i]);
    if (rcount) for (int i = 0; i < n; ++i) fprintf(fp, "\t%s_count", fn[i]);
    fprintf(fp, "\n");
}

}  // namespace

extern "C" int sta_main_bedcov_iter(int argc, char **argv)
{
    int c, status = 0, min_mapQ = 0, skip_DN = 0, do_rcount = 0, tflags, min_depth = -1, max_depth = INT_MAX, print_header = 0, hdr = 0;
    uint32_t flags = 4 | 256 | 512 | 1024;
    static const struct option lopts[] = { { "min-MQ", required_argument, NULL, 'Q' }, { "min-mq", required_argument, NULL, 'Q' },
                                           { "max-depth", required_argument, NULL, 'd' + 1000 }, { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt_long(argc, argv, "Q:g:G:jd:Hc", lopts, NULL)) >= 0) {
        switch (c) {
        case 'Q': min_mapQ = atoi(optarg); break;
        case 'c': do_rcount = 1; break;
        case 'H': print_header = 1; break;
        case 'g':
            tflags = str2flag(optarg);
            if (tflags < 0 || tflags > ((2048 << 1) - 1)) { fprintf(stderr, "samtools bedcov: Flag value \"%s\" is not supported\n", optarg); return 1; }
            flags &= ~(uint32_t)tflags; break;
        case 'G':
            tflags = str2flag(optarg);
            if (tflags < 0 || tflags > ((2048 << 1) - 1)) { fprintf(stderr, "samtools bedcov: Flag value \"%s\" is not supported\n", optarg); return 1; }
            flags |= (uint32_t)tflags; break;
        case 'j': skip_DN = 1; break;
        case 'd': min_depth = atoi(optarg); break;
        case 'd' + 1000: max_depth = atoi(optarg); break;
        default: fprintf(stderr, "Usage: samtools bedcov [options] <in.bed> <in1.bam> [...]\n"); return 1;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, "Usage: samtools bedcov [options] <in.bed> <in1.bam> [...]\n"); return 1; }
    const int n = argc - optind - 1;
    char **fn = argv + optind + 1;
    if (!print_header) hdr = 1;
    if (sta_device_count() < 1) { fprintf(stderr, "samtools bedcov: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::vector<aux_t> aux((size_t)n);
    std::vector<void *> data((size_t)n);
    std::string err;
    auto r0 = AlnReader::open(fn[0], &err);
    if (!r0) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[0]); return 2; }
    const Header h0 = r0->header();
    std::vector<int64_t> cnt((size_t)n), pcov((size_t)n);
    std::vector<int> n_plp((size_t)n);
    std::vector<const bam_pileup1_t *> plp((size_t)n);
    FILE *fp = fopen(argv[optind], "r");
    if (!fp) { fprintf(stderr, "samtools bedcov: can't open BED file '%s'\n", argv[optind]); return 2; }
    char *line = nullptr; size_t cap = 0; ssize_t len;
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
        bool bad = *p == 0;
        int tid = -1; long long beg = 0, end = 0;
        if (!bad) {
            char ch = *p; *p = 0; tid = h0.tid(q); *p = ch;
            if (tid < 0 || sscanf(p + 1, "%lld %lld", &beg, &end) < 2 || end < beg) bad = true;
        }
        if (bad) { fprintf(stderr, "Errors in BED line '%s'\n", line); status = 2; continue; }
        for (int i = 0; i < n; ++i) {
            aux[(size_t)i].fp = AlnReader::open(fn[i], &err);
            if (!aux[(size_t)i].fp) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[i]); return 2; }
            aux[(size_t)i].fp->set_region(tid, beg, end);          // sam_itr_queryi(idx, tid, beg, end)
            aux[(size_t)i].min_mapQ = min_mapQ; aux[(size_t)i].flags = flags; aux[(size_t)i].rcnt = 0;
            data[(size_t)i] = &aux[(size_t)i];
        }
        // ---- from here on: the column loop of bedcov.c:303-352 restated on the engine's iterator (HTSlib names via STA_PLP_DROPIN) ----
        bam_mplp_t mplp = bam_mplp_init(n, read_bam, data.data());
        bam_mplp_set_maxcnt(mplp, min_depth > max_depth ? min_depth : max_depth);
        std::fill(cnt.begin(), cnt.end(), 0); std::fill(pcov.begin(), pcov.end(), 0);
        if (do_rcount) bam_mplp_constructor(mplp, incr_rcnt);
        int ptid = 0, ret; hts_pos_t pos = 0;
        while ((ret = bam_mplp64_auto(mplp, &ptid, &pos, n_plp.data(), plp.data())) > 0)
            if (pos >= beg && pos < end) {
                for (int i = 0; i < n; ++i) {
                    int m = 0;
                    if (skip_DN || min_depth >= 0)
                        for (int j = 0; j < n_plp[(size_t)i]; ++j) { const bam_pileup1_t *pi = plp[(size_t)i] + j; if (pi->is_del || pi->is_refskip) ++m; }
                    int pd = n_plp[(size_t)i] - m;
                    cnt[(size_t)i] += pd;
                    if (min_depth >= 0 && pd >= min_depth) pcov[(size_t)i]++;
                }
            }
        if (ret < 0) { fprintf(stderr, "samtools bedcov: error reading from input file\n"); status = 2; bam_mplp_destroy(mplp); break; }
        fputs(line, stdout);
        for (int i = 0; i < n; ++i) printf("\t%lld", (long long)cnt[(size_t)i]);
        if (min_depth >= 0) for (int i = 0; i < n; ++i) printf("\t%lld", (long long)pcov[(size_t)i]);
        if (do_rcount) for (int i = 0; i < n; ++i) printf("\t%lld", (long long)aux[(size_t)i].rcnt);
        putchar('\n');
        bam_mplp_destroy(mplp);
    }
    free(line); fclose(fp);
    return status;
}
