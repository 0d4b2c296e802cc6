// driver_cov_native.cpp -- `coverage` (tabular, histogram -m and depth plot -D) and `bedcov` with the column work done on the
// device (sta_cov_plan / k_cov_cols) instead of a host loop over the pileup iterator.  Same options, text and exit status
// as driver_coverage.cpp / driver_bedcov.cpp (which keep the reference's loops on the bam_mplp_* surface and are
// selected with STA_COV_ITERATOR=1); reference: coverage.c:176-221,:572-700 and bedcov.c:54-70,:297-360.
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_pump.h"
#include "host_stage.h"
#include <algorithm>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <sys/ioctl.h>
#include <string>
#include <vector>

using namespace sta;

extern "C" int sta_main_coverage_iter(int argc, char **argv);
extern "C" int sta_main_bedcov_iter(int argc, char **argv);

namespace {

struct CovAccum {
    sta_cov_totals tot;
    std::vector<uint64_t> per_file, kept;       // per_file: [nf][2]; kept: reads that entered the pileup, per file
    uint64_t n_kept = 0;
    explicit CovAccum(size_t nf) : per_file(nf * 2, 0), kept(nf, 0) { memset(&tot, 0, sizeof tot); }
};

// every window of contig `tid` restricted to [lo, hi): stage, reduce on the device, accumulate
int cov_run_tid(sta_engine *eng, Pump &pump, std::vector<std::unique_ptr<AlnReader>> &readers, int tid, int64_t lo, int64_t hi,
                const sta_cov_params &cp, int64_t window_cols, bool want_kept, CovAccum &acc, const char *cmd)
{
    const Header &h = readers[0]->header();
    const size_t nf = readers.size();
    std::vector<StagedFile> staged(nf);
    std::vector<sta_reads> views(nf);
    std::vector<std::vector<const Rec *>> reads;
    std::vector<uint32_t> info;
    int64_t cursor = std::max(lo, pump.next_pos(tid));
    bool first = true;
    for (;;) {
        bool more = pump.next_pos(tid) != INT64_MAX;
        if (!more && !pump.has_carry()) break;
        if (!pump.has_carry()) cursor = std::max(cursor, pump.next_pos(tid));
        int64_t ce_target = std::min(cursor + window_cols, hi);
        if (ce_target <= cursor) { pump.fill(tid, cursor, INT64_MAX, reads); pump.drop_tid_carry(); break; }
        int64_t ce = pump.fill(tid, cursor, ce_target, reads);
        if (pump.error()) { fprintf(stderr, "samtools %s: error reading from input file\n", cmd); return -1; }     // (bedcov.c:333; found silent by scripts/hunt7.py)
        if (pump.next_pos(tid) == INT64_MAX) {
            int64_t me = pump.carry_max_end();
            if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
        }
        if (ce > cursor) {
            for (size_t f = 0; f < nf; ++f) {
                staged[f].clear();
                for (const Rec *r : reads[f]) staged[f].add(*r, cursor, nullptr);
                staged[f].finish();
                views[f] = staged[f].view();
            }
            sta_window w; memset(&w, 0, sizeof w);
            w.tid = tid; w.origin = cursor; w.col_beg = 0; w.col_end = (int32_t)(ce - cursor);
            w.tname = h.names[(size_t)tid].c_str(); w.tlen = h.lens[(size_t)tid];
            w.n_files = (int32_t)nf; w.files = views.data(); w.mem = STA_MEM_HOST;
            w.has_reg = 1; w.reg_beg = lo; w.reg_end = hi;
            sta_cov_totals t; std::vector<uint64_t> pf(nf * 2); sta_plan_info pi;
            if (sta_stage_window(eng, &w) != STA_OK || sta_cov_plan(eng, &cp, &t, pf.data(), &pi) != STA_OK) {
                fprintf(stderr, "samtools %s: %s\n", cmd, sta_last_error(eng));
                return -1;
            }
            acc.tot.n_covered_bases += t.n_covered_bases; acc.tot.summed_coverage += t.summed_coverage;
            acc.tot.summed_baseQ += t.summed_baseQ; acc.tot.quality_bases += t.quality_bases; acc.tot.missing_qual += t.missing_qual;
            for (size_t i = 0; i < nf * 2; ++i) acc.per_file[i] += pf[i];
            acc.n_kept += pi.n_kept_reads;
            if (want_kept)
                for (size_t f = 0; f < nf; ++f) {
                    info.resize((size_t)staged[f].n());
                    if (info.empty()) continue;
                    if (sta_fetch_read_state(eng, (int32_t)f, info.data(), nullptr) != STA_OK) { fprintf(stderr, "samtools %s: %s\n", cmd, sta_last_error(eng)); return -1; }
                    // a carried read was counted by the window it arrived in: new reads start at or after the window start
                    for (size_t i = 0; i < info.size(); ++i)
                        if ((info[i] & 2u) && (first || staged[f].pos[i] >= 0)) acc.kept[f]++;
                }
            if (pi.n_maxcnt_dropped)
                for (size_t f = 0; f < nf; ++f) {
                    info.resize((size_t)staged[f].n());
                    if (info.empty() || sta_fetch_read_state(eng, (int32_t)f, info.data(), nullptr) != STA_OK) continue;
                    std::vector<char> dr(info.size());
                    for (size_t i = 0; i < info.size(); ++i) dr[i] = (info[i] & 1u) && !(info[i] & 2u) && reads[f][i]->rlen > 0;
                    pump.drop(f, dr);          // reads the depth cap removed never come back (bam_plp_push did not store them)
                }
            first = false;
        }
        pump.retire(ce);
        cursor = std::max(cursor, ce);
    }
    pump.drop_tid_carry();
    return 0;
}

// ---------------------------------------------------------------------------------------------- coverage
struct CStats {
    unsigned long long n_covered_bases = 0, summed_coverage = 0, summed_baseQ = 0, summed_mapQ = 0, quality_bases = 0;
    unsigned int n_reads = 0, n_selected_reads = 0;
    bool covered = false;
    int64_t beg = 0, end = 0, bin_width = 0;
};

int cigar2qlen(const Rec &r)
{
    int l = 0;
    for (uint32_t c : r.cigar) { int op = c & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) l += (int)(c >> 4); }
    return l;
}

void print_tabular_line(FILE *out, const Header &h, const std::vector<CStats> &stats, int tid, bool *header)
{
    if (*header) { fputs("#rname\tstartpos\tendpos\tnumreads\tcovbases\tcoverage\tmeandepth\tmeanbaseq\tmeanmapq\n", out); *header = false; }
    const CStats &s = stats[(size_t)tid];
    fputs(h.names[(size_t)tid].c_str(), out);
    double region_len = (double)s.end - s.beg;
    fprintf(out, "\t%lld\t%lld\t%u\t%llu\t%g\t%g\t%.3g\t%.3g\n", (long long)s.beg + 1, (long long)s.end, s.n_selected_reads, s.n_covered_bases,
            100.0 * s.n_covered_bases / region_len, s.summed_coverage / region_len,
            s.quality_bases > 0 ? s.summed_baseQ / (double)s.quality_bases : 0,
            s.n_selected_reads > 0 ? s.summed_mapQ / (double)s.n_selected_reads : 0);
}

// coverage.c:163-173: 1234567 -> "1.2M" (as many decimals as thousands were divided out)
const char *readable_bps(double base_pairs, char *buf)
{
    static const char *const units[] = { "", "K", "M", "G", "T" };
    int i = 0;
    while (base_pairs >= 1000 && i < 4) { base_pairs /= 1000; i++; }
    sprintf(buf, "%.*f%s", i, base_pairs, units[i]);
    return buf;
}

// coverage.c:150-161: a label centred in a ten-character field of the x axis
const char *center_text(const char *text, char *buf, int width)
{
    const int len = (int)strlen(text), padding = (width - len) / 2, padding_ex = (width - len) % 2;
    if (padding >= 1) sprintf(buf, " %*s%*s", len + padding, text, padding - 1 + padding_ex, " ");
    else sprintf(buf, "%s", text);
    return buf;
}

// coverage.c:223-304: ten text rows over hist_size columns; a column's height is its bin value relative to the largest bin, drawn
// in eighths of a row with the UTF-8 block elements (or halves with '.' ':' for -A), the contig's statistics to the right
void print_hist(FILE *out, const Header &h, const std::vector<CStats> &stats, int tid, const uint32_t *hist, int hist_size, bool full_utf, bool plot_coverage)
{
    static const char *const blocks8[8] = { "\xE2\x96\x81", "\xE2\x96\x82", "\xE2\x96\x83", "\xE2\x96\x84", "\xE2\x96\x85", "\xE2\x96\x86", "\xE2\x96\x87", "\xE2\x96\x88" };
    static const char *const blocks2[2] = { ".", ":" };
    static const char *const vline = "\xE2\x94\x82";
    const CStats &s = stats[(size_t)tid];
    const int n_rows = 10, blockchar_len = full_utf ? 8 : 2;
    const char *const *block = full_utf ? blocks8 : blocks2;
    const double region_len = (double)(s.end - s.beg);
    std::vector<double> hist_data((size_t)(hist_size > 0 ? hist_size : 0));
    double max_val = 0.0;
    for (int i = 0; i < hist_size; ++i) {
        hist_data[(size_t)i] = (plot_coverage ? 1 : 100) * hist[i] / (double)s.bin_width;
        if (hist_data[(size_t)i] > max_val) max_val = hist_data[(size_t)i];
    }
    char buf[64], buf2[64];
    fprintf(out, "%s (%sbp)\n", h.names[(size_t)tid].c_str(), readable_bps((double)h.lens[(size_t)tid], buf));
    const double row_bin_size = max_val / (double)n_rows;
    for (int i = n_rows - 1; i >= 0; --i) {
        const double current_bin = row_bin_size * i;
        if (plot_coverage) fprintf(out, ">%8.1f ", i * row_bin_size);
        else fprintf(out, ">%7.2f%% ", current_bin);
        fputs(full_utf ? vline : "|", out);
        for (int col = 0; col < hist_size; ++col) {
            int cur_val_diff = (int)(round(blockchar_len * (hist_data[(size_t)col] - current_bin) / row_bin_size) - 1);
            if (cur_val_diff < 0) fputc(' ', out);
            else fputs(block[cur_val_diff >= blockchar_len ? blockchar_len - 1 : cur_val_diff], out);
        }
        fputs(full_utf ? vline : "|", out);
        fputc(' ', out);
        switch (i) {
        case 9: fprintf(out, "Number of reads: %u", s.n_selected_reads); break;
        case 8: if (s.n_reads - s.n_selected_reads > 0) fprintf(out, "    (%i filtered)", (int)(s.n_reads - s.n_selected_reads)); break;
        case 7: fprintf(out, "Covered bases:   %sbp", readable_bps((double)s.n_covered_bases, buf)); break;
        case 6: fprintf(out, "Percent covered: %.4g%%", 100.0 * s.n_covered_bases / region_len); break;
        case 5: fprintf(out, "Mean coverage:   %.3gx", s.summed_coverage / region_len); break;
        case 4: fprintf(out, "Mean baseQ:      %.3g", s.quality_bases > 0 ? s.summed_baseQ / (double)s.quality_bases : 0); break;
        case 3: fprintf(out, "Mean mapQ:       %.3g", s.summed_mapQ / (double)s.n_selected_reads); break;
        case 1: fprintf(out, "Histo bin width: %sbp", readable_bps((double)s.bin_width, buf)); break;
        case 0: if (plot_coverage) fprintf(out, "Histo max cov:   %.5g", max_val); else fprintf(out, "Histo max bin:   %.5g%%", max_val); break;
        }
        fputc('\n', out);
    }
    // the x axis: a label every ten columns, the region's end after the remainder
    fprintf(out, "     %s", center_text(readable_bps((double)(s.beg + 1), buf), buf2, 10));
    for (int rest = 10; rest < 10 * (hist_size / 10); rest += 10)
        fprintf(out, "%s", center_text(readable_bps((double)(s.beg + s.bin_width * rest), buf), buf2, 10));
    fprintf(out, "%*s%s", hist_size % 10, " ", center_text(readable_bps((double)s.end, buf), buf2, 10));
    fprintf(out, "\n");
}

}  // namespace

extern "C" int sta_main_coverage(int argc, char **argv)
{
    if (getenv("STA_COV_ITERATOR")) return sta_main_coverage_iter(argc, argv);
    int c, i, max_depth = 1000000, opt_min_baseQ = 0, opt_min_mapQ = 0, opt_min_len = 0, mindepth = 1;
    int fail_flags = 4 | 256 | 512 | 1024, required_flags = 0;
    bool opt_print_header = true, opt_print_tabular = true, opt_print_histogram = false, opt_plot_coverage = false, opt_full_utf = true, opt_full_width = true;
    int opt_n_bins = 50;
    const char *opt_reg = nullptr, *opt_output_file = nullptr, *opt_file_list = nullptr;
    static const struct option lopts[] = {
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 }, { "incl-flags", required_argument, NULL, 1 },
        { "excl-flags", required_argument, NULL, 2 }, { "min-read-len", required_argument, NULL, 'l' }, { "min-MQ", required_argument, NULL, 'q' },
        { "min-mq", required_argument, NULL, 'q' }, { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "histogram", no_argument, NULL, 'm' }, { "ascii", no_argument, NULL, 'A' }, { "plot-depth", no_argument, NULL, 'D' },
        { "output", required_argument, NULL, 'o' }, { "no-header", no_argument, NULL, 'H' }, { "n-bins", required_argument, NULL, 'w' },
        { "region", required_argument, NULL, 'r' }, { "depth", required_argument, NULL, 'd' }, { "min-depth", required_argument, NULL, 3 },
        { "bam-list", required_argument, NULL, 'b' }, { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt_long(argc, argv, "Ao:l:q:Q:hHw:r:b:md:D", lopts, NULL)) >= 0) {
        switch (c) {
        case 1: if ((required_flags = str2flag(optarg)) < 0) { fprintf(stderr, "Could not parse --rf %s\n", optarg); return 1; } break;
        case 2: if ((fail_flags = str2flag(optarg)) < 0) { fprintf(stderr, "Could not parse --ff %s\n", optarg); return 1; } break;
        case 3: if ((i = atoi(optarg)) > 0) mindepth = i; break;
        case 'o': opt_output_file = optarg; opt_full_width = false; break;
        case 'l': opt_min_len = atoi(optarg); break;
        case 'q': opt_min_mapQ = atoi(optarg); break;
        case 'Q': opt_min_baseQ = atoi(optarg); break;
        case 'd': max_depth = atoi(optarg); break;
        case 'r': opt_reg = optarg; break;
        case 'H': opt_print_header = false; break;
        case 'w': opt_n_bins = atoi(optarg); opt_full_width = false; opt_print_histogram = true; opt_print_tabular = false; break;
        case 'b': opt_file_list = optarg; break;
        case 'm': opt_print_histogram = true; opt_print_tabular = false; break;
        case 'A': opt_full_utf = false; opt_print_histogram = true; opt_print_tabular = false; break;
        case 'D': opt_print_histogram = true; opt_print_tabular = false; opt_plot_coverage = true; break;
        default: fprintf(stderr, "Usage: samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1;
        }
    }
    if (optind == argc && !opt_file_list) { fprintf(stderr, "Usage: samtools coverage [options] in1.bam [in2.bam [...]]\n"); return 1; }
    FILE *file_out = stdout;
    if (opt_output_file && strcmp(opt_output_file, "-") != 0) {
        file_out = fopen(opt_output_file, "w");
        if (!file_out) { fprintf(stderr, "samtools coverage: Cannot open \"%s\" for writing.\n", opt_output_file); return 1; }
    }
    if (opt_n_bins <= 0 || opt_full_width) {
        // coverage.c:427-451: the terminal's width ($COLUMNS, else the tty behind stderr) less the 40 characters beside the plot
        int columns = 0;
        if (const char *env_columns = getenv("COLUMNS")) columns = atoi(env_columns);
        else { struct winsize w; if (ioctl(2, TIOCGWINSZ, &w) == 0) columns = w.ws_col; }
        opt_n_bins = columns > 60 ? columns - 40 : 40;
    }
    std::vector<std::string> fns;
    if (opt_file_list) {
        if (!read_file_list(opt_file_list, &fns)) { fprintf(stderr, "samtools coverage: Cannot open file list \"%s\".\n", opt_file_list); return 1; }
    } else for (i = optind; i < argc; ++i) fns.push_back(argv[i]);
    const int nfiles = (int)fns.size();
    std::vector<std::unique_ptr<AlnReader>> readers;
    std::vector<CStats> stats;
    int reg_tid = -1; int64_t reg_beg = 0, reg_end = INT64_MAX;
    for (i = 0; i < nfiles; ++i) {
        std::string err;
        auto r = AlnReader::open(fns[(size_t)i], &err);
        if (!r) { fprintf(stderr, "samtools coverage: Could not open \"%s\"\n", fns[(size_t)i].c_str()); return 1; }
        if (opt_reg) {
            int t; int64_t b, e;
            if (!parse_region(r->header(), opt_reg, &t, &b, &e)) {
                fprintf(stderr, "samtools coverage: Failed to parse region \"%s\". Check the region format or region name presence in the file \"%s\"\n", opt_reg, fns[(size_t)i].c_str());
                return 1;
            }
            r->set_region(t, b, e);
            if (i == 0) { reg_tid = t; reg_beg = b; reg_end = e; }
        }
        readers.push_back(std::move(r));
    }
    const Header &h = readers[0]->header();
    const int n_targets = h.nref();
    stats.assign((size_t)(n_targets > 0 ? n_targets : 1), CStats());
    int64_t n_bins = opt_n_bins;
    std::vector<uint32_t> hist((size_t)opt_n_bins, 0u);
    if (opt_reg) {
        CStats &s = stats[(size_t)reg_tid];
        s.beg = reg_beg; s.end = reg_end;
        if (s.end == INT64_MAX || s.end > h.lens[(size_t)reg_tid]) s.end = h.lens[(size_t)reg_tid];
        n_bins = opt_n_bins > s.end - s.beg ? s.end - s.beg : opt_n_bins;
        s.bin_width = (s.end - s.beg) / (n_bins > 0 ? n_bins : 1);
    }
    // read-level statistics: what coverage.c's read_bam callback counts (coverage.c:182-196)
    for (auto &r : readers) {
        const int nref = r->header().nref();
        r->on_record = [&stats, nref, fail_flags, required_flags, opt_min_mapQ, opt_min_len](const Rec &rec) {
            if (rec.tid < 0 || rec.tid >= nref || (size_t)rec.tid >= stats.size()) return;
            CStats &s = stats[(size_t)rec.tid];
            s.n_reads++;
            if (fail_flags && (rec.flag & fail_flags)) return;
            if (required_flags && !(rec.flag & required_flags)) return;
            if (rec.mapq < opt_min_mapQ) return;
            if (opt_min_len && cigar2qlen(rec) < opt_min_len) return;
            s.n_selected_reads++; s.summed_mapQ += rec.mapq;
        };
    }
    sta_engine *eng = nullptr;
    if (sta_engine_create(&eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools coverage: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 1; }
    sta_cov_params cp; memset(&cp, 0, sizeof cp);
    cp.mode = 0; cp.min_baseQ = opt_min_baseQ; cp.min_depth = mindepth; cp.max_depth = max_depth > 0 ? max_depth : INT_MAX;
    cp.min_mq = opt_min_mapQ; cp.rflag_require = required_flags; cp.rflag_filter = fail_flags; cp.min_qlen = opt_min_len;
    int64_t window_cols = 1 << 22;
    if (const char *e = getenv("STA_WINDOW_COLS")) window_cols = std::max<long long>(1, atoll(e));
    PumpConfig pc; pc.window_cols = window_cols; pc.max_reads = 4 << 20; pc.use_endpos = false; pc.nref_limit = readers[0]->header().nref();
    Pump pump(readers, pc);
    int status = 0, last_tid = -1;
    bool warn = false;
    for (;;) {
        int tid = pump.next_tid();
        if (pump.error() || tid < 0) break;
        if (tid >= n_targets) { std::vector<std::vector<const Rec *>> dump; pump.fill(tid, 0, INT64_MAX, dump); pump.drop_tid_carry(); continue; }
        CStats &s = stats[(size_t)tid];
        if (!opt_reg) s.end = h.lens[(size_t)tid];
        int64_t tid_bins = 0;
        if (opt_print_histogram && s.end > s.beg) {
            // coverage.c:606-609: at most one bin per base of the contig; the histogram is kept on the device while its windows run
            tid_bins = opt_n_bins > s.end - s.beg ? s.end - s.beg : opt_n_bins;
            s.bin_width = (s.end - s.beg) / tid_bins;
            cp.hist_bins = (int32_t)tid_bins; cp.hist_depth = opt_plot_coverage ? 1 : 0; cp.hist_beg = s.beg; cp.hist_bin_width = s.bin_width;
            if (sta_cov_hist_begin(eng, (int32_t)tid_bins) != STA_OK) { fprintf(stderr, "samtools coverage: %s\n", sta_last_error(eng)); status = 1; break; }
        }
        CovAccum acc((size_t)nfiles);
        if (cov_run_tid(eng, pump, readers, tid, s.beg, s.end, cp, window_cols, false, acc, "coverage") < 0) { status = 1; break; }
        s.n_covered_bases = acc.tot.n_covered_bases; s.summed_coverage = acc.tot.summed_coverage;
        s.summed_baseQ = acc.tot.summed_baseQ; s.quality_bases = acc.tot.quality_bases;
        warn |= acc.tot.missing_qual != 0;
        if (acc.n_kept) {                      // the iterator returned at least one column of this contig
            s.covered = true;
            if (opt_print_histogram) {
                // the previous contig's plot was followed by an empty line when this one turned up (coverage.c:592-596)
                if (last_tid >= 0) fputc('\n', file_out);
                n_bins = tid_bins;
                if (tid_bins > 0 && sta_cov_hist_fetch(eng, hist.data(), (int32_t)tid_bins) != STA_OK) { fprintf(stderr, "samtools coverage: %s\n", sta_last_error(eng)); status = 1; break; }
                print_hist(file_out, h, stats, tid, hist.data(), (int)n_bins, opt_full_utf, opt_plot_coverage);
            } else print_tabular_line(file_out, h, stats, tid, &opt_print_header);
            last_tid = tid;
        }
    }
    if (pump.error()) { fprintf(stderr, "samtools coverage: %s\n", pump.error_text()); status = 1; }
    if (!status) {
        if (last_tid == -1 && opt_reg && *opt_reg != '*') {
            if (opt_print_histogram) { std::fill(hist.begin(), hist.end(), 0u); print_hist(file_out, h, stats, reg_tid, hist.data(), (int)n_bins, opt_full_utf, opt_plot_coverage); }
            else print_tabular_line(file_out, h, stats, reg_tid, &opt_print_header);
        }
        if (!opt_reg && opt_print_tabular)
            for (i = 0; i < n_targets; ++i)
                if (!stats[(size_t)i].covered) { stats[(size_t)i].end = h.lens[(size_t)i]; print_tabular_line(file_out, h, stats, i, &opt_print_header); }
        if (warn) fprintf(stderr, "samtools coverage: Warning:  Missing quality values in alignments.  Mean base quality calculated only on available values.\n");
    }
    sta_engine_destroy(eng);
    if (file_out != stdout) fclose(file_out);
    return status;
}

// ---------------------------------------------------------------------------------------------- bedcov
namespace {
void bedcov_header(FILE *fp, const char *hdr, int fields, int n, char **fn, int depth, int rcount)
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

extern "C" int sta_main_bedcov(int argc, char **argv)
{
    if (getenv("STA_COV_ITERATOR")) return sta_main_bedcov_iter(argc, argv);
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
    sta_engine *eng = nullptr;
    if (sta_engine_create(&eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools bedcov: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::string err;
    auto r0 = AlnReader::open(fn[0], &err);
    if (!r0) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[0]); return 2; }
    const Header h0 = r0->header();
    FILE *fp = fopen(argv[optind], "r");
    if (!fp) { fprintf(stderr, "samtools bedcov: can't open BED file '%s'\n", argv[optind]); return 2; }
    sta_cov_params cp; memset(&cp, 0, sizeof cp);
    cp.mode = 1; cp.min_depth = min_depth; cp.skip_dn = skip_DN; cp.max_depth = min_depth > max_depth ? min_depth : max_depth;
    cp.min_mq = min_mapQ; cp.rflag_filter = (int32_t)flags;
    char *line = nullptr; size_t cap = 0; ssize_t len;
    while ((len = getline(&line, &cap, fp)) >= 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (len == 0) continue;
        if (line[0] == '#') {
            if (!hdr && !strncmp(line, "#chrom", 6)) { bedcov_header(stdout, line, -1, n, fn, min_depth, do_rcount); hdr = 1; }
            continue;
        }
        if (strncmp(line, "track ", 6) == 0 || strncmp(line, "browser ", 8) == 0) continue;
        if (!hdr) {
            int fields = 0;
            for (char *t = line; *t; ++t) if (*t == '\t') fields++;
            bedcov_header(stdout, NULL, fields + 1, n, fn, min_depth, do_rcount);
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
        std::vector<std::unique_ptr<AlnReader>> readers;
        for (int i = 0; i < n; ++i) {
            auto r = AlnReader::open(fn[i], &err);
            if (!r) { fprintf(stderr, "ERROR: fail to open index BAM file '%s'\n", fn[i]); return 2; }
            r->set_region(tid, beg, end);               // sam_itr_queryi(idx, tid, beg, end)
            readers.push_back(std::move(r));
        }
        CovAccum acc((size_t)n);
        PumpConfig pc; pc.window_cols = 1 << 22; pc.max_reads = 4 << 20; pc.use_endpos = false; pc.nref_limit = h0.nref();
        Pump pump(readers, pc);
        int t0 = pump.next_tid();
        if (!pump.error() && t0 == tid && end > beg) {
            if (cov_run_tid(eng, pump, readers, tid, beg, std::min<int64_t>(end, h0.lens[(size_t)tid]), cp, pc.window_cols, do_rcount != 0, acc, "bedcov") < 0) { status = 2; break; }
        } else if (!pump.error() && t0 == tid && do_rcount) {
            // empty interval: reads can still enter the iterator (and be counted by -c) without producing a column in range
            if (cov_run_tid(eng, pump, readers, tid, beg, beg + 1, cp, pc.window_cols, true, acc, "bedcov") < 0) { status = 2; break; }
            std::fill(acc.per_file.begin(), acc.per_file.end(), 0);
        }
        if (pump.error()) { fprintf(stderr, "samtools bedcov: error reading from input file\n"); status = 2; break; }
        fputs(line, stdout);
        for (int i = 0; i < n; ++i) printf("\t%lld", (long long)acc.per_file[(size_t)i * 2]);
        if (min_depth >= 0) for (int i = 0; i < n; ++i) printf("\t%lld", (long long)acc.per_file[(size_t)i * 2 + 1]);
        if (do_rcount) for (int i = 0; i < n; ++i) printf("\t%lld", (long long)acc.kept[(size_t)i]);
        putchar('\n');
    }
    free(line); fclose(fp);
    sta_engine_destroy(eng);
    return status;
}
