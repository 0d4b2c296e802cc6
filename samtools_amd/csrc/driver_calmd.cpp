// driver_calmd.cpp -- `samtools-amd calmd`: calmd's per-record arithmetic on the engine.
// The loop of bam_fillmd (bam_md.c:457-497) restated over batches: records of one contig are staged as a window (the window is
// only a coordinate frame here), sta_calmd_plan runs BAQ (-r) and the MD / NM kernels, and the fields calmd changes are dumped:
//   calmd [-e] [-r] [-A] [-E] [-q] [-n max_nm] in.bam ref.fa
//   qname  flag  rname  pos  mapq  NM|*  MD|*  SEQ  QUAL  BQ:Z:..|ZQ:Z:..|ZQ<-BQ|*
// It is not a SAM/BAM writer: sam_write1 and the aux re-encoding are HTSlib I/O, outside the hot path.
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_stage.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>

using namespace sta;

namespace {

struct Ctx {
    sta_engine *eng = nullptr;
    sta_calmd_params cp{};
    const Header *h = nullptr;
    std::vector<Rec> batch;
    StagedFile staged;
    std::vector<int32_t> nm; std::vector<uint64_t> off; std::vector<char> md; std::vector<uint8_t> state, qual, seq, tag;
    std::string line;
};

void print_plain(const Header &h, const Rec &r)
{
    static const char nt[] = "=ACMGRSVTWYHKDBN";
    printf("%s\t%d\t%s\t%lld\t%d\t*\t*\t", r.qname.c_str(), (int)r.flag, r.tid >= 0 ? h.names[(size_t)r.tid].c_str() : "*", (long long)r.pos + 1, (int)r.mapq);
    if (r.l_qseq == 0) fputs("*\t*\t", stdout);
    else {
        for (int i = 0; i < r.l_qseq; ++i) putchar(nt[(r.seq[(size_t)i >> 1] >> ((~i & 1) << 2)) & 0xf]);
        putchar('\t');
        if (r.qual[0] == 0xff) putchar('*'); else for (int i = 0; i < r.l_qseq; ++i) putchar(r.qual[(size_t)i] + 33);
        putchar('\t');
    }
    fputs("*\n", stdout);
}

// one batch = records of one contig in non-decreasing position order
int flush(Ctx &c, int tid, const std::string *ref)
{
    if (c.batch.empty()) return 0;
    if (!ref) { for (const Rec &r : c.batch) print_plain(*c.h, r); c.batch.clear(); return 0; }
    const int64_t origin = c.batch.front().pos;
    int64_t hi = origin + 1;
    c.staged.clear();
    for (const Rec &r : c.batch) { c.staged.add(r, origin, nullptr); hi = std::max(hi, r.end() + 1); }
    c.staged.finish();
    sta_reads view = c.staged.view();
    if (!(c.cp.flag & STA_CALMD_APPLY)) view.bq = nullptr;
    sta_window w; memset(&w, 0, sizeof w);
    w.tid = tid; w.origin = origin; w.col_beg = 0; w.col_end = (int32_t)std::min<int64_t>(hi - origin, INT32_MAX - 1);
    w.tname = c.h->names[(size_t)tid].c_str(); w.tlen = c.h->lens[(size_t)tid];
    w.n_files = 1; w.files = &view; w.mem = STA_MEM_HOST;
    sta_plan_info pi;
    if (sta_stage_window(c.eng, &w) != STA_OK || sta_calmd_plan(c.eng, &c.cp, &pi) != STA_OK) { fprintf(stderr, "samtools calmd: %s\n", sta_last_error(c.eng)); return -1; }
    const size_t n = c.batch.size();
    c.nm.resize(n); c.off.resize(n + 1); c.md.resize((size_t)pi.out_bytes + 1); c.state.resize(n);
    c.qual.resize(c.staged.qual.size()); c.seq.resize(c.staged.seq.size() + 1); c.tag.resize(c.staged.qual.size());
    if (sta_fetch_calmd(c.eng, c.nm.data(), c.off.data(), c.md.data(), c.state.data(), c.qual.data(), c.seq.data(), c.tag.data()) != STA_OK) {
        fprintf(stderr, "samtools calmd: %s\n", sta_last_error(c.eng)); return -1;
    }
    static const char nt[] = "=ACMGRSVTWYHKDBN";
    for (size_t i = 0; i < n; ++i) {
        const Rec &r = c.batch[i];
        const size_t boff = (size_t)c.staged.base_off8[i] << 3;
        std::string &s = c.line; s.clear();
        char num[64];
        s += r.qname; snprintf(num, sizeof num, "\t%d\t", (int)r.flag); s += num;
        s += c.h->names[(size_t)tid]; snprintf(num, sizeof num, "\t%lld\t%d\t", (long long)r.pos + 1, (int)r.mapq); s += num;
        if (c.state[i] & STA_CALMD_HAS_MD) {
            snprintf(num, sizeof num, "%d\t", c.nm[i]); s += num;
            s.append(c.md.data() + c.off[i], (size_t)(c.off[i + 1] - c.off[i])); s += '\t';
        } else s += "*\t*\t";
        if (r.l_qseq == 0) s += "*\t*\t";
        else {
            for (int k = 0; k < r.l_qseq; ++k) s += nt[(c.seq[(boff >> 1) + ((size_t)k >> 1)] >> ((~k & 1) << 2)) & 0xf];
            s += '\t';
            if (c.qual[boff] == 0xff) s += '*'; else for (int k = 0; k < r.l_qseq; ++k) s += (char)(c.qual[boff + (size_t)k] + 33);
            s += '\t';
        }
        if (c.state[i] & STA_CALMD_NEW_TAG) {
            s += (c.cp.flag & STA_CALMD_APPLY) ? "ZQ:Z:" : "BQ:Z:";
            s.append((const char *)c.tag.data() + boff, (size_t)r.l_qseq);
        } else if (c.state[i] & STA_CALMD_BQ_TO_ZQ) s += "ZQ<-BQ";
        else s += '*';
        s += '\n';
        fwrite(s.data(), 1, s.size(), stdout);
    }
    c.batch.clear();
    return 0;
}

}  // namespace

extern "C" int sta_main_calmd(int argc, char **argv)
{
    Ctx c;
    int o;
    optind = 1;
    while ((o = getopt(argc, argv, "erAEqn:C:dhQ")) >= 0) {
        switch (o) {
        case 'e': c.cp.flag |= STA_CALMD_USE_EQUAL; break;
        case 'r': c.cp.flag |= STA_CALMD_REALN; break;
        case 'A': c.cp.flag |= STA_CALMD_APPLY; break;
        case 'E': c.cp.flag |= STA_CALMD_EXTENDED; break;
        case 'q': c.cp.flag |= STA_CALMD_BIN_QUAL; break;
        case 'n': c.cp.max_nm = atoi(optarg); break;
        case 'Q': break;
        default: fprintf(stderr, "[calmd] option -%c is not part of the engine's rows\n", o); return 1;
        }
    }
    if (argc - optind != 2) { fprintf(stderr, "usage: samtools-amd calmd [-erAEq] [-n max_nm] in.bam ref.fa\n"); return 1; }
    if (sta_device_count() < 1) { fprintf(stderr, "samtools calmd: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::string err;
    auto rd = AlnReader::open(argv[optind], &err);
    if (!rd) { fprintf(stderr, "samtools calmd: %s\n", err.c_str()); return 1; }
    c.h = &rd->header();
    auto fa = Fasta::load(argv[optind + 1]);
    if (!fa) { fprintf(stderr, "samtools calmd: Failed to open reference file '%s'\n", argv[optind + 1]); return 1; }
    if (sta_engine_create(&c.eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools calmd: no usable HIP device\n"); return 2; }
    const bool realn = (c.cp.flag & STA_CALMD_REALN) != 0;
    size_t max_batch = 1 << 18;
    if (const char *e = getenv("STA_CALMD_BATCH")) max_batch = (size_t)std::max<long long>(1, atoll(e));
    int status = 0, cur_tid = -2;
    const std::string *ref = nullptr;
    unsigned skipped = 0;
    Rec r;
    int st;
    while ((st = rd->next(r)) > 0) {
        const bool new_run = r.tid != cur_tid || (!c.batch.empty() && (r.pos < c.batch.back().pos || r.pos - c.batch.front().pos > (1 << 30)));
        if (new_run || c.batch.size() >= max_batch) {
            if (flush(c, cur_tid, ref) < 0) { status = 1; break; }
        }
        if (r.tid != cur_tid) {
            cur_tid = r.tid; ref = nullptr;
            if (r.tid >= 0) {
                ref = fa->fetch(c.h->names[(size_t)r.tid]);
                sta_clear_references(c.eng);
                if (!ref) {
                    fprintf(stderr, "[bam_fillmd] fail to find sequence '%s' in the reference.\n", c.h->names[(size_t)r.tid].c_str());
                    if (realn) { status = 1; break; }
                } else if (sta_set_reference(c.eng, r.tid, ref->data(), (int64_t)ref->size(), STA_MEM_HOST) != STA_OK) { status = 1; break; }
            }
        }
        if (r.tid < 0) { print_plain(*c.h, r); continue; }
        if (ref && r.l_qseq == 0) ++skipped;
        c.batch.push_back(r);
    }
    if (!status && flush(c, cur_tid, ref) < 0) status = 1;
    if (st < 0) { fprintf(stderr, "[bam_fillmd] Error reading input.\n"); status = 1; }
    if (skipped) fprintf(stderr, "[calmd] Warning: %u records skipped due to no query sequence\n", skipped);
    sta_engine_destroy(c.eng);
    return status;
}
