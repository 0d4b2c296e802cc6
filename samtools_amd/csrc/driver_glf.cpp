// driver_glf.cpp -- `samtools-amd glf`: per-column output of the genotype-likelihood packer (row a14).
// tview is the reference's only caller of bcf_call_glfgen (bam_tview.c:197) and is a UI, out of scope; this driver exposes
// the same per-column values as text so that they can be diffed against the oracle:
//   glf [-Q min_baseQ] [-t theta] [-f ref.fa] in.bam
//   name  pos  n_plp  n  flags  qsum[4] (float bits)  p[25] (float bits)  consensus-char  consensus-qual
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_pump.h"
#include "host_stage.h"
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>

using namespace sta;

extern "C" int sta_main_glf(int argc, char **argv)
{
    sta_glf_params gp; memset(&gp, 0, sizeof gp);
    gp.min_baseQ = 13; gp.max_depth = 8000; gp.theta = 0.83;
    const char *fa_fn = nullptr;
    int c;
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt(argc, argv, "Q:t:f:")) >= 0) {
        if (c == 'Q') gp.min_baseQ = atoi(optarg);
        else if (c == 't') gp.theta = atof(optarg);
        else if (c == 'f') fa_fn = optarg;
        else return 1;
    }
    if (argc - optind != 1) { fprintf(stderr, "usage: samtools-amd glf [-Q min_baseQ] [-t theta] [-f ref.fa] in.bam\n"); return 1; }
    if (sta_device_count() < 1) { fprintf(stderr, "samtools glf: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::string err;
    std::vector<std::unique_ptr<AlnReader>> readers;
    readers.push_back(AlnReader::open(argv[optind], &err));
    if (!readers[0]) { fprintf(stderr, "samtools glf: %s\n", err.c_str()); return 1; }
    const Header &h = readers[0]->header();
    std::unique_ptr<Fasta> fa;
    if (fa_fn) { fa = Fasta::load(fa_fn); if (!fa) { fprintf(stderr, "samtools glf: failed to load %s\n", fa_fn); return 1; } }
    sta_engine *eng = nullptr;
    if (sta_engine_create(&eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools glf: no usable HIP device\n"); return 2; }
    int64_t window_cols = 1 << 20;
    if (const char *e = getenv("STA_WINDOW_COLS")) window_cols = std::max<long long>(1, atoll(e));
    PumpConfig pc; pc.window_cols = window_cols; pc.use_endpos = false; pc.nref_limit = readers[0]->header().nref();
    Pump pump(readers, pc);
    StagedFile staged;
    std::vector<std::vector<const Rec *>> reads;
    std::vector<sta_glf_col> cols;
    std::vector<uint32_t> info;
    int status = 0;
    for (;;) {
        int tid = pump.next_tid();
        if (pump.error() || tid < 0) break;
        const std::string *ref = fa ? fa->fetch(h.names[(size_t)tid]) : nullptr;
        sta_clear_references(eng);
        if (ref && sta_set_reference(eng, tid, ref->data(), (int64_t)ref->size(), STA_MEM_HOST) != STA_OK) { status = 1; break; }
        int64_t cursor = pump.next_pos(tid);
        for (;;) {
            if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
            if (!pump.has_carry()) cursor = std::max(cursor, pump.next_pos(tid));
            int64_t ce = pump.fill(tid, cursor, cursor + window_cols, reads);
            if (pump.error()) break;
            if (pump.next_pos(tid) == INT64_MAX) {
                int64_t me = pump.carry_max_end();
                if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
            }
            if (ce > cursor) {
                staged.clear();
                for (const Rec *r : reads[0]) staged.add(*r, cursor, nullptr);
                staged.finish();
                sta_reads view = staged.view();
                sta_window w; memset(&w, 0, sizeof w);
                w.tid = tid; w.origin = cursor; w.col_beg = 0; w.col_end = (int32_t)(ce - cursor);
                w.tname = h.names[(size_t)tid].c_str(); w.tlen = h.lens[(size_t)tid];
                w.n_files = 1; w.files = &view; w.mem = STA_MEM_HOST;
                sta_plan_info pi;
                if (sta_stage_window(eng, &w) != STA_OK || sta_glf_plan(eng, &gp, &pi) != STA_OK) { fprintf(stderr, "samtools glf: %s\n", sta_last_error(eng)); status = 1; break; }
                cols.resize((size_t)(ce - cursor));
                if (sta_fetch_output(eng, (char *)cols.data(), pi.out_bytes) != STA_OK) { fprintf(stderr, "samtools glf: %s\n", sta_last_error(eng)); status = 1; break; }
                for (size_t i = 0; i < cols.size(); ++i) {
                    const sta_glf_col &g = cols[i];
                    if (g.n_plp <= 0) continue;
                    const int64_t pos = cursor + (int64_t)i;
                    const char rb = (ref && pos < (int64_t)ref->size()) ? (*ref)[(size_t)pos] : 'N';
                    char ch = '?';
                    const int q = sta_glf_consensus(&g, rb, &ch);
                    printf("%s\t%lld\t%d\t%d\t%d\t", h.names[(size_t)tid].c_str(), (long long)pos + 1, g.n_plp, g.n, g.flags);
                    for (int k = 0; k < 4; ++k) { uint32_t u; memcpy(&u, &g.qsum[k], 4); printf("%s%08x", k ? "," : "", u); }
                    putchar('\t');
                    for (int k = 0; k < 25; ++k) { uint32_t u; memcpy(&u, &g.p[k], 4); printf("%s%08x", k ? "," : "", u); }
                    printf("\t%c\t%d\n", ch, q);
                }
                if (pi.n_maxcnt_dropped) {
                    info.resize((size_t)staged.n());
                    if (!info.empty() && sta_fetch_read_state(eng, 0, info.data(), nullptr) == STA_OK) {
                        std::vector<char> dr(info.size());
                        for (size_t i = 0; i < info.size(); ++i) dr[i] = (info[i] & 1u) && !(info[i] & 2u) && reads[0][i]->rlen > 0;
                        pump.drop(0, dr);
                    }
                }
            }
            pump.retire(ce);
            cursor = std::max(cursor, ce);
        }
        if (status || pump.error()) break;
        pump.drop_tid_carry();
    }
    if (pump.error()) { fprintf(stderr, "samtools glf: %s\n", pump.error_text()); status = 1; }
    sta_engine_destroy(eng);
    return status;
}
