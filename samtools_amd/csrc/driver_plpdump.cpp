// driver_plpdump.cpp -- `samtools-amd plpdump`: walks alignment files through the bam_plp_* /
// bam_mplp_* / bam_plbuf_* surface (include/samtools_amd_plp.h) and prints every bam_pileup1_t.
// A consumer of the callback surface in the style of the reference's small pileup clients
// (bam_plbuf.c, bedcov.c:316-333); the parity tests diff its output against the oracle's iterator.
//   plpdump [-x] [-d maxcnt] [-p] in1.sam [in2.sam ...]     (-x: no mate-overlap handling, -p: push style via bam_plbuf)
#include "../../include/samtools_amd.h"
#include "../../include/samtools_amd_plp.h"
#include "host_io.h"
#include "bam1_from_rec.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <string>
#include <vector>

using namespace sta;

namespace {

struct Src { std::unique_ptr<AlnReader> rd; Rec rec; };

int read_cb(void *data, bam1_t *b)
{
    Src *s = (Src *)data;
    int ret = s->rd->next(s->rec);
    if (ret == 0) return -1;
    if (ret < 0) return -2;
    rec_to_bam1(s->rec, b);
    return 0;
}

kstring_t g_ins = { 0, 0, nullptr };

void print_entries(FILE *out, int n, const bam_pileup1_t *plp)
{
    fprintf(out, "\t%d", n);
    for (int i = 0; i < n; ++i) {
        const bam_pileup1_t *p = &plp[i];
        int q = p->qpos < p->b->core.l_qseq ? bam_get_qual(p->b)[p->qpos] : -1;
        int del_len = 0;
        int il = sta_bam_plp_insertion(p, &g_ins, &del_len);
        fprintf(out, "\t%s,%d,%d,%d,%d%d%d%d,%d,%d,%s,%d", bam_get_qname(p->b), p->b->core.flag, p->qpos, p->indel, (int)p->is_del, (int)p->is_head,
                (int)p->is_tail, (int)p->is_refskip, p->cigar_ind, q, il > 0 ? g_ins.s : ".", del_len);
    }
}

int plbuf_cb(uint32_t tid, hts_pos_t pos, int n, const bam_pileup1_t *pl, void *data)
{
    FILE *out = (FILE *)data;
    fprintf(out, "%u\t%lld", tid, (long long)pos);
    print_entries(out, n, pl);
    fputc('\n', out);
    return 0;
}

}  // namespace

extern "C" int sta_main_plpdump(int argc, char **argv)
{
    bool overlaps = true, push = false;
    int maxcnt = 8000, c;
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt(argc, argv, "xd:p")) >= 0) {
        if (c == 'x') overlaps = false;
        else if (c == 'd') maxcnt = atoi(optarg);
        else if (c == 'p') push = true;
        else return 1;
    }
    std::vector<Src> src((size_t)(argc - optind));
    if (src.empty()) { fprintf(stderr, "Usage: samtools-amd plpdump [-x] [-d maxcnt] [-p] in.sam [...]\n"); return 1; }
    for (size_t i = 0; i < src.size(); ++i) {
        std::string err;
        src[i].rd = AlnReader::open(argv[optind + (int)i], &err);
        if (!src[i].rd) { fprintf(stderr, "[plpdump] failed to open %s\n", argv[optind + (int)i]); return 1; }
    }
    FILE *out = stdout;
    int ret = 0;
    if (push) {
        // bam_plbuf style: push records of the first file one by one, NULL at the end
        sta_bam_plbuf_t *buf = sta_bam_plbuf_init(plbuf_cb, out);
        bam1_t b; memset(&b, 0, sizeof b);
        int r;
        while ((r = read_cb(&src[0], &b)) >= 0)
            if (sta_bam_plbuf_push(&b, buf) < 0) { ret = 1; break; }
        if (r < -1) { fprintf(stderr, "[plpdump] error reading from input file\n"); ret = 1; }
        if (!ret && sta_bam_plbuf_push(nullptr, buf) < 0) ret = 1;
        sta_bam_plbuf_destroy(buf);
        free(b.data);
    } else {
        std::vector<void *> data(src.size());
        for (size_t i = 0; i < src.size(); ++i) data[i] = &src[i];
        sta_bam_mplp_t it = sta_bam_mplp_init((int)src.size(), read_cb, data.data());
        if (overlaps) sta_bam_mplp_init_overlaps(it);
        sta_bam_mplp_set_maxcnt(it, maxcnt);
        std::vector<int> n_plp(src.size());
        std::vector<const bam_pileup1_t *> plp(src.size());
        int tid = 0, r; hts_pos_t pos = 0;
        while ((r = sta_bam_mplp64_auto(it, &tid, &pos, n_plp.data(), plp.data())) > 0) {
            fprintf(out, "%d\t%lld", tid, (long long)pos);
            for (size_t i = 0; i < src.size(); ++i) print_entries(out, n_plp[i], plp[i]);
            fputc('\n', out);
        }
        if (r < 0) { fprintf(stderr, "[plpdump] error reading from input file\n"); ret = 1; }
        sta_bam_mplp_destroy(it);
    }
    free(g_ins.s); g_ins.s = nullptr; g_ins.m = g_ins.l = 0;
    return ret;
}
