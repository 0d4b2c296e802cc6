// tests/cpu/cons_emul.cpp -- TEST INFRASTRUCTURE: the consensus command with its window compute emulated on the CPU.
// The product's driver (driver_consensus.cpp), table builder (cons_host.cpp) and per-read / per-column step functions
// (cons_window.h, cons_core.h -- the code the HIP kernels run one lane per read / column) are compiled for the host and the
// steps are executed in plain loops, so that the not-gpu suite can diff the exact logic the device runs against the
// reference's goldens and the oracle.  Nothing here is linked into libsamtools_amd.so; the product refuses to run
// without a device.
#include "../../samtools_amd/csrc/cons_host.h"
#include "../../samtools_amd/csrc/cons_window.h"
#include <algorithm>
#include <cstdio>
#include <cstring>

using namespace sta;

static int emul(const sta_window &w, const sta_cons_params &p, ConsWindowOut &out, std::string &err)
{
    static cons::Tables tab; static bool have = false;
    if (!have) { cons_build_tables(p, tab); have = true; }
    const cons::Par o = cons_par(p);
    const sta_reads &f = w.files[0];
    const int64_t n = f.n_reads;
    const int32_t W = w.col_end - w.col_beg;
    const bool bayes_mq = o.mode != cons::MODE_SIMPLE && o.use_mqual;
    cons::Win d; memset(&d, 0, sizeof d);
    d.n_reads = n; d.pos = f.pos; d.flag = f.flag; d.mapq = f.mapq; d.l_qseq = f.l_qseq; d.cig_off = f.cig_off; d.base_off8 = f.base_off8;
    d.cigar = f.cigar; d.seq = f.seq; d.qual_in = f.qual; d.n_xcols = f.n_xcols; d.xcol_off = f.xcol_off; d.xcol_text = f.xcol_text;
    d.col_beg = w.col_beg; d.col_end = w.col_end;
    std::vector<uint8_t> wq; std::vector<int32_t> nm;
    if (bayes_mq) { wq.assign(f.qual, f.qual + f.n_bases_total); nm.assign(f.n_bases_total + 8, 0); d.qual = wq.data(); d.nm = nm.data(); }
    else d.qual = const_cast<uint8_t *>(f.qual);
    std::vector<uint32_t> ins((size_t)W + 1, 0), keep((size_t)n), cnt((size_t)n);
    std::vector<uint64_t> colbase((size_t)W + 1), rowoff((size_t)n + 1);
    std::vector<int32_t> r_last((size_t)n), r_tail((size_t)n), cs((size_t)n), ce((size_t)n), pmax((size_t)n);
    unsigned long long counters[2] = { 0, 0 };
    d.ins = ins.data(); d.colbase = colbase.data(); d.r_last = r_last.data(); d.r_tail = r_tail.data(); d.r_keep = keep.data();
    d.cs = cs.data(); d.ce = ce.data(); d.pmax = pmax.data(); d.cnt = cnt.data(); d.rowoff = rowoff.data(); d.counters = counters;
    auto amax = [](uint32_t *q, uint32_t v) { if (*q < v) *q = v; };
    for (int64_t r = 0; r < n; ++r) { const int code = cons::step_read_a(d, o, tab, r, amax, true); counters[code < 0 ? 1 : 0] += code != 0; }
    if (counters[1]) { err = "a CIGAR holds an operation outside MIDNSHP=X"; return -1; }
    colbase[0] = 0;
    for (int32_t i = 0; i < W; ++i) colbase[(size_t)i + 1] = colbase[(size_t)i] + 1 + ins[(size_t)i + 1];
    std::vector<int32_t> colpos(colbase[(size_t)W] + 1), clist;
    std::vector<cons::Meta> meta((size_t)n + 1);
    d.meta = meta.data();
    d.colpos = colpos.data();
    for (int32_t i = 0; i < W; ++i) cons::step_colpos(d, i);
    uint64_t sum_depth = 0;
    for (int64_t r = 0; r < n; ++r) { uint32_t alive; if (cons::step_read_b(d, r, alive)) clist.push_back((int32_t)r); sum_depth += alive; }
    rowoff[0] = 0;
    for (int64_t r = 0; r < n; ++r) { rowoff[(size_t)r + 1] = rowoff[(size_t)r] + cnt[(size_t)r]; pmax[(size_t)r] = r ? std::max(pmax[(size_t)r - 1], ce[(size_t)r]) : ce[(size_t)r]; }
    const uint64_t n_entries = rowoff[(size_t)n], n_cols = colbase[(size_t)W];
    std::vector<uint32_t> E(n_entries + 1), Enm(bayes_mq ? n_entries + 1 : 1), depth(n_cols + 1);
    d.E = E.data(); d.Enm = bayes_mq ? Enm.data() : nullptr;
    for (int32_t r : clist) cons::step_walk(d, o, r);
    out.cols.assign(n_cols, sta_cons_col{ 0, 0, 0 });
    d.cols = out.cols.data(); d.depth = depth.data();
    const int kind = cons::col_kind(o);
    const cons::Probs &cp1 = cons::first_probs(o, tab);
    for (uint64_t c = 0; c < n_cols; ++c) {                      // 64-column spans, as the device's waves take them
        const int32_t c0 = (int32_t)(c & ~63ull), c1 = (int32_t)std::min<uint64_t>(c0 + 63, n_cols - 1);
        if (kind == 0) cons::step_col<0>(d, o, tab, cp1, tab.recall, tab.q2p, tab.mqual_pow_1m, (int64_t)c, c0, c1);
        else if (kind == 1) cons::step_col<1>(d, o, tab, cp1, tab.recall, tab.q2p, tab.mqual_pow_1m, (int64_t)c, c0, c1);
        else cons::step_col<2>(d, o, tab, cp1, tab.recall, tab.q2p, tab.mqual_pow_1m, (int64_t)c, c0, c1);
    }
    out.ins.assign(ins.begin() + 1, ins.end());
    out.info.n_cols = n_cols; out.info.n_entries = sum_depth; out.info.n_kept_reads = counters[0];
    if (p.want_pileup) {
        out.col_off.resize(n_cols + 1);
        out.col_off[0] = 0;
        for (uint64_t c = 0; c < n_cols; ++c) out.col_off[c + 1] = out.col_off[c] + depth[c];
        if (out.col_off[n_cols] != sum_depth) { err = "column depths do not add up to the reads' columns"; return -1; }
        out.seq.assign(sum_depth + 1, 0); out.qual.assign(sum_depth + 1, 0);
        d.col_off = out.col_off.data(); d.seq_chars = out.seq.data(); d.qual_chars = out.qual.data();
        for (uint64_t c = 0; c < n_cols; ++c) { const int32_t c0 = (int32_t)(c & ~63ull); cons::step_text(d, o, (int64_t)c, c0, (int32_t)std::min<uint64_t>(c0 + 63, n_cols - 1)); }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2 || strcmp(argv[1], "consensus")) { fprintf(stderr, "usage: cons_emul consensus [options] in.bam\n"); return 1; }
    return consensus_cli(argc - 1, argv + 1, emul);
}
