// tests/cpu/cons_mock_engine.cpp -- TEST INFRASTRUCTURE: the five engine entry points that sta_pileup_loop()
// (samtools_amd/csrc/cons_loop_api.cpp) calls, implemented on the CPU with the shared step functions of cons_window.h, so that the
// not-gpu suite can run the pileup_loop() host logic (record batches, window cuts, look-back reads, seq_init / seq_free
// bookkeeping) behind the external C client tests/cabi/cons_client.c.  Never linked into libsamtools_amd.so.
#include "../../include/samtools_amd.h"
#include "../../samtools_amd/csrc/cons_window.h"
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

struct sta_engine {
    std::string err;
    // deep copy of the staged file-0 arrays
    std::vector<int32_t> pos, l_qseq; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, seq, qual;
    std::vector<uint32_t> cig_off, base_off8, cigar;
    int32_t col_beg = 0, col_end = 0; bool staged = false;
    // results
    std::vector<uint32_t> ins, E, Eso, keep, cnt; std::vector<int32_t> cs, ce; std::vector<uint64_t> rowoff;
};

extern "C" {

int sta_device_count(void) { return 1; }
int sta_engine_create(sta_engine **out, int, void *) { *out = new sta_engine; return STA_OK; }
void sta_engine_destroy(sta_engine *e) { delete e; }
const char *sta_last_error(const sta_engine *e) { return e ? e->err.c_str() : ""; }

int sta_stage_window(sta_engine *e, const sta_window *w)
{
    if (!e || !w || w->n_files != 1) return STA_ERR_ARG;
    const sta_reads &f = w->files[0];
    const size_t n = (size_t)f.n_reads;
    e->pos.assign(f.pos, f.pos + n); e->l_qseq.assign(f.l_qseq, f.l_qseq + n); e->flag.assign(f.flag, f.flag + n); e->mapq.assign(f.mapq, f.mapq + n);
    e->cig_off.assign(f.cig_off, f.cig_off + n + 1); e->base_off8.assign(f.base_off8, f.base_off8 + n);
    e->cigar.assign(f.cigar, f.cigar + f.n_cigar_total);
    e->qual.assign(f.qual, f.qual + f.n_bases_total); e->seq.assign(f.seq, f.seq + f.n_bases_total / 2);
    e->col_beg = w->col_beg; e->col_end = w->col_end; e->staged = true;
    return STA_OK;
}

// the iterator half: every read walked (walk_all), second entry word = seq_offset
int sta_cons_entries_run(sta_engine *e, sta_cons_info *info)
{
    if (!e || !e->staged) return STA_ERR_ARG;
    cons::Par o; memset(&o, 0, sizeof o); o.mode = cons::MODE_SIMPLE;
    static cons::Tables tab;
    const int64_t n = (int64_t)e->pos.size();
    const int32_t W = e->col_end - e->col_beg;
    cons::Win d; memset(&d, 0, sizeof d);
    d.n_reads = n; d.pos = e->pos.data(); d.flag = e->flag.data(); d.mapq = e->mapq.data(); d.l_qseq = e->l_qseq.data();
    d.cig_off = e->cig_off.data(); d.base_off8 = e->base_off8.data(); d.cigar = e->cigar.data(); d.seq = e->seq.data(); d.qual_in = e->qual.data();
    d.qual = e->qual.data(); d.col_beg = e->col_beg; d.col_end = e->col_end;
    e->ins.assign((size_t)W + 1, 0); e->keep.assign((size_t)n + 1, 0); e->cnt.assign((size_t)n + 1, 0);
    e->cs.assign((size_t)n + 1, 0); e->ce.assign((size_t)n + 1, 0); e->rowoff.assign((size_t)n + 1, 0);
    std::vector<uint64_t> colbase((size_t)W + 1); std::vector<int32_t> r_last((size_t)n + 1), r_tail((size_t)n + 1), pmax((size_t)n + 1);
    std::vector<cons::Meta> meta((size_t)n + 1);
    unsigned long long counters[4] = { 0, 0, 0, 0 };
    d.ins = e->ins.data(); d.colbase = colbase.data(); d.r_last = r_last.data(); d.r_tail = r_tail.data(); d.r_keep = e->keep.data();
    d.cs = e->cs.data(); d.ce = e->ce.data(); d.pmax = pmax.data(); d.cnt = e->cnt.data(); d.rowoff = e->rowoff.data(); d.meta = meta.data(); d.counters = counters;
    auto amax = [](uint32_t *q, uint32_t v) { if (*q < v) *q = v; };
    for (int64_t r = 0; r < n; ++r) { const int code = cons::step_read_a(d, o, tab, r, amax, false, true); if (code < 0) { e->err = "a CIGAR holds an operation outside MIDNSHP=X"; return STA_ERR_ARG; } counters[0] += code > 0; }
    colbase[0] = 0;
    for (int32_t i = 0; i < W; ++i) colbase[(size_t)i + 1] = colbase[(size_t)i] + 1 + e->ins[(size_t)i + 1];
    std::vector<int32_t> clist;
    uint64_t sum_depth = 0;
    for (int64_t r = 0; r < n; ++r) { uint32_t alive; if (cons::step_read_b(d, r, alive)) clist.push_back((int32_t)r); sum_depth += alive; }
    for (int64_t r = 0; r < n; ++r) e->rowoff[(size_t)r + 1] = e->rowoff[(size_t)r] + e->cnt[(size_t)r];
    e->E.assign((size_t)e->rowoff[(size_t)n] + 1, 0); e->Eso.assign((size_t)e->rowoff[(size_t)n] + 1, 0);
    d.E = e->E.data(); d.Enm = e->Eso.data();
    for (int32_t r : clist) cons::step_walk(d, o, r, true);
    if (info) { info->n_cols = colbase[(size_t)W]; info->n_entries = sum_depth; info->n_kept_reads = counters[0]; }
    return STA_OK;
}

int sta_fetch_cons_entries(sta_engine *e, int32_t *ins, int32_t *first_col, int32_t *last_col, uint64_t *entry_off, uint32_t *entries, uint32_t *seq_offs)
{
    const size_t n = e->pos.size(), W = (size_t)(e->col_end - e->col_beg), ne = (size_t)e->rowoff[n];
    if (ins) for (size_t i = 0; i < W; ++i) ins[i] = (int32_t)e->ins[i + 1];
    if (first_col) memcpy(first_col, e->cs.data(), n * 4);
    if (last_col) memcpy(last_col, e->ce.data(), n * 4);
    if (entry_off) memcpy(entry_off, e->rowoff.data(), (n + 1) * 8);
    if (entries) memcpy(entries, e->E.data(), ne * 4);
    if (seq_offs) memcpy(seq_offs, e->Eso.data(), ne * 4);
    return STA_OK;
}

}
