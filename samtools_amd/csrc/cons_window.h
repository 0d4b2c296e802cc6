// cons_window.h -- the steps that turn one staged window into consensus columns, written once over plain arrays: the HIP
// kernels of kernels_cons.hip run each step with one lane per read / per column on the device; the CPU harness of tests/cpu
// runs the same functions in loops (test infrastructure, see cons_core.h).
//
//   step A  (read)    filters of readaln2 + pileup_loop (bam_consensus.c:2083-2103, consensus_pileup.c:341-349), CIGAR shape,
//                     insertion runs -> max into ins[position]; Bayesian mode: nm_init -> nm[] (+ working qualities)
//   scan              colbase = exclusive sum of 1 + ins[1 ..]      (column index of (position, 0))
//   step B  (read)    first / last column of the read inside the window, number of entries
//   scans             rowoff = exclusive sum of entries; pmax = running max of last columns
//   step W  (read)    the read's cursor walks its columns and writes one entry word (+ nm word) per column
//   step C  (column)  alive reads = [lo, hi) by two binary searches, gathered in file order -> call, quality, depth
//   scan + step T     `-f pileup` only: base / quality characters of every column
#pragma once
#include "cons_core.h"
#include "../../include/samtools_amd.h"

namespace cons {

struct Win {
    int64_t n_reads;
    const int32_t *pos; const uint16_t *flag; const uint8_t *mapq; const int32_t *l_qseq;
    const uint32_t *cig_off, *base_off8, *cigar;
    const uint8_t *seq, *qual_in;
    int32_t n_xcols; const uint32_t *xcol_off; const char *xcol_text;
    int32_t col_beg, col_end;
    // workspace
    uint8_t *qual;               // working qualities (a copy when homopolymer fixing rewrites them, else qual_in)
    int32_t *nm;                 // one word per staged base (Bayesian mode with mapping qualities), else null
    uint32_t *ins;               // [W + 1] inserted columns after positions col_beg - 1 (look-back, see step_walk) .. col_end - 1
    uint64_t *colbase;           // [W + 1]
    int32_t *r_last, *r_tail; uint32_t *r_keep;
    int32_t *cs, *ce, *pmax; uint32_t *cnt; uint64_t *rowoff;
    uint32_t *E, *Enm;
    sta_cons_col *cols; uint32_t *depth; uint64_t *col_off; char *seq_chars, *qual_chars;
    unsigned long long *counters;    // [0] kept reads, [1] bad CIGAR ops
};

CONS_HD ReadView view_of(const Win &w, int64_t r, bool working_qual)
{
    ReadView v;
    v.start = w.pos[r]; v.l_qseq = w.l_qseq[r];
    v.n_cigar = (int32_t)(w.cig_off[r + 1] - w.cig_off[r]);
    v.cigar = w.cigar + w.cig_off[r];
    v.seq = w.seq + (size_t)w.base_off8[r] * 4;
    v.qual = (working_qual ? w.qual : w.qual_in) + (size_t)w.base_off8[r] * 8;
    return v;
}

// AMAX(ptr, value): atomic max on the device, plain max in the harness; ADD likewise
template <class AMAX, class ADD> CONS_HD void step_read_a(const Win &w, const Par &o, const Tables &t, int64_t r, AMAX amax, ADD add)
{
    w.r_keep[r] = 0; w.r_last[r] = w.pos[r] - 1; w.r_tail[r] = 0;
    const int fl = w.flag[r];
    if (o.incl_flags && !(fl & o.incl_flags)) return;
    if (o.excl_flags && (fl & o.excl_flags)) return;
    if (w.mapq[r] < o.min_mqual) return;
    if (fl & 4) return;
    const bool bayes_mq = o.mode != MODE_SIMPLE && o.use_mqual;
    if (bayes_mq && w.l_qseq[r] <= 0) return;                      // nm_init: "discard"
    ReadView v = view_of(w, r, false);
    const int32_t cb = w.col_beg, ce = w.col_end;
    uint32_t *ins = w.ins;
    Shape s = read_shape(v, [&](int32_t p, int32_t run) { if (p >= cb - 1 && p < ce) amax(&ins[p - (cb - 1)], (uint32_t)run); });
    if (s.bad_op) { add(&w.counters[1], 1ull); return; }
    if (s.last < v.start) return;                                  // no reference-consuming op: see DESIGN.md
    if (bayes_mq) {
        uint8_t *wq = w.qual + (size_t)w.base_off8[r] * 8;
        const char *md = nullptr; int md_len = 0;
        if (w.n_xcols > 0) { const uint32_t a = w.xcol_off[r * w.n_xcols], b = w.xcol_off[r * w.n_xcols + 1]; md = w.xcol_text + a; md_len = (int)(b - a); }
        if (!read_prepare(o, t, v, wq, md, md_len, w.nm + (size_t)w.base_off8[r] * 8)) return;
    }
    w.r_keep[r] = 1 | ((fl & 16) ? 2u : 0u);
    w.r_last[r] = s.last; w.r_tail[r] = s.tail_run;
    add(&w.counters[0], 1ull);
}

CONS_HD void step_read_b(const Win &w, int64_t r)
{
    const int32_t W = w.col_end - w.col_beg;
    int32_t st = w.pos[r]; if (st < w.col_beg) st = w.col_beg; if (st > w.col_end) st = w.col_end;
    const int32_t cs = (int32_t)w.colbase[st - w.col_beg];
    int32_t ce = cs - 1;
    if (w.r_keep[r] && w.pos[r] < w.col_end && w.r_last[r] >= w.col_beg) {
        const int32_t last = w.r_last[r];
        ce = last < w.col_end ? (int32_t)w.colbase[last - w.col_beg] + w.r_tail[r] : (int32_t)w.colbase[W] - 1;
    }
    w.cs[r] = cs; w.ce[r] = ce; w.cnt[r] = (uint32_t)(ce - cs + 1);
}

CONS_HD void step_walk(const Win &w, const Par &o, int64_t r)
{
    const uint32_t cnt = w.cnt[r];
    if (!cnt) return;
    const bool bayes_mq = o.mode != MODE_SIMPLE && o.use_mqual;
    ReadView v = view_of(w, r, bayes_mq && o.homopoly_on);
    const bool rev = (w.r_keep[r] & 2u) != 0;
    uint32_t *E = w.E + w.rowoff[r];
    uint32_t *En = bayes_mq ? w.Enm + w.rowoff[r] : nullptr;
    const int32_t *nm = bayes_mq ? w.nm + (size_t)w.base_off8[r] * 8 : nullptr;
    const int32_t cs = w.cs[r];
    Cursor cur; cur.init(v.start);
    uint32_t k = 0;
    bool done = false;
    // Columns before the window are walked for the state they leave behind.  That state depends on the read's own insertions
    // (known) and, in one place, on other reads': a pad column at the last position of a reference skip clears the flag that
    // would mark the first base after the skip (get_next_base's eof / ref_skip handling).  It can only reach into the window
    // from the position just before it, so the insertion count of col_beg - 1 is computed too (the driver stages the reads
    // alive there); further back only the read's own insertions are iterated.
    const int32_t col_lo = w.col_beg - 1;
    for (int32_t p = v.start; !done && p < w.col_end; ++p) {
        const bool inw = p >= w.col_beg, known = p >= col_lo;
        const int32_t n_ins = known ? (int32_t)w.ins[p - col_lo] : 0;
        for (int32_t n = 0;; ++n) {
            int32_t ins = 0;
            if (cur.step(v, p, n, ins) <= 0) { done = true; break; }
            if (inw) {
                k = (uint32_t)((int32_t)w.colbase[p - w.col_beg] + n - cs);
                if (k < cnt) { E[k] = cur.entry(rev); if (En) En[k] = nm_word(nm, v.l_qseq, cur.seq_off); }
                k++;
            }
            if (cur.eof == 1) { done = true; break; }
            if (known ? n >= n_ins : ins <= 0) break;
        }
    }
    for (; k < cnt; ++k) { E[k] = CONS_E_REFSKIP | CONS_E_SKIPCOL; if (En) En[k] = 0; }   // only after a malformed CIGAR
}

// number of reads r with key[r] <= c (key ascending)
CONS_HD int64_t upper_le(const int32_t *key, int64_t n, int32_t c)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (key[m] <= c) lo = m + 1; else hi = m; }
    return lo;
}
// first r with key[r] >= c (key ascending)
CONS_HD int64_t lower_ge(const int32_t *key, int64_t n, int32_t c)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (key[m] < c) lo = m + 1; else hi = m; }
    return lo;
}

CONS_HD void step_col(const Win &w, const Par &o, const Tables &t, int64_t c)
{
    const int32_t ci = (int32_t)c;
    const int64_t hi = upper_le(w.cs, w.n_reads, ci), lo = lower_ge(w.pmax, w.n_reads, ci);
    int32_t td = 0;
    for (int64_t r = lo; r < hi; ++r) td += w.ce[r] >= ci;
    sta_cons_col out; out.depth = td; out.base = 'N'; out.qual = 0;
    w.depth[c] = (uint32_t)td;
    if (td == 0) { w.cols[c] = out; return; }
    if (o.mode == MODE_SIMPLE) {
        SimpleAcc acc; acc.init();
        for (int64_t r = lo; r < hi; ++r) if (w.ce[r] >= ci) acc.add(o, w.E[w.rowoff[r] + (uint32_t)(ci - w.cs[r])]);
        int32_t q; out.base = acc.finish(o, q); out.qual = q;
    } else {
        const bool mixed = o.mode == MODE_MIXED;
        const Probs &cp1 = o.mode == MODE_PRECISE || mixed ? t.precise : t.recall;
        Gap5Acc a1, a2; a1.init(); a2.init();
        for (int64_t r = lo; r < hi; ++r) {
            if (w.ce[r] < ci) continue;
            const uint64_t at = w.rowoff[r] + (uint32_t)(ci - w.cs[r]);
            const uint32_t e = w.E[at], nmw = o.use_mqual ? w.Enm[at] : 0u;
            const int q0 = w.l_qseq[r] > 0 ? w.qual_in[(size_t)w.base_off8[r] * 8] : 0;
            a1.add(o, t, cp1, e, nmw, w.mapq[r], q0, td);
            if (mixed) a2.add(o, t, t.recall, e, nmw, w.mapq[r], q0, td);
        }
        Call c1; a1.finish(t, cp1, c1);
        if (mixed) { Call c2; a2.finish(t, t.recall, c2); c1 = mix_calls(c1, c2); }
        int32_t q; out.base = final_call(o, c1, q); out.qual = q;
    }
    w.cols[c] = out;
}

CONS_HD void step_text(const Win &w, int64_t c)
{
    const int32_t ci = (int32_t)c;
    if (!w.depth[c]) return;
    const int64_t hi = upper_le(w.cs, w.n_reads, ci), lo = lower_ge(w.pmax, w.n_reads, ci);
    uint64_t at = w.col_off[c];
    for (int64_t r = lo; r < hi; ++r) {
        if (w.ce[r] < ci) continue;
        const uint32_t e = w.E[w.rowoff[r] + (uint32_t)(ci - w.cs[r])];
        const int b4 = CONS_E_BASE4(e);
        char ch = (e & CONS_E_SKIPCOL) ? '.' : b4 >= 16 ? '*' : "NACMGRSVTWYHKDBN"[b4];
        if (e & CONS_E_REV) ch = ch == '*' ? '#' : (ch >= 'A' && ch <= 'Z' ? (char)(ch + 32) : ch);
        const int q = CONS_E_QUAL(e);
        w.seq_chars[at] = ch; w.qual_chars[at] = (char)((q < 93 ? q : 93) + '!');
        ++at;
    }
}

}  // namespace cons
