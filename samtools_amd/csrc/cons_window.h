// cons_window.h -- the steps that turn one staged window into consensus columns, written once over plain arrays: the HIP
// kernels of kernels_cons.hip run each step with one lane per read / per column on the device; the CPU harness of tests/cpu
// runs the same functions in loops (test infrastructure, see cons_core.h).
//
//   step A  (read)    filters of readaln2 + pileup_loop (bam_consensus.c:2083-2103, consensus_pileup.c:341-349), CIGAR shape,
//                     insertion runs -> max into ins[position]; Bayesian mode: nm_init -> nm[] (+ working qualities)
//   scan              colbase = exclusive sum of 1 + ins[1 ..]      (column index of (position, 0))
//   step B  (read)    first / last column of the read inside the window, number of entries
//   scans             rowoff = exclusive sum of entries; pmax = running max of last columns
//   step W  (read)    reads with indels / skips / pads only: the read's cursor walks its columns and writes one entry word
//                     (+ nm word) per column.  A read that is one aligned block between clips ("plain") stores nothing: what
//                     it shows in a column follows from its start (plain_entry)
//   step P  (position) colpos[column] = position, so that a column can place itself on a plain read
//   step C  (column)  alive reads = [lo, hi) by two binary searches, gathered in file order -> call, quality, depth
//   scan + step T     `-f pileup` only: base / quality characters of every column
#pragma once
#include "cons_core.h"
#include "../../include/samtools_amd.h"

namespace cons {

// what the column kernels need of a read, in one 32-byte record (two 16-byte loads instead of eight scattered ones)
struct alignas(16) Meta {
    int32_t cs, ce, pos, lead;   // first / last column in the window, leftmost coordinate, leading clip (plain reads)
    uint32_t pool8, lq;          // base_off8, l_qseq
    uint32_t bits;               // r_keep bits | mapq << 8 | (first quality byte == 0xff) << 16
    uint32_t pad;
};

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
    int32_t *r_last, *r_tail; uint32_t *r_keep;      // r_keep: 1 in the pileup, 2 reverse strand, 4 plain; r_tail of a plain read = its leading clip
    int32_t *colpos;             // [n_cols] position (relative to the origin) of every column
    int32_t *clist;              // reads that need the cursor walk
    Meta *meta;
    int32_t *cs, *ce, *pmax; uint32_t *cnt; uint64_t *rowoff;
    uint32_t *E, *Enm;
    sta_cons_col *cols; uint32_t *depth; uint64_t *col_off; char *seq_chars, *qual_chars;
    unsigned long long *counters;    // [0] kept reads, [1] bad CIGAR ops, [2] reads on clist, [3] sum of column depths
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

CONS_HD void md_of(const Win &w, int64_t r, const char *&md, int &md_len)
{
    md = nullptr; md_len = 0;
    if (w.n_xcols > 0) { const uint32_t a = w.xcol_off[r * w.n_xcols], b = w.xcol_off[r * w.n_xcols + 1]; md = w.xcol_text + a; md_len = (int)(b - a); }
}

// AMAX(ptr, value): atomic max on the device, plain max in the harness.  1 = the read is in the pileup, 0 = filtered,
// -1 = its CIGAR holds an operation outside MIDNSHP=X
// walk_all: every read gets stored entries (the iterator surface hands the entries themselves out), none counts as plain
template <class AMAX> CONS_HD int step_read_a(const Win &w, const Par &o, const Tables &t, int64_t r, AMAX amax, bool prepare_here, bool walk_all = false)
{
    w.r_keep[r] = 0; w.r_last[r] = w.pos[r] - 1; w.r_tail[r] = 0;
    const int fl = w.flag[r];
    if (o.incl_flags && !(fl & o.incl_flags)) return 0;
    if (o.excl_flags && (fl & o.excl_flags)) return 0;
    if (w.mapq[r] < o.min_mqual) return 0;
    if (fl & 4) return 0;
    const bool bayes_mq = o.mode != MODE_SIMPLE && o.use_mqual;
    if (bayes_mq && w.l_qseq[r] <= 0) return 0;                    // nm_init: "discard"
    ReadView v = view_of(w, r, false);
    const int32_t cb = w.col_beg, ce = w.col_end;
    uint32_t *ins = w.ins;
    Shape s = read_shape(v, [&](int32_t p, int32_t run) { if (p >= cb - 1 && p < ce) amax(&ins[p - (cb - 1)], (uint32_t)run); });
    if (s.bad_op) return -1;
    if (s.last < v.start) return 0;                                // no reference-consuming op: see DESIGN.md
    // plain: clips around exactly one M / = / X block
    int32_t n_al = 0, lead = 0; bool plain = true;
    for (int k = 0; k < v.n_cigar; ++k) {
        const int op = (int)(v.cigar[k] & 15u); const int32_t len = (int32_t)(v.cigar[k] >> 4);
        if (op == 0 || op == 7 || op == 8) { n_al++; if (len == 0) plain = false; }
        else if (op == 4) { if (!n_al) lead += len; }
        else if (op != 5) plain = false;
    }
    plain = plain && n_al == 1 && !walk_all;
    // the per-read preparation (only l_qseq <= 0 makes it drop a read, and that was tested above) runs here in the harness
    // and in its own LDS-staged kernel on the device
    if (bayes_mq && prepare_here) {
        const char *md = nullptr; int md_len = 0;
        md_of(w, r, md, md_len);
        int32_t *nm = w.nm + (size_t)w.base_off8[r] * 8;
        if (prepare_is_per_base(o)) {                              // the formulation the device runs one lane per 8 bases
            for (int i0 = 0; i0 < v.l_qseq; i0 += 8) prepare_granule(o, v, i0, nm + i0);
            read_prepare_md(o, v, md, md_len, nm);
        } else read_prepare(o, t, v, w.qual + (size_t)w.base_off8[r] * 8, md, md_len, nm);
    }
    w.r_keep[r] = 1u | ((fl & 16) ? 2u : 0u) | (plain ? 4u : 0u);
    w.r_last[r] = s.last; w.r_tail[r] = plain ? lead : s.tail_run;
    return 1;
}

// true: the read needs the cursor walk (goes on clist)
CONS_HD bool step_read_b(const Win &w, int64_t r, uint32_t &alive /* columns the read is in */)
{
    const int32_t W = w.col_end - w.col_beg;
    int32_t st = w.pos[r]; if (st < w.col_beg) st = w.col_beg; if (st > w.col_end) st = w.col_end;
    const int32_t cs = (int32_t)w.colbase[st - w.col_beg];
    int32_t ce = cs - 1;
    if (w.r_keep[r] && w.pos[r] < w.col_end && w.r_last[r] >= w.col_beg) {
        const int32_t last = w.r_last[r];
        const int32_t tail = (w.r_keep[r] & 4u) ? 0 : w.r_tail[r];
        ce = last < w.col_end ? (int32_t)w.colbase[last - w.col_beg] + tail : (int32_t)w.colbase[W] - 1;
    }
    alive = ce >= cs ? (uint32_t)(ce - cs + 1) : 0u;
    const bool walk = ce >= cs && !(w.r_keep[r] & 4u);
    w.cs[r] = cs; w.ce[r] = ce; w.cnt[r] = walk ? (uint32_t)(ce - cs + 1) : 0u;
    Meta m; m.cs = cs; m.ce = ce; m.pos = w.pos[r]; m.lead = w.r_tail[r]; m.pool8 = w.base_off8[r]; m.lq = (uint32_t)w.l_qseq[r]; m.pad = 0;
    const bool q0_absent = w.l_qseq[r] > 0 && w.qual_in[(size_t)w.base_off8[r] * 8] == 255;
    m.bits = w.r_keep[r] | ((uint32_t)w.mapq[r] << 8) | (q0_absent ? 1u << 16 : 0u);
    w.meta[r] = m;
    return walk;
}

CONS_HD void step_colpos(const Win &w, int64_t i)               // i = position index inside the window
{
    const uint64_t c0 = w.colbase[i], c1 = w.colbase[i + 1];
    for (uint64_t c = c0; c < c1; ++c) w.colpos[c] = w.col_beg + (int32_t)i;
}

// what a plain read shows in column ci: the base at its start-relative offset, or a pad opposite somebody else's insertion
// (Cursor::step reduced to a single aligned block: no reference skips, no own insertions, the read ends on nth 0)
CONS_HD uint32_t plain_entry(const Win &w, bool bayes_mq, bool working_qual, const Meta &m, int32_t ci, uint32_t &nmw)
{
    const int32_t p = w.colpos[ci], nth = ci - (int32_t)w.colbase[p - w.col_beg];
    const int32_t so = p - m.pos + m.lead, lq = (int32_t)m.lq;
    const size_t pool = (size_t)m.pool8 * 8;
    const uint8_t *q = (working_qual ? w.qual : w.qual_in) + pool;
    int base4, qual;
    if (so < lq) { qual = q[so]; base4 = seqi(w.seq + pool / 2, so); } else { qual = 0xff; base4 = 15; }
    if (nth > 0) {
        base4 = 16;
        if (so < lq) { const int q1 = so + 1 < lq ? q[so + 1] : 0; if (q1 < qual) qual = q1; } else qual = 0;
    }
    nmw = bayes_mq ? nm_word(w.nm + pool, lq, so) : 0u;
    return (uint32_t)base4 | ((uint32_t)(qual & 255) << 5) | ((m.bits & 2u) ? CONS_E_REV : 0u);
}

CONS_HD uint32_t entry_at(const Win &w, bool bayes_mq, bool working_qual, int64_t r, const Meta &m, int32_t ci, uint32_t &nmw)
{
    if (m.bits & 4u) return plain_entry(w, bayes_mq, working_qual, m, ci, nmw);
    const uint64_t at = w.rowoff[r] + (uint32_t)(ci - m.cs);
    nmw = bayes_mq ? w.Enm[at] : 0u;
    return w.E[at];
}

// so_words: the second word of an entry is the query offset (pileup_t.seq_offset) instead of the nm word
CONS_HD void step_walk(const Win &w, const Par &o, int64_t r, bool so_words = false)
{
    const uint32_t cnt = w.cnt[r];
    if (!cnt) return;
    const bool bayes_mq = o.mode != MODE_SIMPLE && o.use_mqual;
    ReadView v = view_of(w, r, bayes_mq && o.homopoly_on);
    const bool rev = (w.r_keep[r] & 2u) != 0;
    uint32_t *E = w.E + w.rowoff[r];
    uint32_t *En = bayes_mq || so_words ? w.Enm + w.rowoff[r] : nullptr;
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
                if (k < cnt) { E[k] = cur.entry(rev); if (En) En[k] = so_words ? (uint32_t)cur.seq_off : nm_word(nm, v.l_qseq, cur.seq_off); }
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

// One wave takes 64 consecutive columns [c0, c1].  The reads that can be alive in any of them are [lo, hi) (two binary
// searches on wave-uniform keys), and every lane walks that same range: the per-read record is then a wave-uniform (scalar)
// load and the alive test a compare, instead of every lane searching and gathering for itself.  Each lane still adds its own
// alive reads in file order.  CONS_UNIFORM tells the device compiler the value is the same in every lane.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define CONS_UNIFORM(v) __builtin_amdgcn_readfirstlane(v)
#else
#define CONS_UNIFORM(v) (v)
#endif
struct Span { int32_t lo, hi; };
CONS_HD Span span_of(const Win &w, int32_t c0, int32_t c1)
{
    Span s;
    s.lo = CONS_UNIFORM((int32_t)lower_ge(w.pmax, w.n_reads, CONS_UNIFORM(c0)));
    s.hi = CONS_UNIFORM((int32_t)upper_le(w.cs, w.n_reads, CONS_UNIFORM(c1)));
    return s;
}

// KIND: 0 frequency caller, 1 one Bayesian parameter set, 2 both (mixed mode) -- separate instantiations keep the
// accumulators of the modes that are not running out of the registers
// cp1 (+ cp2 in the mixed mode), q2p, mqpow: the parameter sets and the two per-entry lookup tables of t, wherever the caller
// keeps them (LDS copies on the device); t itself is only read once per column (fast_exp).  [c0, c1]: the wave's columns.
template <int KIND> CONS_HD void step_col(const Win &w, const Par &o, const Tables &t, const Probs &cp1, const Probs &cp2, const double *q2p, const double *mqpow,
                                          int64_t c, int32_t c0, int32_t c1)
{
    const int32_t ci = (int32_t)c;
    const Span sp = span_of(w, c0, c1);
    const bool bayes_mq = o.mode != MODE_SIMPLE && o.use_mqual, workq = bayes_mq && o.homopoly_on;
    sta_cons_col out; out.depth = 0; out.base = 'N'; out.qual = 0;
    if (KIND == 0) {
        SimpleAcc acc; acc.init();
        int32_t td = 0;
        for (int32_t r = sp.lo; r < sp.hi; ++r) {
            const Meta m = w.meta[r];
            if (m.cs > ci || m.ce < ci) continue;
            ++td;
            uint32_t nmw; acc.add(o, entry_at(w, false, false, r, m, ci, nmw));
        }
        out.depth = td;
        if (td) { int32_t q; out.base = acc.finish(o, q); out.qual = q; }
    } else {
        int32_t td = 0;
        for (int32_t r = sp.lo; r < sp.hi; ++r) { const Meta m = w.meta[r]; td += m.cs <= ci && m.ce >= ci; }
        out.depth = td;
        if (td) {
            const bool mixed = KIND == 2;
            Gap5Acc a1, a2; a1.init(); if (mixed) a2.init();
            for (int32_t r = sp.lo; r < sp.hi; ++r) {
                const Meta m = w.meta[r];
                if (m.cs > ci || m.ce < ci) continue;
                uint32_t nmw;
                const uint32_t e = entry_at(w, bayes_mq, workq, r, m, ci, nmw);
                const int mapq = (int)((m.bits >> 8) & 255u); const bool q0a = (m.bits >> 16) & 1u;
                a1.add(o, q2p, mqpow, cp1, e, nmw, mapq, q0a, td);
                if (mixed) a2.add(o, q2p, mqpow, cp2, e, nmw, mapq, q0a, td);
            }
            Call c1r; a1.finish(t, cp1, c1r);
            if (mixed) { Call c2r; a2.finish(t, cp2, c2r); c1r = mix_calls(c1r, c2r); }
            int32_t q; out.base = final_call(o, c1r, q); out.qual = q;
        }
    }
    w.depth[c] = (uint32_t)out.depth;
    w.cols[c] = out;
}

CONS_HD int col_kind(const Par &o) { return o.mode == MODE_SIMPLE ? 0 : o.mode == MODE_MIXED ? 2 : 1; }
// first (or only) parameter set of the mode; the mixed mode's second one is always the recall set
CONS_HD const Probs &first_probs(const Par &o, const Tables &t) { return o.mode == MODE_PRECISE || o.mode == MODE_MIXED ? t.precise : t.recall; }

CONS_HD void step_text(const Win &w, const Par &o, int64_t c, int32_t c0, int32_t c1)
{
    const int32_t ci = (int32_t)c;
    const Span sp = span_of(w, c0, c1);
    if (!w.depth[c]) return;
    const bool workq = o.mode != MODE_SIMPLE && o.use_mqual && o.homopoly_on;
    uint64_t at = w.col_off[c];
    for (int32_t r = sp.lo; r < sp.hi; ++r) {
        const Meta m = w.meta[r];
        if (m.cs > ci || m.ce < ci) continue;
        uint32_t nmw;
        const uint32_t e = entry_at(w, false, workq, r, m, ci, nmw);
        const int b4 = CONS_E_BASE4(e);
        char ch = (e & CONS_E_SKIPCOL) ? '.' : b4 >= 16 ? '*' : "NACMGRSVTWYHKDBN"[b4];
        if (e & CONS_E_REV) ch = ch == '*' ? '#' : (ch >= 'A' && ch <= 'Z' ? (char)(ch + 32) : ch);
        const int q = CONS_E_QUAL(e);
        w.seq_chars[at] = ch; w.qual_chars[at] = (char)((q < 93 ? q : 93) + '!');
        ++at;
    }
}

}  // namespace cons
