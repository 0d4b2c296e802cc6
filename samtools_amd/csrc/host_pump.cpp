// host_pump.cpp -- see host_pump.h
#include "host_pump.h"
#include <algorithm>
#include <climits>

namespace sta {

Pump::Pump(std::vector<std::unique_ptr<AlnReader>> &readers, const PumpConfig &cfg) : rd_(readers), cfg_(cfg)
{
    size_t n = rd_.size();
    pend_.resize(n); has_pend_.assign(n, 0); eof_.assign(n, 0); carry_.resize(n);
    last_pos_.assign(n, -1); last_tid_.assign(n, -1);
    next_id_.assign(n, 0); n_carry_staged_.assign(n, 0); n_fresh_paired_.assign(n, 0); dclip_.resize(n); onames_.resize(n);
    for (size_t f = 0; f < n; ++f) advance(f);
}

void Pump::advance(size_t f)
{
    has_pend_[f] = 0;
    if (eof_[f]) return;
    for (;;) {
        int r = rd_[f]->next(pend_[f]);
        if (r == 0) { eof_[f] = 1; return; }
        if (r < 0) { eof_[f] = 1; err_ = -1; errtxt_ = "error reading from input file"; return; }
        if (pend_[f].tid < 0) continue;              // unplaced reads never reach the engines
        if (pend_[f].tid >= cfg_.nref_limit) { eof_[f] = 1; err_ = -3; errtxt_ = "a record names a reference sequence that is not in the first input's header"; return; }
        if (!(pend_[f].flag & 4)) {
            if (pend_[f].tid < last_tid_[f] || (pend_[f].tid == last_tid_[f] && pend_[f].pos < last_pos_[f])) {
                eof_[f] = 1; err_ = -2; errtxt_ = "the input is not position sorted";
                return;
            }
            last_tid_[f] = pend_[f].tid; last_pos_[f] = pend_[f].pos;
        } else if (pend_[f].tid < last_tid_[f] || (pend_[f].tid == last_tid_[f] && pend_[f].pos < last_pos_[f])) {
            continue;                                  // out-of-order unmapped-flagged record: filtered anyway
        }
        has_pend_[f] = 1;
        return;
    }
}

int Pump::next_tid()
{
    int best = INT_MAX;
    for (size_t f = 0; f < rd_.size(); ++f) {
        if (has_pend_[f]) best = std::min(best, (int)pend_[f].tid);
        if (!carry_[f].empty()) best = std::min(best, (int)carry_[f].front().tid);
    }
    return best == INT_MAX ? -1 : best;
}

int64_t Pump::next_pos(int tid)
{
    int64_t best = INT64_MAX;
    for (size_t f = 0; f < rd_.size(); ++f)
        if (has_pend_[f] && pend_[f].tid == tid) best = std::min(best, pend_[f].pos);
    return best;
}

bool Pump::has_carry() const
{
    for (auto &c : carry_) if (!c.empty()) return true;
    return false;
}

int64_t Pump::carry_next_covered(int64_t cursor) const
{
    int64_t best = INT64_MAX;
    for (auto &c : carry_) for (auto &r : c) if (span_end(r) > cursor) best = std::min(best, std::max(r.pos, cursor));
    return best;
}

int64_t Pump::carry_max_end() const
{
    int64_t m = INT64_MIN;
    for (auto &c : carry_) for (auto &r : c) m = std::max(m, span_end(r));
    return m;
}

// a record leaves its file for the window: its running number, and -- depth -s -- its visit to the name hash (bam2depth.c:598-623)
void Pump::take(size_t f)
{
    Rec &r = pend_[f];
    r.id = next_id_[f]++;
    if (cfg_.tpl == PumpConfig::TPL_DEPTH) {
        r.clip = 0;
        if (cfg_.depth_filter.passes(r.flag, r.mapq, r.l_qseq, r.cigar.data(), r.cigar.size()))
            r.clip = dclip_[f].visit(r.qname.c_str(), r.flag, r.tid, r.endpos(), r.mtid, r.mpos);
    } else if (cfg_.tpl == PumpConfig::TPL_MPLP) r.name_h = qname_hash64(r.qname.data(), r.qname.size());
    carry_[f].push_back(std::move(r));
    advance(f);
}

int64_t Pump::fill(int tid, int64_t cb, int64_t ce_target, std::vector<std::vector<const Rec *>> &reads)
{
    size_t n = rd_.size();
    int64_t ce = ce_target;
    for (size_t f = 0; f < n; ++f) {
        n_carry_staged_[f] = carry_[f].size(); n_fresh_paired_[f] = 0;
        int64_t count = 0;
        while (has_pend_[f] && pend_[f].tid == tid && pend_[f].pos < ce) {
            int64_t p = pend_[f].pos;
            take(f);
            if (++count >= cfg_.max_reads && f == 0 && p >= cb) {
                // cut the window after this start position (all reads sharing it stay together)
                while (has_pend_[f] && pend_[f].tid == tid && pend_[f].pos == p) take(f);
                if (p + 1 > cb) ce = std::min(ce, p + 1);
                break;
            }
        }
    }
    if (cfg_.tpl == PumpConfig::TPL_MPLP) {
        // lookahead up to the first read that reaches bam_plp_push: see ChunkPump::fill_window (the iterators of different files advance
        // independently, bam_mplp_*: per file)
        for (size_t f = 0; f < n; ++f) {
            int64_t me = INT64_MIN;
            for (auto &r : carry_[f]) me = std::max(me, span_end(r));
            bool sure = false;
            if (!cfg_.pushed_on_device) for (auto &r : carry_[f]) if (r.pos >= ce && (!cfg_.pushed || cfg_.pushed(r))) { sure = true; break; }
            while (!sure && has_pend_[f] && pend_[f].tid == tid && pend_[f].pos < me) {
                if (!cfg_.pushed_on_device) sure = !cfg_.pushed || cfg_.pushed(pend_[f]);
                take(f);
            }
        }
    }
    reads.assign(n, {});
    for (size_t f = 0; f < n; ++f) {
        reads[f].reserve(carry_[f].size());
        for (auto &r : carry_[f]) reads[f].push_back(&r);
    }
    return ce;
}

int64_t Pump::fill_unstaged(int tid, int64_t cb, int64_t ce_target)
{
    const int64_t ce = fill(tid, cb, ce_target, last_);
    // a window the lane only passes over: its reads visit the overlap hash at once (the host's own verdict on who is pushed)
    if (cfg_.tpl == PumpConfig::TPL_MPLP) for (size_t f = 0; f < rd_.size(); ++f) pair_fresh(f, nullptr, 0);
    return ce;
}

// ---- mpileup's overlap hash (host_names.h); the twin of ChunkPump::pair_fresh ----
void Pump::pair_fresh(size_t f, const uint32_t *info, int64_t n_info)
{
    std::deque<Rec> &c = carry_[f];
    for (size_t i = n_carry_staged_[f] + (size_t)n_fresh_paired_[f]; i < c.size(); ++i) {
        Rec &x = c[i];
        bool pushed, dropped = false;
        if (info) {
            const uint32_t w = (int64_t)i < n_info ? info[i] : 0;
            pushed = (w & 1u) != 0;
            dropped = pushed && !(w & 2u) && x.rlen > 0;
        } else pushed = cfg_.pushed ? cfg_.pushed(x) : !(x.flag & 4);
        if (!pushed) continue;
        OverlapNames::Read r;
        r.h = x.name_h; r.qname = x.qname.data(); r.l_qname = (uint32_t)x.qname.size();
        r.flag = x.flag; r.tid = x.tid; r.mtid = x.mtid; r.l_qseq = x.l_qseq;
        r.pos = x.pos; r.end = x.end(); r.mpos = x.mpos; r.isize = x.isize; r.id = x.id;
        const int64_t holder = onames_[f].push(r, dropped);
        if (holder < 0) continue;
        x.mate_id = holder;
        auto it = std::lower_bound(c.begin(), c.end(), holder, [](const Rec &q, int64_t v) { return q.id < v; });
        if (it != c.end() && it->id == holder) { x.mate_end = it->end(); it->mate_end = x.end(); }      // the two stay staged together
    }
    n_fresh_paired_[f] = (int64_t)(c.size() - n_carry_staged_[f]);
}

void Pump::fill_mates(size_t f, int32_t *mate, int64_t n) const
{
    const std::deque<Rec> &c = carry_[f];
    for (int64_t i = 0; i < n && (size_t)i < c.size(); ++i) {
        mate[i] = -1;
        const int64_t id = c[(size_t)i].mate_id;
        if (id < 0) continue;
        auto it = std::lower_bound(c.begin(), c.end(), id, [](const Rec &q, int64_t v) { return q.id < v; });
        if (it != c.end() && it->id == id) mate[i] = (int32_t)(it - c.begin());
    }
}

void Pump::pair_staged(std::vector<StagedFile> &staged)
{
    if (cfg_.tpl != PumpConfig::TPL_MPLP) return;
    for (size_t f = 0; f < rd_.size() && f < staged.size(); ++f) {
        pair_fresh(f, nullptr, 0);
        StagedFile &s = staged[f];
        s.tpl = 2;
        s.mate.resize((size_t)s.n());
        fill_mates(f, s.mate.data(), s.n());
    }
}

void Pump::pair_from_info(size_t f, const uint32_t *info, int64_t n, int32_t *mate_out)
{
    for (int64_t i = 0; i < n; ++i) mate_out[i] = -1;
    if (cfg_.tpl != PumpConfig::TPL_MPLP || f >= rd_.size()) return;
    pair_fresh(f, info, n);
    fill_mates(f, mate_out, n);
}

int64_t Pump::fill_staged(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> &staged)
{
    int64_t ce = fill(tid, cb, ce_target, last_);
    staged.resize(rd_.size());
    for (size_t f = 0; f < rd_.size(); ++f) {
        XcolSpec xs; xs.rnext = cfg_.xs_rnext; xs.hdr = &rd_[f]->header(); xs.n_tags = cfg_.xs_n_tags; xs.empty = cfg_.xs_empty; xs.mods = cfg_.xs_mods;
        staged[f].clear();
        for (const Rec *r : last_[f]) staged[f].add(*r, cb, cfg_.rg_excl, (xs.n_cols() || xs.mods) ? &xs : nullptr);
        staged[f].finish();
        if (cfg_.tpl == PumpConfig::TPL_DEPTH) {
            staged[f].tpl = 1;
            staged[f].clip.resize(last_[f].size());
            for (size_t i = 0; i < last_[f].size(); ++i) staged[f].clip[i] = last_[f][i]->clip;
        }
    }
    return ce;
}

void Pump::drop(size_t f, const std::vector<char> &dropped)
{
    std::deque<Rec> keep;
    size_t i = 0;
    for (auto &r : carry_[f]) { if (!(i < dropped.size() && dropped[i])) keep.push_back(std::move(r)); ++i; }
    carry_[f].swap(keep);
}

void Pump::retire(int64_t ce)
{
    // see ChunkPump::retire: a record stays while its span reaches beyond the cut, or -- mpileup with overlap detection -- while its
    // partner in the overlap hash does
    const bool partners = cfg_.tpl == PumpConfig::TPL_MPLP;
    for (size_t f = 0; f < carry_.size(); ++f) {
        std::deque<Rec> keep;
        for (auto &r : carry_[f]) if (span_end(r) > ce || (partners && r.mate_end > ce)) keep.push_back(std::move(r));
        carry_[f].swap(keep);
        for (auto &r : carry_[f]) r.accepted = true;       // what stays was accepted by this window's -d replay
        n_carry_staged_[f] = carry_[f].size(); n_fresh_paired_[f] = 0;
    }
}

void Pump::drop_tid_carry() { for (auto &c : carry_) c.clear(); }

}  // namespace sta
