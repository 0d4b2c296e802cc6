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

int64_t Pump::fill(int tid, int64_t cb, int64_t ce_target, std::vector<std::vector<const Rec *>> &reads)
{
    size_t n = rd_.size();
    int64_t ce = ce_target;
    for (size_t f = 0; f < n; ++f) {
        int64_t count = 0;
        while (has_pend_[f] && pend_[f].tid == tid && pend_[f].pos < ce) {
            int64_t p = pend_[f].pos;
            carry_[f].push_back(std::move(pend_[f]));
            advance(f);
            if (++count >= cfg_.max_reads && f == 0 && p >= cb) {
                // cut the window after this start position (all reads sharing it stay together)
                while (has_pend_[f] && pend_[f].tid == tid && pend_[f].pos == p) { carry_[f].push_back(std::move(pend_[f])); advance(f); }
                if (p + 1 > cb) ce = std::min(ce, p + 1);
                break;
            }
        }
    }
    if (cfg_.surely_pushed) {
        for (size_t f = 0; f < n; ++f) {
            // the iterators of different files advance independently (bam_mplp_*): lookahead is per file
            int64_t me = INT64_MIN;
            for (auto &r : carry_[f]) me = std::max(me, span_end(r));
            bool sure = false;
            for (auto &r : carry_[f]) if (r.pos >= ce && cfg_.surely_pushed(r)) { sure = true; break; }   // cut windows
            while (!sure && has_pend_[f] && pend_[f].tid == tid && pend_[f].pos < me) {
                sure = cfg_.surely_pushed(pend_[f]);
                carry_[f].push_back(std::move(pend_[f]));
                advance(f);
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

int64_t Pump::fill_staged(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> &staged)
{
    int64_t ce = fill(tid, cb, ce_target, last_);
    staged.resize(rd_.size());
    for (size_t f = 0; f < rd_.size(); ++f) {
        XcolSpec xs; xs.rnext = cfg_.xs_rnext; xs.hdr = &rd_[f]->header(); xs.n_tags = cfg_.xs_n_tags; xs.empty = cfg_.xs_empty; xs.mods = cfg_.xs_mods;
        staged[f].clear();
        for (const Rec *r : last_[f]) staged[f].add(*r, cb, cfg_.rg_excl, (xs.n_cols() || xs.mods) ? &xs : nullptr);
        staged[f].finish();
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
    for (auto &c : carry_) {
        std::vector<const Rec *> stay;
        if (cfg_.keep_mates)
            for (auto &r : c) if (span_end(r) > ce && (r.flag & 1) && ((r.flag & 2) || !cfg_.mates_proper_only) && !(r.flag & 8)) stay.push_back(&r);
        auto mate_stays = [&](const Rec &r) {
            if (!(r.flag & 1) || (!(r.flag & 2) && cfg_.mates_proper_only) || (r.flag & 8) || r.mtid != r.tid) return false;
            auto lo = std::lower_bound(stay.begin(), stay.end(), r.mpos, [](const Rec *s, int64_t p) { return s->pos < p; });   // carry is position sorted
            for (; lo != stay.end() && (*lo)->pos == r.mpos; ++lo) if (*lo != &r && (*lo)->qname == r.qname) return true;
            return false;
        };
        // A record that stays ONLY for its mate's sake holds nothing any more once a read beyond its end was pushed before that mate:
        // bam_plp_next frees it (overlap_remove takes the entry of its name along) as soon as max_pos has passed its end.  Carried on, it
        // would look to the next window's replay (k_name_groups) like the holder of the template's entry -- the record that freed it is
        // not staged there.  Seen with three records of one template: a supplementary alignment upstream of the primary pair, a window
        // cut between them (scripts/hunt5.py, round 5).
        // "Before that mate" means before the NEXT record of its template that is pushed, which need not be the one at its mate position: a
        // supplementary alignment (mate position = the second primary) is still in the buffer when the FIRST primary arrives right behind
        // its end -- that one finds the entry and deletes it, and the second primary finds nothing.  Dropped, the primaries would pair up
        // in the next window's replay (scripts/hunt6.py seed 29, round 5).
        auto freed_before_mate = [&](const Rec &r) {
            if (!cfg_.surely_pushed) return false;
            const int64_t e = span_end(r);
            bool behind = false;
            for (auto &q : c) {
                if (&q == &r) { behind = true; continue; }
                if (!behind) continue;
                if (q.pos > r.mpos) break;            // (a record AT the mate position in front of the mate in the file frees it too)
                if (!cfg_.surely_pushed(q)) continue;
                if (q.qname == r.qname) return false;
                if (q.pos > e) return true;
            }
            return false;
        };
        // Where the host cannot tell who is pushed (-l, -G, -C, --min-read-len: surely_pushed says no), the record stays and so does
        // every record that starts between its end and its mate: the replay then sees, from their RI_PUSHED, whether one of them freed it.
        struct Ctx { int64_t pos, end, mpos; const std::string *qname; };      // a record kept for its mate only
        std::vector<Ctx> ctx;
        std::vector<char> gone(c.size(), 0);       // decided before anything moves: `stay` points into c
        size_t i = 0;
        for (auto &r : c) {
            bool keep_r = span_end(r) > ce;
            if (!keep_r && !stay.empty() && mate_stays(r) && !freed_before_mate(r)) { keep_r = true; ctx.push_back(Ctx{ r.pos, span_end(r), r.mpos, &r.qname }); }
            gone[i++] = !keep_r;
        }
        if (cfg_.keep_mates) {
            // (1) A record whose span ends at the cut is still in the reference's buffer -- and its template's entry in the hash -- while no
            // pushed read has started beyond its end: the next window's first read still meets it (a supplementary alignment right in
            // front of its primaries: the first primary finds its entry, deletes it, and the pair is never resolved).
            // (Only while the contig has reads to come: at its end the reference flushes its buffer, and a record kept here for ever
            // would keep the window loop going for ever.)
            int64_t max_start = INT64_MIN;
            const bool more = !c.empty() && next_pos(c.front().tid) != INT64_MAX;
            if (more) for (auto &r : c) if (r.pos < ce && (!cfg_.surely_pushed || cfg_.pushed_unknown || cfg_.surely_pushed(r))) max_start = std::max(max_start, r.pos);
            i = 0;
            if (more && max_start != INT64_MIN) for (auto &r : c) { if (gone[i] && span_end(r) >= max_start) gone[i] = 0; ++i; }
            // (2) Templates with more than two records (one of them secondary / supplementary): what a record that stays finds in the hash
            // depends on every record of its template the window has seen -- a primary that consumed the supplementary's entry three
            // windows ago must not insert its own when the windows are replayed.  They all stay while one of them does, each ended one
            // with the records up to the template's last one as context (scripts/hunt6.py seed 29, round 5).
            std::vector<const Rec *> multi;
            for (auto &r : c) if (r.flag & 0x900) multi.push_back(&r);
            if (!multi.empty()) {
                std::vector<Ctx> tpl;                      // (first pos, -, last pos, name) of a template with a staying record
                i = 0;
                for (auto &r : c) {
                    if (!gone[i++]) {
                        bool is_multi = false;
                        for (const Rec *m : multi) if (m->qname == r.qname) { is_multi = true; break; }
                        if (is_multi) {
                            bool known = false;
                            for (auto &t : tpl) if (*t.qname == r.qname) { known = true; break; }
                            if (!known) tpl.push_back(Ctx{ INT64_MAX, 0, INT64_MIN, &r.qname });
                        }
                    }
                }
                if (!tpl.empty()) {
                    for (auto &r : c) for (auto &t : tpl) if (r.qname == *t.qname) { t.pos = std::min(t.pos, r.pos); t.mpos = std::max(t.mpos, r.pos); }
                    i = 0;
                    for (auto &r : c) {
                        if (gone[i]) for (auto &t : tpl) if (r.qname == *t.qname) { gone[i] = 0; ctx.push_back(Ctx{ r.pos, span_end(r), t.mpos, &r.qname }); break; }
                        ++i;
                    }
                }
            }
        }
        if (!ctx.empty()) {
            i = 0;
            for (auto &r : c) { if (gone[i]) for (auto &iv : ctx) if (r.pos > iv.end && r.pos <= iv.mpos) { gone[i] = 0; break; } ++i; }
            // ... and the other records of its template (one that starts inside its span is no context record by position)
            i = 0;
            for (auto &r : c) { if (gone[i]) for (auto &iv : ctx) if (r.pos >= iv.pos && r.pos <= iv.mpos && r.qname == *iv.qname) { gone[i] = 0; break; } ++i; }
        }
        std::deque<Rec> keep;
        i = 0;
        for (auto &r : c) { if (!gone[i++]) keep.push_back(std::move(r)); }
        c.swap(keep);
        for (auto &r : c) r.accepted = true;       // what stays was accepted by this window's -d replay
    }
}

void Pump::drop_tid_carry() { for (auto &c : carry_) c.clear(); }

}  // namespace sta
