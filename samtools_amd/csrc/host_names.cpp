// host_names.cpp -- see host_names.h
#include "host_names.h"
#include <climits>

namespace sta {

OverlapNames::Entry *OverlapNames::find(const Read &r)
{
    if (!n_entries_) return nullptr;
    const size_t mask = tab_.size() - 1;
    for (size_t s = (size_t)r.h & mask;; s = (s + 1) & mask) {
        Entry &e = tab_[s];
        if (!e.used && !e.tomb) return nullptr;
        if (e.used && e.h == r.h && e.name_len == r.l_qname && !memcmp(arena_.data() + e.name_off, r.qname, r.l_qname)) return &e;
    }
}

void OverlapNames::rebuild(size_t cap)
{
    std::vector<Entry> old;
    old.swap(tab_);
    std::vector<char> names;
    names.swap(arena_);
    tab_.assign(cap, Entry{});
    n_tomb_ = 0;
    const size_t mask = cap - 1;
    for (const Entry &o : old) {
        if (!o.used) continue;
        size_t s = (size_t)o.h & mask;
        while (tab_[s].used) s = (s + 1) & mask;
        tab_[s] = o;
        tab_[s].name_off = (uint32_t)arena_.size();
        arena_.insert(arena_.end(), names.begin() + o.name_off, names.begin() + o.name_off + o.name_len);
    }
}

void OverlapNames::insert(const Read &r, const Pt &kill)
{
    if ((n_entries_ + n_tomb_ + 1) * 2 > tab_.size() || arena_.size() > ((size_t)1 << 22)) {
        size_t cap = 64;
        while (cap < (n_entries_ + 1) * 4) cap <<= 1;
        rebuild(cap);
    }
    const size_t mask = tab_.size() - 1;
    size_t s = (size_t)r.h & mask;
    while (tab_[s].used) s = (s + 1) & mask;
    Entry &e = tab_[s];
    if (e.tomb) { e.tomb = false; --n_tomb_; }
    e.used = true; e.h = r.h; e.holder = r.id; e.kill = kill;
    e.name_off = (uint32_t)arena_.size(); e.name_len = r.l_qname;
    arena_.insert(arena_.end(), r.qname, r.qname + r.l_qname);
    ++n_entries_;
}

void OverlapNames::erase(Entry *e)
{
    e->used = false; e->tomb = true;
    --n_entries_; ++n_tomb_;
    if (!n_entries_) { for (Entry &x : tab_) x = Entry{}; n_tomb_ = 0; arena_.clear(); }     // (an empty table starts over: no tombstones, no names)
}

void OverlapNames::flush_pending()
{
    for (const Slot &b : pend_) if (!before(b.end, last_)) buf_add(b.h, b.end);
    pend_.clear();
}

void OverlapNames::buf_add(uint64_t h, const Pt &end)
{
    if ((buf_used_ + 1) * 2 > buf_.size()) {
        // what the iterator has passed is no longer in the buffer
        std::vector<Slot> old;
        old.swap(buf_);
        size_t alive = 0;
        for (const Slot &o : old) if (o.h && !before(o.end, last_)) ++alive;
        size_t cap = 256;
        while (cap < (alive + 1) * 4) cap <<= 1;
        buf_.assign(cap, Slot{ 0, { 0, 0 } });
        buf_used_ = 0;
        for (const Slot &o : old) {
            if (!o.h || before(o.end, last_)) continue;
            size_t s = (size_t)o.h & (cap - 1);
            while (buf_[s].h) s = (s + 1) & (cap - 1);
            buf_[s] = o; ++buf_used_;
        }
    }
    const size_t mask = buf_.size() - 1;
    size_t s = (size_t)h & mask;
    while (buf_[s].h) s = (s + 1) & mask;
    buf_[s] = Slot{ h, end }; ++buf_used_;
}

int64_t OverlapNames::push(const Read &r, bool dropped)
{
    // (max_tid, max_pos) at this push = the previous record that was pushed and not turned away: bam_plp_auto has asked bam_plp_next for
    // every column in front of it, so whatever ended before that position has left the buffer -- and taken the entry of its NAME along
    // (overlap_remove)
    if (dropped) {
        // turned away by the -d cap: overlap_remove by name, max_pos stays
        if (Entry *e = find(r)) erase(e);
        return -1;
    }
    int64_t found = -1;
    if (r.end > r.pos) {                                  // a record without a reference span never enters the buffer
        Entry *e = find(r);
        if (e && before(e->kill, last_)) { erase(e); e = nullptr; }
        // overlap_push's conditions (the device's RI_OLAP_EL)
        const Pt my_end{ r.tid, r.end };
        if (eligible(r.flag, r.tid, r.mtid, r.l_qseq, r.end, r.mpos, r.isize)) {
            if (e) { found = e->holder; erase(e); }
            else if (r.mpos >= r.pos || ((r.flag & 1) && r.mpos == -1)) {
                // the entry leaves with the first record of this name that leaves the buffer: this one, or one that is in there already
                Pt kill = my_end;
                flush_pending();
                const size_t mask = buf_.size() - 1;
                for (size_t s = (size_t)r.h & mask; buf_[s].h; s = (s + 1) & mask) {
                    const Slot &b = buf_[s];
                    if (b.h != r.h || before(b.end, last_)) continue;
                    if (b.end.tid < kill.tid || (b.end.tid == kill.tid && b.end.pos < kill.pos)) kill = b.end;
                }
                insert(r, kill);
            }
        } else if (e) {
            // in the buffer under this name without touching the hash: it takes the entry along when it leaves
            if (my_end.tid < e->kill.tid || (my_end.tid == e->kill.tid && my_end.pos < e->kill.pos)) e->kill = my_end;
        }
        // (into the buffer's table only when somebody looks: single-end input never does, and the table insert -- with its rebuilds -- was a
        // third of this function's time on the producer thread, profiles/r06_sessionI_e2e_big.log)
        pend_.push_back(Slot{ r.h, my_end });
        if (pend_.size() >= 8192) thin_pending();
    }
    last_ = Pt{ r.tid, r.pos };
    return found;
}

void OverlapNames::thin_pending()
{
    size_t k = 0;
    for (const Slot &b : pend_) if (!before(b.end, last_)) pend_[k++] = b;
    pend_.resize(k);
    if (pend_.size() >= 4096) flush_pending();          // (really that many records in the pileup buffer: a deep pile)
}

}  // namespace sta
