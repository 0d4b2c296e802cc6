// host_names.h -- the two read-name hashes of the path, kept on the HOST in file order.
//
// Both reference hashes are sequential state over the whole input: what a record finds under its name depends on every earlier record
// of its template, wherever the window cuts fall.
//   * depth -s    bam2depth.c:598-623 -- name -> end of the first-seen record; the next record of the name is clipped below that end and
//                 takes the entry out; an entry never leaves otherwise ("never forgets").
//   * mpileup     HTSlib's overlap_push / overlap_remove (switched on at bam_plcmd.c:586): name -> the record that put
//                 it; the next eligible record of the name is resolved against that record (tweak_overlap_quality) and takes the entry
//                 out; the entry also leaves BY NAME when any record of that name leaves the pileup buffer or is turned away by the -d cap.
// Rounds 2-5 replayed both per window on the device from the staged records and had the host guess which ended records still had to
// be staged for the replay to come out right (six "retired too early / kept too long" findings in round 5).  Here the input lanes
// (host_pump / host_chunk) run the state machines themselves while they take records off the files, and hand the device what each
// record found: a clip column (depth) or the staged index of its partner (mpileup).  No window depends on a record it does not stage.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace sta {

// 64-bit name hash (FNV-1a and a final avalanche); computed by the decode threads for the chunk lane (Chunk::name_h)
inline uint64_t qname_hash64(const char *s, size_t l)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < l; ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    return h ? h : 1;
}

// the read filters of fastdepth_core (bam2depth.c:552-571 = :646-660), as k_prep_reads_depth applies them on the device: only a record
// that passes them reaches the name hash
struct DepthReadFilter {
    int flag = 0, incl_flag = 0, require_flag = 0, min_mqual = 0, min_len = 0;
    bool passes(unsigned rflag, int mapq, int32_t l_qseq, const uint32_t *cigar, size_t n_cigar) const
    {
        if (rflag & (unsigned)flag) return false;
        if (incl_flag && (rflag & (unsigned)incl_flag) == 0) return false;
        if ((rflag & (unsigned)require_flag) != (unsigned)require_flag) return false;
        if (mapq < min_mqual) return false;
        if (min_len) {
            // qlen_used (bam2depth.c:124-159)
            int64_t l;
            if (l_qseq) {
                l = l_qseq;
                size_t kl, kr;
                for (kl = 0; kl < n_cigar; ++kl) { if ((cigar[kl] & 0xf) == 4) l -= (cigar[kl] >> 4); else break; }
                for (kr = n_cigar; kr > kl + 1; --kr) { if ((cigar[kr - 1] & 0xf) == 4) l -= (cigar[kr - 1] >> 4); else break; }
            } else {
                l = 0;
                for (size_t k = 0; k < n_cigar; ++k) { const int op = (int)(cigar[k] & 0xf); if (op == 0 || op == 1 || op == 7 || op == 8) l += (cigar[k] >> 4); }
            }
            if (l < min_len) return false;
        }
        return true;
    }
};

// depth -s, one input file (the reference keeps one hash per file: bam2depth.c:518-531)
class DepthMateClip {
public:
    // a record that passed the read filters, in file order; returns the column below which it is not counted (0: none)
    int64_t visit(const char *qname, unsigned flag, int32_t tid, int64_t endpos, int32_t mtid, int64_t mpos)
    {
        if (!(flag & 1) || (flag & 8)) return 0;                    // BAM_FPAIRED && !BAM_FMUNMAP
        auto it = h_.find(qname);
        if (it == h_.end()) {
            // not seen before: "Don't add if mate location is known and can't overlap"
            if (mpos == -1 || (tid == mtid && mpos <= endpos)) h_.emplace(qname, endpos);
            return 0;
        }
        const int64_t clip = it->second;
        h_.erase(it);
        return clip;
    }
    size_t size() const { return h_.size(); }
private:
    std::unordered_map<std::string, int64_t> h_;
};

// mpileup's overlap hash, one input file (one bam_plp_t per file: bam_mplp_init_overlaps)
class OverlapNames {
public:
    struct Read {
        uint64_t h; const char *qname; uint32_t l_qname;          // name, its qname_hash64
        unsigned flag; int32_t tid, mtid, l_qseq; int64_t pos, end /* pos + reference span */, mpos, isize;
        int64_t id;                                               // the lane's running number of the record
    };
    // A record that reaches bam_plp_push, in file order.  dropped: the -d cap turned it away at the push.  Returns the id of the record
    // whose entry it found -- tweak_overlap_quality(that record, this one) -- or -1.
    int64_t push(const Read &r, bool dropped);
    size_t live_entries() const { return n_entries_; }
    // overlap_push's conditions (the device's RI_OLAP_EL)
    static bool eligible(unsigned flag, int32_t tid, int32_t mtid, int32_t l_qseq, int64_t end, int64_t mpos, int64_t isize)
    {
        if ((flag & (8u | 2u)) != 2u) return false;               // mate unmapped, or not a proper pair
        const int64_t isz = isize < 0 ? -isize : isize;
        return !((mtid >= 0 && mtid != tid) || (isz >= 2 * (int64_t)l_qseq && mpos >= end));
    }
    // push() of a record that was not turned away, is not eligible and meets an empty table (single-end input: every record): it only
    // enters the buffer and moves the iterator -- the lanes call this without building a Read (the name is not looked at)
    bool plain_case(bool dropped, bool is_eligible) const { return !dropped && !is_eligible && !n_entries_; }
    void push_plain(uint64_t h, int32_t tid, int64_t pos, int64_t end)
    {
        if (end > pos) { pend_.push_back(Slot{ h, Pt{ tid, end } }); if (pend_.size() >= 8192) thin_pending(); }
        last_ = Pt{ tid, pos };
    }
private:
    struct Pt { int32_t tid; int64_t pos; };
    static bool before(const Pt &kill, const Pt &max) { return max.tid > kill.tid || (max.tid == kill.tid && max.pos > kill.pos); }    // the iterator has passed `kill`
    Pt last_{ -1, INT64_MIN };                                    // (max_tid, max_pos) of the iterator
    // the entries: open addressing on the name hash, names kept in an arena for the exact comparison
    struct Entry { uint64_t h; uint32_t name_off, name_len; int64_t holder; Pt kill; bool used, tomb; };
    std::vector<Entry> tab_ = std::vector<Entry>(64);
    std::vector<char> arena_;
    size_t n_entries_ = 0, n_tomb_ = 0;
    Entry *find(const Read &r);
    void insert(const Read &r, const Pt &kill);
    void erase(Entry *e);
    void rebuild(size_t cap);
    // every record in the pileup buffer under its name hash (a multiset; stale slots are skipped by their end and dropped at rebuilds)
    struct Slot { uint64_t h; Pt end; };
    std::vector<Slot> buf_ = std::vector<Slot>(256);
    size_t buf_used_ = 0;
    void buf_add(uint64_t h, const Pt &end);
    std::vector<Slot> pend_;                                      // buffered records not in the table yet (flush_pending)
    void flush_pending();
    void thin_pending();
};

}  // namespace sta
