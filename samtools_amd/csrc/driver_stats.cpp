// driver_stats.cpp -- `samtools-amd stats`: the coverage distribution of `samtools stats` (the "COV" section).
//
// stats.c keeps a round buffer of per-position depths (stats.c:311-391): every aligned block of a read adds 1 to its slots
// (:1452-1508), the slots behind the next read are binned with coverage_idx and cleared.  The engine does not keep that buffer:
// an aligned block is two marks (+1 at its first position, -1 behind its last) and the device bins runs of equal depth between
// sorted marks (sta_statcov_add, kernels_statcov.hip).  What this file keeps of the ring is its BOOKKEEPING, because the numbers the
// reference prints depend on it:
//   1. fold:  a block [from, to) of a read at P goes to ring offsets (from - P) mod size .. (to - P) mod size; size = 5 x the longest
//             read seen so far (at least 300).  A block further than `size` behind the read's start folds back onto its first positions.
//   2. stale: when the next read starts `size` or more later (or at a contig change / the end of the input) round_buffer_flush counts
//             every slot but the last; what that slot holds is added to the next read's first position (across a contig change: to the
//             position its slot index maps to in the next contig; at the end of the input it is dropped).
//   3. grow:  a read at least as long as the per-cycle arrays re-allocates the ring, copying `n` BYTES where n counts elements
//             (stats.c:771-774): of the pending depths only the first quarter (and the first quarter of the wrapped part) survives,
//             the element on the boundary keeps its low bytes.
// Pending marks (those at or beyond the last read's start) sit in an ordered map; a flush moves the marks in front of the new start
// to the output stream, which is therefore sorted; rules 2 and 3 read the pending depths back from the map (rare events).
// Target regions (-t file, stats.c:1954-2043; region arguments, :2104-2149): a read counts if it overlaps a region (is_in_regions,
// :2067-2102) and its aligned blocks are clipped to the regions it overlaps (:1454-1487).  With region arguments the reference reads
// through the index; here the whole file is read and the same filter decides, which gives the same section.
// -p (remove_overlaps, stats.c:1088-1210): the part of a second mate that the first mate's blocks already cover is not counted; the
// pair table's clean-up schedule (:1392-1402) is kept, because it decides which line of a template counts as the first.
// Options: -c min,max,step  -f / -F  -d  -l  -I  -t  -p  regions; -r -q -i -m -x -s -g are accepted (no effect on this section); -S is
// refused.  Only this section is printed (the comment line and the COV lines of stats.c:1884-1892).
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <map>
#include <set>
#include <unordered_map>

using namespace sta;

namespace {

struct Marks { std::vector<int64_t> pos; std::vector<int32_t> delta; };

// the ring's bookkeeping without the ring (see the file comment)
struct CovRing {
    int64_t S = 1500, s = 0, P = 0;            // size, start slot, position of slot `start` (round_buffer_t)
    int nbases = 300;
    std::map<int64_t, int64_t> pend;           // position -> depth change, positions >= P
    int64_t carry = 0;                         // depth in front of the first pending mark = sum of what has been emitted
    Marks out;                                 // the epoch's emitted marks, sorted by position
    std::string err;

    void mark(int64_t x, int64_t d) { if (d) { auto it = pend.emplace(x, 0).first; it->second += d; if (!it->second) pend.erase(it); } }
    void emit(int64_t x, int64_t d)
    {
        if (!d) return;
        // (a delta beyond int32 cannot happen: it is bounded by the number of reads starting or ending at one position within a batch)
        out.pos.push_back(x); out.delta.push_back((int32_t)d); carry += d;
    }
    void emit_below(int64_t lim) { while (!pend.empty() && pend.begin()->first < lim) { emit(pend.begin()->first, pend.begin()->second); pend.erase(pend.begin()); } }
    int64_t depth_at(int64_t x) const { int64_t d = carry; for (auto it = pend.begin(); it != pend.end() && it->first <= x; ++it) d += it->second; return d; }

    // round_buffer_insert_read (stats.c:373-391)
    bool insert(int64_t from, int64_t to)
    {
        if (to - from > S) { err = "The read length too big (" + std::to_string(to - from) + "), please increase the buffer length (currently " + std::to_string(S) + ")"; return false; }
        if (from < P) { err = "The reads are not sorted (" + std::to_string(from) + " comes after " + std::to_string(P) + ")."; return false; }
        const int64_t a = (from - P) % S, b = (to - P) % S;
        if (a < b) { mark(P + a, 1); mark(P + b, -1); }
        else if (a > b) { mark(P + a, 1); mark(P + S, -1); mark(P, 1); mark(P + b, -1); }
        return true;
    }

    // round_buffer_flush (stats.c:327-368); epoch_end: the caller takes `out` away afterwards (pos == -1)
    bool flush(int64_t pos, bool at_eof, int64_t *stale_slot, int64_t *stale_depth)
    {
        *stale_depth = 0;
        if (pos == P) return true;
        const int64_t new_pos = pos;
        const bool whole = pos == -1 || pos - P >= S;
        if (whole) pos = P + S - 1;
        if (pos < P) { err = "Expected coordinates in ascending order, got " + std::to_string(pos) + " after " + std::to_string(P); return false; }
        emit_below(pos);
        if (whole) {
            // every slot but the last was counted: the depth at P + S - 1 stays in its slot
            const int64_t x = pos, c = depth_at(x);
            emit_below(INT64_MAX);                      // the closing marks of the blocks that reached it
            if (c) { emit(x, -c); emit(x + 1, c); }     // ... but the stream must not count position x
            // (out is sorted up to here except for these two marks: x and x + 1 are <= P + S, the last pending key; re-sorted by take())
            const int64_t slot = (s + (pos - P) % S) % S;
            if (new_pos == -1) { *stale_slot = slot; *stale_depth = at_eof ? 0 : c; s = 0; }
            else { s = slot; if (c) { mark(new_pos, c); mark(new_pos + 1, -c); } }
        } else s = (s + (pos - P) % S) % S;
        P = new_pos;
        return true;
    }

    // realloc_buffers (stats.c:690-692, :766-778)
    void grow(int seq_len)
    {
        nbases = 2 * (1 + seq_len - nbases) + nbases;
        const int64_t S2 = (int64_t)seq_len * 5;
        if (!pend.empty()) {
            std::vector<int32_t> ring((size_t)S, 0), big((size_t)S2, 0);
            {   // the pending depths in ring order (offset k = position P + k)
                int64_t d = carry; auto it = pend.begin();
                for (int64_t k = 0; k < S; ++k) { while (it != pend.end() && it->first <= P + k) { d += it->second; ++it; } ring[(size_t)k] = (int32_t)d; }
            }
            const int64_t n = S - s;
            memcpy(big.data(), ring.data(), (size_t)n);                                  // n BYTES of the part from `start` on
            if (s > 1) memcpy(big.data() + n, ring.data() + n, (size_t)s);              // s bytes of the wrapped part
            pend.clear();
            int64_t prev = carry;
            for (int64_t k = 0; k < S2; ++k) if (big[(size_t)k] != prev) { pend[P + k] = big[(size_t)k] - prev; prev = big[(size_t)k]; }
            if (prev) pend[P + S2] = -prev;
        }
        s = 0; S = S2;
    }
};

// target regions per contig: 1-based, both ends included, sorted, overlapping ones merged (stats.c:2018-2031)
struct Regions {
    struct Ival { int64_t beg, end; };
    std::vector<std::vector<Ival>> pos;          // [tid]
    std::vector<size_t> cpos;                    // [tid]: first region that can still overlap a read (reads arrive sorted)
    std::vector<Ival> chunks;                    // the regions the current read overlaps, clipped to it
    bool on = false;
    void add(int tid, int64_t beg, int64_t end) { if ((size_t)tid >= pos.size()) pos.resize((size_t)tid + 1); pos[(size_t)tid].push_back({ beg, end }); on = true; }
    void finish()
    {
        for (auto &v : pos) {
            if (v.size() > 1) {
                std::sort(v.begin(), v.end(), [](const Ival &a, const Ival &b) { return a.beg != b.beg ? a.beg < b.beg : a.end < b.end; });
                size_t n = 0;
                for (size_t p = 1; p < v.size(); ++p) {
                    if (v[n].end < v[p].beg) v[++n] = v[p];
                    else if (v[n].end < v[p].end) v[n].end = v[p].end;
                }
                v.resize(n + 1);
            }
        }
        cpos.assign(pos.size(), 0);
    }
    // stats.c:2067-2102 (the caller has checked that the input is still sorted)
    bool contains(const Rec &r)
    {
        if (r.tid < 0 || (size_t)r.tid >= pos.size()) return false;
        const std::vector<Ival> &v = pos[(size_t)r.tid];
        size_t &c = cpos[(size_t)r.tid];
        if (c == v.size()) return false;
        size_t i = c;
        while (i < v.size() && v[i].end <= r.pos) ++i;
        if (i >= v.size()) { c = v.size(); return false; }
        const int64_t endpos = r.endpos();
        if (endpos < v[i].beg) return false;
        c = i;
        chunks.clear();
        for (; i < v.size(); ++i)
            if (r.pos < v[i].end && endpos >= v[i].beg) chunks.push_back({ std::max(r.pos + 1, v[i].beg), std::min(endpos, v[i].end) });
        return true;
    }
};

// -p: the pair table of stats.c (khash qn2pair)
struct PairTab {
    struct Pair { std::vector<std::pair<int64_t, int64_t>> chunks; unsigned first = 0; };
    std::unordered_map<std::string, Pair> tab;
    unsigned pair_count = 0, last_read_flush = 0; int last_pair_tid = -2;
    // stats.c:1056-1085
    unsigned cleanup(int64_t max)
    {
        unsigned count = 0;
        for (auto it = tab.begin(); it != tab.end();) { if (it->second.chunks.back().second < max) { it = tab.erase(it); ++count; } else ++it; }
        return count;
    }
    // stats.c:1392-1402, once per read that reaches the coverage code
    void schedule(int tid, int64_t pos)
    {
        last_read_flush++;
        if (pair_count > 10000 && last_read_flush > 10000) { pair_count -= cleanup(pos); last_read_flush = 0; }
        if (last_pair_tid != tid) { pair_count -= cleanup(INT64_MAX - 1); last_pair_tid = tid; last_read_flush = 0; }
    }
    // stats.c:1088-1210; [pmin, pmax) 0-based half open, pmin == -1: the line is finished.  What is to be counted goes to `ring`.
    bool block(CovRing &ring, const Rec &r, int64_t pmin, int64_t pmax)
    {
        const unsigned order = ((r.flag & 64) ? 1u : 0u) + ((r.flag & 128) ? 2u : 0u);
        const long long isz = r.isize < 0 ? -(long long)r.isize : (long long)r.isize;
        if (!(r.flag & 1) || (r.flag & 8) || isz >= 2ll * r.l_qseq || (order != 1 && order != 2)) return pmin >= 0 ? ring.insert(pmin, pmax) : true;
        auto it = tab.find(r.qname);
        if (it == tab.end()) {
            if (pmin == -1) return true;
            Pair &pc = tab[r.qname];
            pc.chunks.push_back({ pmin, pmax }); pc.first = order;
            pair_count++;
        } else {
            Pair &pc = it->second;
            if (order == pc.first) {
                if (pmin == -1) return true;
                pc.chunks.push_back({ pmin, pmax });
            } else {
                if (pmin == -1) { tab.erase(it); pair_count--; return true; }
                for (const auto &ch : pc.chunks) {
                    if (pmin >= ch.second) continue;
                    if (pmax <= ch.first) break;
                    if (pmin < ch.first) { if (!ring.insert(pmin, ch.first)) return false; pmin = ch.first; }
                    if (pmax <= ch.second) return true;
                    pmin = ch.second;
                }
            }
        }
        return ring.insert(pmin, pmax);
    }
};

int unclipped_length(const Rec &r)
{
    int len = r.l_qseq;
    for (uint32_t c : r.cigar) if ((c & 0xf) == 5) len += (int)(c >> 4);
    return len;
}

struct Sink {
    sta_engine *eng = nullptr;
    FILE *dump = nullptr;          // --marks-out: the stream as text instead of the device (host-side check of the bookkeeping)
    int epoch = 0;
    int64_t carry_in = 0;          // depth in front of the batch
    bool send(Marks &m, bool last)
    {
        // marks of one position may have been appended out of order by the stale rule: a stable sort restores the order
        const size_t n = m.pos.size();
        bool sorted = true;
        for (size_t i = 1; i < n && sorted; ++i) sorted = m.pos[i - 1] <= m.pos[i];
        if (!sorted) {
            std::vector<size_t> ix(n); for (size_t i = 0; i < n; ++i) ix[i] = i;
            std::stable_sort(ix.begin(), ix.end(), [&](size_t a, size_t b) { return m.pos[a] < m.pos[b]; });
            Marks t; t.pos.resize(n); t.delta.resize(n);
            for (size_t i = 0; i < n; ++i) { t.pos[i] = m.pos[ix[i]]; t.delta[i] = m.delta[ix[i]]; }
            m.pos.swap(t.pos); m.delta.swap(t.delta);
        }
        // the last mark of a batch only closes the last run: keep it for the next batch unless the epoch ends
        size_t take = last ? n : (n ? n - 1 : 0);
        if (dump) { for (size_t i = 0; i < take; ++i) fprintf(dump, "%d\t%lld\t%d\n", epoch, (long long)m.pos[i], (int)m.delta[i]); }
        else if (n >= 2 || (last && n)) {
            if (last) { m.pos.push_back(m.pos.back()); m.delta.push_back(0); }       // a closing sentinel
            if (sta_statcov_add(eng, m.pos.data(), m.delta.data(), (int64_t)m.pos.size(), carry_in, STA_MEM_HOST) != STA_OK) {
                fprintf(stderr, "samtools stats: %s\n", sta_last_error(eng)); return false;
            }
            if (last) { m.pos.pop_back(); m.delta.pop_back(); }
        }
        for (size_t i = 0; i < take; ++i) carry_in += m.delta[i];
        m.pos.erase(m.pos.begin(), m.pos.begin() + (long)take);
        m.delta.erase(m.delta.begin(), m.delta.begin() + (long)take);
        if (last) { ++epoch; carry_in = 0; }
        return true;
    }
};

}  // namespace

extern "C" int sta_main_stats(int argc, char **argv)
{
    int c, flag_require = 0, flag_filter = 0, filter_readlen = -1, tmp;
    bool remove_olap = false;
    int cov_min = 1, cov_max = 1000, cov_step = 1;
    const char *group_id = nullptr, *marks_out = nullptr, *targets = nullptr;
    static const struct option lopts[] = {
        { "coverage", required_argument, NULL, 'c' }, { "required-flag", required_argument, NULL, 'f' }, { "filtering-flag", required_argument, NULL, 'F' },
        { "remove-dups", no_argument, NULL, 'd' }, { "read-length", required_argument, NULL, 'l' }, { "id", required_argument, NULL, 'I' },
        { "ref-seq", required_argument, NULL, 'r' }, { "insert-size", required_argument, NULL, 'i' }, { "most-inserts", required_argument, NULL, 'm' },
        { "trim-quality", required_argument, NULL, 'q' }, { "sparse", no_argument, NULL, 'x' }, { "sam", no_argument, NULL, 's' },
        { "target-regions", required_argument, NULL, 't' }, { "cov-threshold", required_argument, NULL, 'g' },
        { "marks-out", required_argument, NULL, 1 }, { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    while ((c = getopt_long(argc, argv, "dsxr:c:l:i:m:q:f:F:I:t:g:pS:", lopts, NULL)) >= 0) {
        switch (c) {
        case 'f': if ((tmp = str2flag(optarg)) < 0) { fprintf(stderr, "samtools stats: Unknown flag '%s'\n", optarg); return 1; } flag_require = tmp; break;
        case 'F': if ((tmp = str2flag(optarg)) < 0) { fprintf(stderr, "samtools stats: Unknown flag '%s'\n", optarg); return 1; } flag_filter |= tmp; break;
        case 'd': flag_filter |= 1024; break;
        case 'c': if (sscanf(optarg, "%d,%d,%d", &cov_min, &cov_max, &cov_step) != 3) { fprintf(stderr, "Unable to parse -c %s\n", optarg); return 1; } break;
        case 'l': filter_readlen = atoi(optarg); break;
        case 'I': group_id = optarg; break;
        case 1: marks_out = optarg; break;
        case 't': targets = optarg; break;
        case 'p': remove_olap = true; break;
        case 'r': case 'i': case 'm': case 'q': case 'x': case 's': case 'g': break;
        default: fprintf(stderr, "[stats] option -%c is not part of the engine's section (COV)\n", c); return 1;
        }
    }
    if (argc - optind < 1) { fprintf(stderr, "usage: samtools-amd stats [-c min,max,step] [-f INT] [-F INT] [-d] [-l INT] [-I ID] [-t targets] in.bam [region ...]   (prints the COV section)\n"); return 1; }
    if (!marks_out && sta_device_count() < 1) { fprintf(stderr, "samtools stats: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    std::string err;
    auto rd = AlnReader::open(argv[optind], &err);
    if (!rd) { fprintf(stderr, "samtools stats: failed to open \"%s\"\n", argv[optind]); return 1; }
    const Header &h = rd->header();
    // stats.c:2396-2411
    if (cov_step > cov_max - cov_min + 1) { cov_step = cov_max - cov_min; if (cov_step <= 0) cov_step = 1; }
    const int ncov = 3 + (cov_max - cov_min) / cov_step;
    cov_max = cov_min + ((cov_max - cov_min) / cov_step + 1) * cov_step - 1;
    // stats.c:2151-2177: the read groups whose ID or SM is the -I value
    std::set<std::string> rg_ok;
    if (group_id) {
        size_t p = 0;
        while (p < h.text.size()) {
            size_t e = h.text.find('\n', p); if (e == std::string::npos) e = h.text.size();
            if (e - p > 4 && h.text.compare(p, 4, "@RG\t") == 0) {
                std::string id, sm; bool has_id = false, has_sm = false;
                size_t f = p + 4;
                while (f < e) {
                    size_t g = h.text.find('\t', f); if (g == std::string::npos || g > e) g = e;
                    if (g - f >= 3 && h.text.compare(f, 3, "ID:") == 0 && !has_id) { id = h.text.substr(f + 3, g - f - 3); has_id = true; }
                    if (g - f >= 3 && h.text.compare(f, 3, "SM:") == 0 && !has_sm) { sm = h.text.substr(f + 3, g - f - 3); has_sm = true; }
                    f = g + 1;
                }
                if (has_id && (id == group_id || (has_sm && sm == group_id))) rg_ok.insert(id);
            }
            p = e + 1;
        }
    }
    Regions regs;
    if (targets) {
        // stats.c:1954-2016: "name beg end" lines, '#' comments; names the header does not know are skipped with one warning
        FILE *fp = fopen(targets, "r");
        if (!fp) { fprintf(stderr, "%s: cannot open\n", targets); return 1; }
        char line[4096];
        bool warned = false; int prev_tid = -1; long long prev_pos = -1;
        while (fgets(line, sizeof line, fp)) {
            if (line[0] == '#') continue;
            size_t l = strlen(line);
            while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
            size_t i = 0;
            while (i < l && !isspace((unsigned char)line[i])) i++;
            if (i >= l) { fprintf(stderr, "Could not parse the file: %s [%s]\n", targets, line); fclose(fp); return 1; }
            line[i] = 0;
            const int tid = h.tid(line);
            if (tid < 0) {
                if (!warned) fprintf(stderr, "Warning: Some sequences not present in the BAM, e.g. \"%s\". This message is printed only once.\n", line);
                warned = true;
                continue;
            }
            long long b, e;
            if (sscanf(line + i + 1, "%lld %lld", &b, &e) != 2) { fprintf(stderr, "Could not parse the region [%s]\n", line + i + 1); fclose(fp); return 1; }
            if (prev_tid == -1 || prev_tid != tid) { prev_tid = tid; prev_pos = b; }
            if (prev_pos > b) { fprintf(stderr, "The positions are not in chromosomal order (%s:%lld comes after %lld)\n", line, b, prev_pos); fclose(fp); return 1; }
            regs.add(tid, b, e);
        }
        fclose(fp);
        if (!regs.on) { fprintf(stderr, "Unable to map the -t sequences to the BAM sequences.\n"); return 1; }
        regs.finish();
    } else if (argc - optind > 1) {
        for (int a = optind + 1; a < argc; ++a) {
            int t; int64_t rb, re;
            if (!parse_region(h, argv[a], &t, &rb, &re)) { fprintf(stderr, "Multi-region iterator could not be created\n"); return 1; }
            regs.add(t, rb + 1, re);
        }
        regs.finish();
    }
    Sink sink;
    if (marks_out) { sink.dump = fopen(marks_out, "w"); if (!sink.dump) { fprintf(stderr, "samtools stats: cannot write %s\n", marks_out); return 1; } }
    else {
        if (sta_engine_create(&sink.eng, 0, nullptr) != STA_OK) { fprintf(stderr, "samtools stats: no usable HIP device\n"); return 2; }
        sta_statcov_params sp{ cov_min, cov_max, cov_step };
        int32_t nc = 0;
        if (sta_statcov_begin(sink.eng, &sp, &nc) != STA_OK || nc != ncov) { fprintf(stderr, "samtools stats: %s\n", sta_last_error(sink.eng)); return 1; }
    }
    CovRing ring;
    PairTab pairs;
    auto count = [&](const Rec &rec, int64_t a, int64_t b) { return remove_olap ? pairs.block(ring, rec, a, b) : ring.insert(a, b); };
    size_t batch = 1 << 21;                     // marks per device call
    if (const char *e = getenv("STA_STATS_BATCH")) batch = (size_t)std::max<long long>(2, atoll(e));
    bool is_sorted = true;
    int cur_tid = -1, status = 0;
    int64_t last_pos = -1, stale_slot = 0, stale_depth = 0;
    auto end_epoch = [&](bool at_eof) -> bool {
        if (!ring.flush(-1, at_eof, &stale_slot, &stale_depth)) return false;
        if (!sink.send(ring.out, true)) return false;
        ring.carry = 0;
        // rule 2 across a contig change: the slot keeps its index; with start = 0 and pos = -1 it now stands for position slot - 1
        if (stale_depth) { ring.mark(stale_slot - 1, stale_depth); ring.mark(stale_slot, -stale_depth); }
        return true;
    };
    Rec r;
    int st;
    while ((st = rd->next(r)) > 0) {
        // stats.c:1212-1273
        if (regs.on) {
            if (r.tid < 0 || (size_t)r.tid >= regs.pos.size()) continue;
            if (!is_sorted) { fprintf(stderr, "The BAM must be sorted in order for -t to work.\n"); status = 1; break; }
            if (!regs.contains(r)) continue;
        }
        if (group_id) { if (r.rg.empty() || !rg_ok.count(r.rg)) continue; }
        if (flag_require && (r.flag & flag_require) != flag_require) continue;
        if (flag_filter && (r.flag & flag_filter)) continue;
        if (filter_readlen != -1 && r.l_qseq != filter_readlen) continue;
        if (r.flag & 256) continue;
        if (!r.l_qseq) continue;
        const int read_len = unclipped_length(r);
        if (read_len >= ring.nbases) ring.grow(read_len);
        if (r.flag & 4) continue;
        if (r.cigar.empty()) { fprintf(stderr, "FIXME: mapped read with no cigar?\n"); status = 1; break; }
        // stats.c:1380-1393
        if (cur_tid == r.tid && r.pos < last_pos) is_sorted = false;
        last_pos = r.pos;
        if (!is_sorted) continue;
        if (cur_tid == -1 || cur_tid != r.tid) { if (!end_epoch(false)) { status = 1; break; } }
        pairs.schedule(r.tid, r.pos);
        cur_tid = r.tid;
        // stats.c:1452-1508
        int64_t sslot, sdepth;
        if (!ring.flush(r.pos, false, &sslot, &sdepth)) { status = 1; break; }
        int64_t p = r.pos;
        bool bad = false;
        if (regs.on) {
            // stats.c:1454-1487: every aligned block clipped to the chunks; a block that reaches beyond a chunk is looked at again with the next
            size_t j = 0, i = 0;
            while (j < r.cigar.size() && i < regs.chunks.size()) {
                const int op = (int)(r.cigar[j] & 0xf); const int64_t len = (int64_t)(r.cigar[j] >> 4);
                if (op == 0 || op == 7 || op == 8) {
                    const int64_t pmin = std::max(p, regs.chunks[i].beg - 1), pmax = std::min(p + len, regs.chunks[i].end);
                    if (pmax > pmin && !count(r, pmin, pmax)) { bad = true; break; }
                }
                const int64_t pnew = p + ((op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ? len : 0);
                if (pnew >= regs.chunks[i].end) ++i;
                else { ++j; p = pnew; }
            }
        } else
        for (uint32_t cg : r.cigar) {
            const int op = (int)(cg & 0xf); const int64_t len = (int64_t)(cg >> 4);
            if (op == 0 || op == 7 || op == 8) { if (!count(r, p, p + len)) { bad = true; break; } }
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) p += len;
        }
        if (bad) { status = 1; break; }
        if (remove_olap) pairs.block(ring, r, -1, -1);               // the line is finished (stats.c:1509-1510)
        if (ring.out.pos.size() >= batch && !sink.send(ring.out, false)) { status = 1; break; }
    }
    if (!ring.err.empty()) fprintf(stderr, "%s\n", ring.err.c_str());
    if (st < 0) { fprintf(stderr, "Failure while decoding file\n"); status = 1; }
    if (!status && !end_epoch(true)) { if (!ring.err.empty()) fprintf(stderr, "%s\n", ring.err.c_str()); status = 1; }
    if (sink.dump) { fprintf(sink.dump, "#sorted\t%d\n", is_sorted ? 1 : 0); fclose(sink.dump); if (sink.eng) sta_engine_destroy(sink.eng); return status; }
    if (!status && is_sorted) {
        std::vector<uint64_t> cov((size_t)ncov, 0);
        if (sta_statcov_fetch(sink.eng, cov.data(), ncov) != STA_OK) { fprintf(stderr, "samtools stats: %s\n", sta_last_error(sink.eng)); status = 1; }
        else {
            // stats.c:1884-1892
            printf("# Coverage distribution. Use `grep ^COV | cut -f 2-` to extract this part.\n");
            if (cov[0]) printf("COV\t[<%d]\t%d\t%ld\n", cov_min, cov_min - 1, (long)cov[0]);
            for (int i = 1; i < ncov - 1; i++)
                if (cov[(size_t)i]) printf("COV\t[%d-%d]\t%d\t%ld\n", cov_min + (i - 1) * cov_step, cov_min + i * cov_step - 1, cov_min + i * cov_step - 1, (long)cov[(size_t)i]);
            if (cov[(size_t)ncov - 1]) printf("COV\t[%d<]\t%d\t%ld\n", cov_min + (ncov - 2) * cov_step - 1, cov_min + (ncov - 2) * cov_step - 1, (long)cov[(size_t)ncov - 1]);
        }
    }
    sta_engine_destroy(sink.eng);
    return status;
}
