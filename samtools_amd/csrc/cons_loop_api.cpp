// cons_loop_api.cpp -- sta_pileup_loop(): consensus_pileup.h's pileup_loop() (reference implementation
// consensus_pileup.c:301-608) on the device engine.  See include/samtools_amd_cons.h for the contract.
#include "../../include/samtools_amd.h"
#include "../../include/samtools_amd_cons.h"
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>

namespace {

struct Live {                     // one record that is (or was, for the look-back position) in the pileup
    sta_pileup_t *p;
    int64_t end;                  // position after its last reference base (0-based, exclusive)
    bool dead;                    // seq_free already ran: stays only so that the next window sees its insertions
};

struct Stage {                    // structure-of-arrays copy of the records of one window (sta_reads)
    std::vector<int32_t> pos, l_qseq, mtid, isize;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq, aux, seq, qual;
    std::vector<uint32_t> cig_off, base_off8, name_off, cigar;
    std::vector<int64_t> mpos;
    std::vector<char> names;
    void clear()
    {
        pos.clear(); l_qseq.clear(); mtid.clear(); isize.clear(); flag.clear(); mapq.clear(); aux.clear(); seq.clear(); qual.clear();
        cig_off.clear(); base_off8.clear(); name_off.clear(); cigar.clear(); mpos.clear(); names.clear();
    }
    void add(const bam1_t *b, int64_t origin)
    {
        pos.push_back((int32_t)(b->core.pos - origin));
        flag.push_back(b->core.flag); mapq.push_back(b->core.qual); aux.push_back(0);
        const int32_t lq = b->core.l_qseq;
        l_qseq.push_back(lq);
        cig_off.push_back((uint32_t)cigar.size());
        const uint32_t *cg = bam_get_cigar(b);
        cigar.insert(cigar.end(), cg, cg + b->core.n_cigar);
        const size_t b0 = qual.size(), padded = ((size_t)lq + 7) & ~(size_t)7;
        base_off8.push_back((uint32_t)(b0 >> 3));
        qual.resize(b0 + padded, 0); seq.resize((b0 + padded) / 2, 0);
        if (lq) { memcpy(&qual[b0], bam_get_qual(b), (size_t)lq); memcpy(&seq[b0 / 2], bam_get_seq(b), ((size_t)lq + 1) / 2); }
        mtid.push_back(b->core.mtid); mpos.push_back(b->core.mpos); isize.push_back(0);
        name_off.push_back((uint32_t)names.size());
        names.push_back('\0');
    }
    sta_reads view()
    {
        cig_off.push_back((uint32_t)cigar.size()); name_off.push_back((uint32_t)names.size());
        sta_reads v; memset(&v, 0, sizeof v);
        v.n_reads = (int64_t)pos.size();
        v.pos = pos.data(); v.flag = flag.data(); v.mapq = mapq.data(); v.aux = aux.data(); v.l_qseq = l_qseq.data();
        v.cig_off = cig_off.data(); v.base_off8 = base_off8.data(); v.mtid = mtid.data(); v.mpos = mpos.data(); v.isize = isize.data();
        v.name_off = name_off.data(); v.cigar = cigar.data(); v.seq = seq.data(); v.qual = qual.data(); v.names = names.data();
        v.n_cigar_total = cigar.size(); v.n_bases_total = qual.size(); v.n_name_bytes = names.size();
        return v;
    }
};

int64_t ref_end(const bam1_t *b)
{
    int64_t e = b->core.pos;
    const uint32_t *cg = bam_get_cigar(b);
    for (uint32_t k = 0; k < b->core.n_cigar; ++k) { const int op = (int)(cg[k] & 15u); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += cg[k] >> 4; }
    return e;
}

}  // namespace

static int pileup_loop_impl(samFile *fp, sam_hdr_t *h,
                            int (*seq_fetch)(void *, samFile *, sam_hdr_t *, bam1_t *),
                            int (*seq_init)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                            int (*seq_column)(void *, samFile *, sam_hdr_t *, sta_pileup_t *, int, hts_pos_t, int, int),
                            void (*seq_free)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                            void *cd);

extern "C" int sta_pileup_loop(samFile *fp, sam_hdr_t *h,
                               int (*seq_fetch)(void *, samFile *, sam_hdr_t *, bam1_t *),
                               int (*seq_init)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                               int (*seq_column)(void *, samFile *, sam_hdr_t *, sta_pileup_t *, int, hts_pos_t, int, int),
                               void (*seq_free)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                               void *cd)
{
    // no C++ exception may cross the C boundary
    try { return pileup_loop_impl(fp, h, seq_fetch, seq_init, seq_column, seq_free, cd); }
    catch (const std::exception &e) { fprintf(stderr, "pileup_loop: %s\n", e.what()); return -1; }
    catch (...) { fprintf(stderr, "pileup_loop: unexpected failure\n"); return -1; }
}

static int pileup_loop_impl(samFile *fp, sam_hdr_t *h,
                               int (*seq_fetch)(void *, samFile *, sam_hdr_t *, bam1_t *),
                               int (*seq_init)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                               int (*seq_column)(void *, samFile *, sam_hdr_t *, sta_pileup_t *, int, hts_pos_t, int, int),
                               void (*seq_free)(void *, samFile *, sam_hdr_t *, sta_pileup_t *),
                               void *cd)
{
    if (!seq_fetch || !seq_column) return -1;
    if (sta_device_count() < 1) { fprintf(stderr, "pileup_loop: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return -1; }
    sta_engine *eng = nullptr;
    int dev = 0;
    if (const char *e = getenv("STA_DEVICE")) dev = atoi(e);
    if (sta_engine_create(&eng, dev, nullptr) != STA_OK) { fprintf(stderr, "pileup_loop: no usable HIP device\n"); return -1; }
    size_t batch = 65536;
    if (const char *e = getenv("STA_PLP_BATCH")) batch = (size_t)std::max(1, atoi(e));

    std::vector<Live> reads;                  // staged order = file order: carried reads, then the batch
    std::vector<sta_pileup_t *> pool;         // recycled records (their bam1_t buffers are reused by seq_fetch)
    sta_pileup_t *ahead = nullptr;            // pulled, accepted by seq_init, belongs to the next window (or contig)
    bool at_eof = false;
    int ret = -1, cur_tid = -1;
    int64_t cursor = 0, last_pos = -1;        // first column not yet handed out; start of the last pulled record
    Stage st;
    std::vector<int32_t> ins, first_col, last_col;
    std::vector<uint64_t> entry_off;
    std::vector<uint32_t> entries, seq_offs;
    auto recycle = [&](sta_pileup_t *p) { pool.push_back(p); };
    auto fresh = [&]() -> sta_pileup_t * {
        if (!pool.empty()) { sta_pileup_t *p = pool.back(); pool.pop_back(); return p; }
        return (sta_pileup_t *)calloc(1, sizeof(sta_pileup_t));
    };
    // pulls the next record that enters the pileup; 1 = got one, 0 = end of input, -1 = error
    auto pull = [&](sta_pileup_t *&out) -> int {
        for (;;) {
            sta_pileup_t *p = fresh();
            if (!p) return -1;
            const int r = seq_fetch(cd, fp, h, &p->b);
            if (r < -1) { fprintf(stderr, "pileup_loop() seq_fetch failure.\n"); recycle(p); return -1; }
            if (r < 0) { recycle(p); return 0; }
            if ((p->b.core.flag & 4) || p->b.core.tid == -1) { recycle(p); continue; }
            if (p->b.core.tid == cur_tid && (int64_t)p->b.core.pos < last_pos) {
                fprintf(stderr, "BAM/SAM file is not sorted by position. Aborting\n"); recycle(p); return -1;
            }
            // (a record of an earlier contig than the current one is the reference's "new contig": it is not checked there either)
            bam1_t keep = p->b;
            memset(p, 0, sizeof *p);
            p->b = keep;
            p->b_is_rev = (p->b.core.flag & 16) != 0;
            p->b_qual = bam_get_qual(&p->b); p->b_seq = bam_get_seq(&p->b); p->b_cigar = bam_get_cigar(&p->b);
            p->start = 2; p->seq_offset = -1; p->cigar_op = -1; p->pos = p->b.core.pos;
            const int tid_before = cur_tid; const int64_t pos_before = last_pos;
            cur_tid = p->b.core.tid; last_pos = p->b.core.pos;
            if (seq_init) {
                const int v = seq_init(cd, fp, h, p);
                if (v == -1) { recycle(p); return -1; }
                if (v != 1) { recycle(p); (void)tid_before; (void)pos_before; continue; }
            }
            out = p;
            return 1;
        }
    };

    for (;;) {
        // ---- pull a batch of one contig ----
        int win_tid = reads.empty() ? -1 : reads.front().p->b.core.tid;
        size_t pulled = 0;
        bool more_same_contig = false;
        while (!at_eof) {
            sta_pileup_t *p = ahead;
            ahead = nullptr;
            if (!p) {
                const int r = pull(p);
                if (r < 0) goto done;
                if (r == 0) { at_eof = true; break; }
            }
            if (win_tid == -1) win_tid = p->b.core.tid;
            if (p->b.core.tid != win_tid) { ahead = p; break; }
            if (pulled >= batch) { ahead = p; more_same_contig = true; break; }
            const int64_t e = ref_end(&p->b);
            if (e <= (int64_t)p->b.core.pos) {                      // no reference-consuming operation: never in a column
                if (seq_free) seq_free(cd, fp, h, p);
                recycle(p);
                continue;
            }
            reads.push_back(Live{ p, e, false });
            ++pulled;
        }
        if (reads.empty()) { cursor = 0; if (at_eof && !ahead) break; continue; }

        // ---- window [cursor, we): everything before the next unread record of this contig, and no further than the reads
        //      reach without a gap (columns exist only while some read is alive; pileup_loop jumps over holes) ----
        int64_t alive_from = INT64_MAX;
        for (const Live &l : reads) if (!l.dead && l.end > cursor) alive_from = std::min<int64_t>(alive_from, std::max<int64_t>(cursor, l.p->b.core.pos));
        const bool any_alive = alive_from != INT64_MAX;
        if (!any_alive || alive_from > cursor) {
            // the reads that ended on the column before `cursor` were kept for that position's insertion count only
            size_t keep = 0;
            for (Live &l : reads) { if (l.dead) recycle(l.p); else reads[keep++] = l; }
            reads.resize(keep);
        }
        if (!any_alive) { if (reads.empty()) { cursor = 0; continue; } }
        if (any_alive) cursor = alive_from;
        int64_t cover_end = cursor, we = INT64_MAX;
        for (const Live &l : reads) {
            if (l.dead) continue;
            if ((int64_t)l.p->b.core.pos > cover_end) break;       // a hole: the window stops in front of it
            cover_end = std::max(cover_end, l.end);
        }
        we = cover_end;
        if (more_same_contig) we = std::min<int64_t>(we, (int64_t)ahead->b.core.pos);
        if (any_alive && we > cursor) {
            if (we - cursor > (int64_t)1 << 24) we = cursor + ((int64_t)1 << 24);       // long reference skips: several windows
            st.clear();
            for (const Live &l : reads) st.add(&l.p->b, cursor);
            sta_reads view = st.view();
            sta_window w; memset(&w, 0, sizeof w);
            w.tid = win_tid; w.origin = cursor; w.col_beg = 0; w.col_end = (int32_t)(we - cursor);
            w.tname = ""; w.tlen = INT64_MAX; w.n_files = 1; w.files = &view; w.mem = STA_MEM_HOST;
            sta_cons_info info;
            if (sta_stage_window(eng, &w) != STA_OK || sta_cons_entries_run(eng, &info) != STA_OK) { fprintf(stderr, "pileup_loop: %s\n", sta_last_error(eng)); goto done; }
            const size_t n = reads.size(), W = (size_t)(we - cursor);
            ins.resize(W); first_col.resize(n); last_col.resize(n); entry_off.resize(n + 1);
            entries.resize((size_t)info.n_entries + 1); seq_offs.resize((size_t)info.n_entries + 1);
            if (sta_fetch_cons_entries(eng, ins.data(), first_col.data(), last_col.data(), entry_off.data(), entries.data(), seq_offs.data()) != STA_OK) {
                fprintf(stderr, "pileup_loop: %s\n", sta_last_error(eng)); goto done;
            }
            // ---- hand the columns out ----
            sta_pileup_t *head = nullptr, *tail = nullptr;
            size_t next_read = 0;                              // reads enter in staged (= file) order: first_col is ascending
            std::vector<size_t> active;                        // indices into reads, in list order
            int32_t c = 0;
            for (size_t pi = 0; pi < W; ++pi) {
                for (int32_t nth = 0; nth <= ins[pi]; ++nth, ++c) {
                    while (next_read < n && (last_col[next_read] < first_col[next_read] || first_col[next_read] <= c)) {
                        if (last_col[next_read] >= first_col[next_read] && !reads[next_read].dead) active.push_back(next_read);
                        ++next_read;
                    }
                    if (active.empty()) continue;
                    head = tail = nullptr;
                    int depth = 0;
                    for (size_t k : active) {
                        sta_pileup_t *p = reads[k].p;
                        const uint64_t at = entry_off[k] + (uint64_t)(c - first_col[k]);
                        const uint32_t e = entries[at];
                        const int b4 = (int)(e & 31u);
                        p->base4 = (e & 0x8000u) ? 0 : b4;
                        p->base = (e & 0x8000u) ? '.' : b4 >= 16 ? '*' : "NACMGRSVTWYHKDBN"[b4];
                        p->qual = (int)((e >> 5) & 255u);
                        p->ref_skip = (e & 0x2000u) ? 1 : 0;
                        p->padding = (e & 0x10000u) ? 1 : 0;
                        p->seq_offset = (int)seq_offs[at];
                        p->pos = cursor + (int64_t)pi + 1; p->nth = nth;
                        p->next = nullptr;
                        if (tail) tail->next = p; else head = p;
                        tail = p;
                        ++depth;
                    }
                    const int v = seq_column(cd, fp, h, head, depth, (hts_pos_t)(cursor + (int64_t)pi + 1), nth, ins[pi] - nth);
                    size_t keep = 0;
                    for (size_t k : active) {
                        // (a read that continues beyond the window also has the window's last column as its last_col)
                        if (last_col[k] == c && reads[k].end <= we) { reads[k].dead = true; if (seq_free) seq_free(cd, fp, h, reads[k].p); }
                        else active[keep++] = k;
                    }
                    active.resize(keep);
                    if (v == 1) { ret = 0; goto done; }
                    if (v != 0) goto done;
                }
            }
            cursor = we;
        }
        // ---- what stays for the next window: reads that reach `cursor` (alive there, or ended on the column before it) ----
        {
            size_t keep = 0;
            for (Live &l : reads) {
                if (l.end >= cursor) { reads[keep++] = l; continue; }
                if (!l.dead && seq_free) seq_free(cd, fp, h, l.p);      // (cannot happen: a read's last column lies in the window that covers it)
                recycle(l.p);
            }
            reads.resize(keep);
        }
        if (at_eof && !ahead && reads.empty()) break;
    }
    ret = 0;
done:
    for (Live &l : reads) { if (!l.dead && seq_free) seq_free(cd, fp, h, l.p); recycle(l.p); }
    if (ahead) { if (seq_free) seq_free(cd, fp, h, ahead); recycle(ahead); }
    for (sta_pileup_t *p : pool) { free(p->b.data); free(p); }
    sta_engine_destroy(eng);
    return ret;
}
