// plp_api.cpp -- bam_plp_* / bam_mplp_* / bam_plbuf_* over the MI355X engine
// (include/samtools_amd_plp.h; replaces HTSlib sam.c's pileup iterator as called from
// bam_plcmd.c:581-607, bam_plbuf.c:40-69, bedcov.c:303-335, coverage.c:572-589).
//
// The iterator reads ahead through the caller's callback, keeps its own copies of the records
// (bam_copy1 semantics), stages a window of them into HBM, has the device resolve every
// (read, column) pair, and hands the columns out one by one.  No CPU pileup exists here: without a
// HIP device the iterators fail.
#include "../../include/samtools_amd.h"
#include "../../include/samtools_amd_plp.h"
#include "host_stage.h"
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>
#include <mutex>
#include <new>
#include <string>
#include <algorithm>
#include "host_names.h"

namespace {

enum { OP_M = 0, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X };

struct LiveRead {
    bam1_t b;
    int64_t end = 0;                 // bam_endpos: pos + max(reference span, 1)
    bool constructed = false;
    bool accepted = false;           // carried from an earlier window whose -d replay kept it
    bool cap_dropped = false;        // removed by the -d cap in the window just processed
    bool ghost = false;              // left the pileup (destructor done) but still staged: its partner in the overlap hash is live
    // the overlap hash, kept in push order by the iterator itself (host_names.h): has this record been there; the record whose entry it
    // found (-1: none); for both partners of such a pair the other one's end
    bool hashed = false; int64_t mate_id = -1, mate_end = INT64_MIN;
    bam_pileup_cd cd;
    std::vector<uint8_t> orig_qual;  // qualities as pushed (mate-overlap resolution restarts from these in every window)
};

int64_t ref_span(const bam1_t *b)
{
    const uint32_t *cig = bam_get_cigar(b);
    int64_t l = 0;
    for (uint32_t k = 0; k < b->core.n_cigar; ++k) {
        int op = cig[k] & 0xf;
        if (op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X) l += cig[k] >> 4;
    }
    return l;
}

LiveRead *copy_read(const bam1_t *b)
{
    LiveRead *r = new LiveRead();
    r->b = *b;
    r->b.data = (uint8_t *)malloc(b->l_data > 0 ? (size_t)b->l_data : 1);
    if (!r->b.data) { delete r; return nullptr; }
    if (b->l_data > 0) memcpy(r->b.data, b->data, (size_t)b->l_data);
    r->b.m_data = (uint32_t)(b->l_data > 0 ? b->l_data : 1);
    r->b.mempolicy = 0;
    int64_t l = ref_span(b);
    r->end = b->core.pos + (l > 0 ? l : 1);
    r->cd.i = 0;
    return r;
}

void free_read(LiveRead *r)
{
    if (!r) return;
    free(r->b.data);
    delete r;
}

// structure-of-arrays staging of the live reads (the sta_reads layout of samtools_amd.h)
struct Soa {
    std::vector<int32_t> pos, l_qseq, mtid, isize;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq, aux, seq, qual;
    std::vector<uint32_t> cig_off, base_off8, name_off, cigar;
    std::vector<int64_t> mpos;
    std::vector<char> names;
    void clear()
    {
        pos.clear(); l_qseq.clear(); mtid.clear(); isize.clear(); flag.clear(); mapq.clear(); aux.clear(); seq.clear();
        qual.clear(); cig_off.clear(); base_off8.clear(); name_off.clear(); cigar.clear(); mpos.clear(); names.clear();
    }
    void add(const LiveRead &r, int64_t origin)
    {
        const bam1_t *b = &r.b;
        pos.push_back((int32_t)(b->core.pos - origin));
        flag.push_back(b->core.flag);
        mapq.push_back(b->core.qual);
        aux.push_back(r.accepted ? STA_AUX_ACCEPTED : 0);
        int32_t lq = b->core.l_qseq;
        l_qseq.push_back(lq);
        cig_off.push_back((uint32_t)cigar.size());
        const uint32_t *cg = bam_get_cigar(b);
        cigar.insert(cigar.end(), cg, cg + b->core.n_cigar);
        size_t b0 = qual.size();
        base_off8.push_back((uint32_t)(b0 >> 3));
        size_t padded = ((size_t)lq + 7) & ~(size_t)7;
        qual.resize(b0 + padded, 0);
        const uint8_t *q = r.orig_qual.empty() ? bam_get_qual(b) : r.orig_qual.data();
        if (lq) memcpy(&qual[b0], q, (size_t)lq);
        seq.resize((b0 + padded) / 2, 0);
        if (lq) memcpy(&seq[b0 / 2], bam_get_seq(b), ((size_t)lq + 1) / 2);
        mtid.push_back(b->core.mtid);
        mpos.push_back(b->core.mpos);
        int64_t is = b->core.isize;
        if (is > INT32_MAX) is = INT32_MAX;
        if (is < -INT32_MAX) is = -INT32_MAX;
        isize.push_back((int32_t)is);
        name_off.push_back((uint32_t)names.size());
        const char *qn = bam_get_qname(b);
        size_t ql = strnlen(qn, b->core.l_qname);
        names.insert(names.end(), qn, qn + ql);
        names.push_back('\0');
    }
    sta_reads view()
    {
        cig_off.push_back((uint32_t)cigar.size());
        name_off.push_back((uint32_t)names.size());
        sta_reads v;
        memset(&v, 0, sizeof v);
        v.n_reads = (int64_t)pos.size();
        v.pos = pos.data(); v.flag = flag.data(); v.mapq = mapq.data(); v.aux = aux.data(); v.l_qseq = l_qseq.data();
        v.cig_off = cig_off.data(); v.base_off8 = base_off8.data(); v.mtid = mtid.data(); v.mpos = mpos.data();
        v.isize = isize.data(); v.name_off = name_off.data(); v.cigar = cigar.data(); v.seq = seq.data(); v.qual = qual.data();
        v.bq = nullptr; v.names = names.data();
        v.n_cigar_total = cigar.size(); v.n_bases_total = qual.size(); v.n_name_bytes = names.size();
        return v;
    }
};

const int64_t MAX_WINDOW_COLS = (int64_t)1 << 24;

// Engines (HBM workspace, streams) are expensive to create; callers like bedcov create one iterator per BED interval,
// so finished iterators park their engine here.  An iterator is single-threaded by contract, like the HTSlib one;
// different iterators may live on different threads, hence the lock around the pool.
std::vector<sta_engine *> g_engine_pool;
std::mutex g_engine_pool_m;                    // iterators may live on different threads
enum { ST_OK = 0, ST_NEED_MORE = 1, ST_END = 2, ST_ERR = -1 };

}  // namespace

struct sta_bam_plp {
    sta_bam_plp_auto_f func = nullptr;
    void *data = nullptr;
    sta_engine *eng = nullptr;
    int maxcnt = 8000;
    bool overlaps = false;
    int (*ctor)(void *, const bam1_t *, bam_pileup_cd *) = nullptr;
    int (*dtor)(void *, const bam1_t *, bam_pileup_cd *) = nullptr;
    int batch = 65536;
    std::deque<LiveRead *> live;       // reads of the current / next window, position sorted (carried reads first)
    std::deque<LiveRead *> pending;    // pushed (bam_plp_push) but not yet taken into a window
    LiveRead *peek = nullptr;          // first record beyond the window being built
    size_t n_new = 0;                  // records added to `live` since the last window
    bool eof = false;
    int error = 0;
    uint64_t next_id = 0;
    sta::OverlapNames onames;          // HTSlib's overlap hash (overlap_push / overlap_remove) over everything pushed so far
    int max_tid = -1; int64_t max_pos = -1;
    // current window
    bool have_win = false;
    int win_tid = -1; int64_t cb = 0, ce = 0, cur = 0;
    int prev_tid = -1; int64_t prev_ce = -1;
    std::vector<uint64_t> offs;
    std::vector<sta_plp_entry> ent;
    std::vector<uint32_t> info;
    std::vector<uint8_t> qpool;
    std::vector<LiveRead *> win_reads;
    std::vector<bam_pileup1_t> plp;
    // mate-overlap resolution becomes visible in b->qual[] at the column HTSlib would have applied it (push order)
    struct Pending { int64_t vis_col; LiveRead *r; int32_t y; uint8_t q_new; };
    std::vector<Pending> pend_vis; size_t pend_next = 0;
    std::vector<int32_t> fix_y, fix_mate; std::vector<uint8_t> fix_q;
    Soa soa;
    bam1_t tmp;                        // caller-owned record the callback fills

    sta_bam_plp() { memset(&tmp, 0, sizeof tmp); }
};

namespace {

// bam_plp_push admission (HTSlib sam.c): unmapped / unplaced records are ignored, input must be sorted
int admit(sta_bam_plp *it, const bam1_t *b, LiveRead **out)
{
    *out = nullptr;
    if (b->core.tid < 0 || (b->core.flag & 4)) return 0;
    if (b->core.tid < it->max_tid) { fprintf(stderr, "[E::bam_plp_push] The input is not sorted (chromosomes out of order)\n"); it->error = 1; return -1; }
    if (b->core.tid == it->max_tid && b->core.pos < it->max_pos) { fprintf(stderr, "[E::bam_plp_push] The input is not sorted (reads out of order)\n"); it->error = 1; return -1; }
    it->max_tid = b->core.tid; it->max_pos = b->core.pos;
    LiveRead *r = copy_read(b);
    if (!r) { it->error = 1; return -1; }
    r->b.id = it->next_id++;
    *out = r;
    return 0;
}

// next admitted record: from the push queue, or pulled through the callback
int fetch_one(sta_bam_plp *it, LiveRead **out)
{
    *out = nullptr;
    for (;;) {
        if (!it->pending.empty()) { *out = it->pending.front(); it->pending.pop_front(); return ST_OK; }
        if (it->eof) return ST_END;
        if (!it->func) return ST_NEED_MORE;
        int ret = it->func(it->data, &it->tmp);
        if (ret == -1) { it->eof = true; return ST_END; }
        if (ret < -1) { it->error = 1; return ST_ERR; }
        LiveRead *r = nullptr;
        if (admit(it, &it->tmp, &r) < 0) return ST_ERR;
        if (r) { *out = r; return ST_OK; }
    }
}

void retire(sta_bam_plp *it)
{
    // reads that cannot reach a column >= ce leave the iterator (destructor hook, like bam_plp_next's mp_free)
    std::deque<LiveRead *> keep;
    // Overlap resolution is re-derived from the pushed qualities in every window, and tweak_overlap_quality rewrites both partners from
    // both: a read whose partner in the overlap hash (found by the iterator's own hash, in push order) can still touch a column is kept
    // staged as a "ghost" -- destructor hook fired, no columns -- until the partner leaves too.
    auto mate_stays = [&](const LiveRead *r) { return it->overlaps && r->mate_end > it->ce; };
    for (LiveRead *r : it->live) {
        if (r->end <= it->ce || r->cap_dropped) {
            if (r->constructed && it->dtor) it->dtor(it->data, &r->b, &r->cd);
            r->constructed = false;
            if (!r->cap_dropped && mate_stays(r)) { r->ghost = true; r->accepted = true; keep.push_back(r); }
            else free_read(r);
        } else { r->accepted = true; keep.push_back(r); }
    }
    it->live.swap(keep);
    it->prev_tid = it->win_tid; it->prev_ce = it->ce;
    it->have_win = false;
    it->n_new = 0;
}

int build_window(sta_bam_plp *it)
{
    // gather: carried reads are already in `live`
    for (;;) {
        if (!it->peek) {
            LiveRead *r = nullptr;
            int st = fetch_one(it, &r);
            if (st == ST_ERR) return ST_ERR;
            if (st == ST_NEED_MORE) return ST_NEED_MORE;
            if (st == ST_END) break;
            it->peek = r;
        }
        int tid = it->live.empty() ? it->peek->b.core.tid : it->live.front()->b.core.tid;
        if (it->peek->b.core.tid != tid) break;
        if (it->n_new >= (size_t)it->batch && !it->live.empty() && it->peek->b.core.pos > it->live.back()->b.core.pos) break;
        it->live.push_back(it->peek); it->peek = nullptr; it->n_new++;
    }
    if (it->live.empty()) return ST_END;
    const int tid = it->live.front()->b.core.tid;
    const int64_t origin = it->live.front()->b.core.pos;
    int64_t cb = origin;
    if (it->prev_tid == tid && it->prev_ce > cb) cb = it->prev_ce;
    int64_t ce;
    if (it->peek && it->peek->b.core.tid == tid) ce = it->peek->b.core.pos;
    else { ce = cb; for (LiveRead *r : it->live) ce = std::max(ce, r->end); }
    if (ce - cb > MAX_WINDOW_COLS) ce = cb + MAX_WINDOW_COLS;
    if (ce - origin > (int64_t)INT32_MAX - 1) ce = origin + INT32_MAX - 1;
    it->win_tid = tid; it->cb = cb; it->ce = ce; it->cur = 0;
    it->offs.assign(1, 0); it->ent.clear();
    it->have_win = true;
    if (ce <= cb) return ST_OK;

    if (!it->eng) { std::lock_guard<std::mutex> g(g_engine_pool_m); if (!g_engine_pool.empty()) { it->eng = g_engine_pool.back(); g_engine_pool.pop_back(); } }
    if (!it->eng) {
        if (sta_engine_create(&it->eng, 0, nullptr) != STA_OK) {
            fprintf(stderr, "[E::bam_plp] no usable HIP device (the MI355X engine has no CPU fallback)\n");
            it->error = 1; return ST_ERR;
        }
    }
    it->soa.clear();
    it->win_reads.assign(it->live.begin(), it->live.end());
    // Mate overlaps: HTSlib resolves a pair when its second mate is pushed, and the record that bounds this window (peek)
    // is the push that releases the window's last columns -- if it is the mate of a staged read, those columns already
    // see the resolved quality.  So peek is staged too (it has no column here) whenever it can overlap a staged read.
    LiveRead *lookahead = nullptr;
    if (it->overlaps && it->peek && it->peek->b.core.tid == tid) {
        int64_t max_end = INT64_MIN;
        for (LiveRead *r : it->live) max_end = std::max(max_end, r->end);
        if (it->peek->b.core.pos < max_end) { lookahead = it->peek; it->win_reads.push_back(lookahead); }
    }
    // reads starting at or beyond ce have no column here; they are staged anyway (they are few) to keep indices simple
    for (LiveRead *r : it->win_reads) {
        if (it->overlaps && r->orig_qual.empty() && r->b.core.l_qseq > 0)
            r->orig_qual.assign(bam_get_qual(&r->b), bam_get_qual(&r->b) + r->b.core.l_qseq);
        it->soa.add(*r, origin);
    }
    sta_reads rv = it->soa.view();
    sta_window w; memset(&w, 0, sizeof w);
    w.tid = tid; w.origin = origin; w.col_beg = (int32_t)(cb - origin); w.col_end = (int32_t)(ce - origin);
    w.tname = ""; w.tlen = INT64_MAX; w.n_files = 1; w.files = &rv; w.mem = STA_MEM_HOST;
    sta_plan_info pi;
    // The overlap hash is the iterator's own (host_names.h), in push order; who the -d cap turns away at the push is the plan's to say, so
    // the plan stops in front of its overlap pass and asks (sta_set_mate_resolver): the window's new reads visit the hash, every read gets
    // the staged index of the record whose entry it found
    if (it->overlaps)
        sta_set_mate_resolver(it->eng, [](void *user, int32_t, const uint32_t *state, int64_t n, int32_t *mate_out) {
            sta_bam_plp *p = (sta_bam_plp *)user;
            const std::vector<LiveRead *> &wr = p->win_reads;
            for (int64_t i = 0; i < n && (size_t)i < wr.size(); ++i) {
                LiveRead *r = wr[(size_t)i];
                mate_out[i] = -1;
                if (!r->hashed) {
                    r->hashed = true;
                    const bam1_t *b = &r->b;
                    const int64_t span = ref_span(b);
                    const bool pushed = (state[i] & 1u) != 0, dropped = pushed && !(state[i] & 2u) && span > 0;
                    if (pushed) {
                        sta::OverlapNames::Read q;
                        q.qname = bam_get_qname(b); q.l_qname = (uint32_t)strnlen(q.qname, b->core.l_qname); q.h = sta::qname_hash64(q.qname, q.l_qname);
                        q.flag = b->core.flag; q.tid = b->core.tid; q.mtid = b->core.mtid; q.l_qseq = b->core.l_qseq;
                        q.pos = b->core.pos; q.end = b->core.pos + span; q.mpos = b->core.mpos; q.isize = b->core.isize; q.id = (int64_t)b->id;
                        r->mate_id = p->onames.push(q, dropped);
                    }
                }
                if (r->mate_id < 0) continue;
                // (win_reads is in push order: ids ascend)
                auto h = std::lower_bound(wr.begin(), wr.begin() + i, r->mate_id, [](const LiveRead *x, int64_t v) { return (int64_t)x->b.id < v; });
                if (h != wr.begin() + i && (int64_t)(*h)->b.id == r->mate_id) {
                    mate_out[i] = (int32_t)(h - wr.begin());
                    r->mate_end = (*h)->end; (*h)->mate_end = r->end;
                }
            }
            return 0;
        }, it);
    const bool plan_ok = sta_stage_window(it->eng, &w) == STA_OK && sta_plp_plan(it->eng, it->maxcnt, it->overlaps ? 1 : 0, &pi) == STA_OK;
    sta_set_mate_resolver(it->eng, nullptr, nullptr);
    if (!plan_ok || sta_plp_emit(it->eng, nullptr, 0) != STA_OK) {
        fprintf(stderr, "[E::bam_plp] %s\n", sta_last_error(it->eng));
        it->error = 1; return ST_ERR;
    }
    const uint64_t ncols = (uint64_t)(ce - cb), n_ent = pi.out_bytes / 16;
    it->offs.resize(ncols + 1);
    it->ent.resize(n_ent);
    it->info.resize(it->win_reads.size());
    if (it->overlaps) it->qpool.resize(it->soa.qual.size());
    if (sta_fetch_col_offsets(it->eng, it->offs.data(), ncols + 1) != STA_OK
        || sta_fetch_output(it->eng, (char *)it->ent.data(), n_ent * 16) != STA_OK
        || sta_fetch_read_state(it->eng, 0, it->info.data(), it->overlaps ? it->qpool.data() : nullptr) != STA_OK) {
        fprintf(stderr, "[E::bam_plp] %s\n", sta_last_error(it->eng));
        it->error = 1; return ST_ERR;
    }
    it->pend_vis.clear(); it->pend_next = 0;
    if (it->overlaps) {
        const size_t nr = it->win_reads.size();
        it->fix_y.assign(nr, -1); it->fix_mate.assign(nr, 0); it->fix_q.assign(nr, 0);
        if (sta_fetch_overlap_fixups(it->eng, 0, it->fix_y.data(), it->fix_mate.data(), it->fix_q.data()) != STA_OK) {
            fprintf(stderr, "[E::bam_plp] %s\n", sta_last_error(it->eng));
            it->error = 1; return ST_ERR;
        }
    }
    for (size_t i = 0; i < it->win_reads.size(); ++i) {
        LiveRead *r = it->win_reads[i];
        if (it->overlaps && r->b.core.l_qseq > 0) {
            uint8_t *q = bam_get_qual(&r->b);
            memcpy(q, &it->qpool[(size_t)it->soa.base_off8[i] << 3], (size_t)r->b.core.l_qseq);
            const int32_t y = it->fix_y[i];
            if (y >= 0 && y < r->b.core.l_qseq && q[y] != it->fix_q[i]) {
                // the pair was resolved when the mate was pushed; columns handed out before that still show the old value.
                // They are the columns below the start of the last read that advanced the iterator before the mate.
                const size_t m = (size_t)it->fix_mate[i];
                int64_t vis = it->win_reads[m]->b.core.pos;
                for (size_t j = m; j-- > 0;) {
                    const uint32_t inf = it->info[j];
                    const LiveRead *rj = it->win_reads[j];
                    const bool dropped = (inf & 1u) && !(inf & 2u) && ref_span(&rj->b) > 0;      // pushed, but removed by the -d cap
                    if ((inf & 1u) && !dropped) { vis = rj->b.core.pos; break; }
                }
                it->pend_vis.push_back(sta_bam_plp::Pending{ vis, r, y, q[y] });
                q[y] = it->fix_q[i];
            }
        }
        r->cap_dropped = (it->info[i] & 1u) && !(it->info[i] & 2u) && ref_span(&r->b) > 0;      // bam_plp_push never stored it
        if ((it->info[i] & 2u) && !r->constructed && !r->ghost) {          // read entered the pileup: constructor hook
            r->constructed = true;
            if (it->ctor) it->ctor(it->data, &r->b, &r->cd);
        }
    }
    if (lookahead) lookahead->cap_dropped = false;       // it is a new read of the next window, tested there
    std::sort(it->pend_vis.begin(), it->pend_vis.end(), [](const sta_bam_plp::Pending &a, const sta_bam_plp::Pending &b) { return a.vis_col < b.vis_col; });
    return ST_OK;
}

const bam_pileup1_t *next64(sta_bam_plp *it, int *tid, hts_pos_t *pos, int *n_plp)
{
    if (!it || it->error) { if (n_plp) *n_plp = -1; return nullptr; }
    for (;;) {
        if (it->have_win) {
            const int64_t ncols = (int64_t)it->offs.size() - 1;
            while (it->cur < ncols) {
                const int64_t c = it->cur++;
                const uint64_t a = it->offs[(size_t)c], b = it->offs[(size_t)c + 1];
                if (b == a) continue;
                while (it->pend_next < it->pend_vis.size() && it->pend_vis[it->pend_next].vis_col <= it->cb + c) {
                    const auto &pv = it->pend_vis[it->pend_next++];
                    bam_get_qual(&pv.r->b)[pv.y] = pv.q_new;
                }
                it->plp.resize((size_t)(b - a));
                for (uint64_t k = a; k < b; ++k) {
                    const sta_plp_entry &e = it->ent[(size_t)k];
                    bam_pileup1_t &p = it->plp[(size_t)(k - a)];
                    memset(&p, 0, sizeof p);
                    LiveRead *r = it->win_reads[(size_t)e.read];
                    p.b = &r->b; p.qpos = e.qpos; p.indel = e.indel; p.level = 0;
                    p.is_del = e.bits & 1; p.is_head = (e.bits >> 1) & 1; p.is_tail = (e.bits >> 2) & 1; p.is_refskip = (e.bits >> 3) & 1;
                    p.aux = 0; p.cd = r->cd; p.cigar_ind = (int)(e.bits >> 4);
                }
                *tid = it->win_tid; *pos = it->cb + c; *n_plp = (int)(b - a);
                return it->plp.data();
            }
            // cd values may have been updated by the caller through plp[].cd?  HTSlib hands out copies too.
            retire(it);
        }
        int st = build_window(it);
        if (st == ST_ERR) { *n_plp = -1; return nullptr; }
        if (st == ST_NEED_MORE || st == ST_END) { *n_plp = 0; return nullptr; }
    }
}

void clear_reads(sta_bam_plp *it)
{
    for (LiveRead *r : it->live) { if (r->constructed && it->dtor) it->dtor(it->data, &r->b, &r->cd); free_read(r); }
    for (LiveRead *r : it->pending) free_read(r);
    if (it->peek) free_read(it->peek);
    it->live.clear(); it->pending.clear(); it->peek = nullptr;
}

}  // namespace

extern "C" {

sta_bam_plp_t sta_bam_plp_init(sta_bam_plp_auto_f func, void *data)
{
    sta_bam_plp *it = new sta_bam_plp();
    it->func = func; it->data = data;
    if (const char *e = getenv("STA_PLP_BATCH")) { int v = atoi(e); if (v > 0) it->batch = v; }
    return it;
}

void sta_bam_plp_destroy(sta_bam_plp_t it)
{
    if (!it) return;
    clear_reads(it);
    free(it->tmp.data);
    if (it->eng) {
        // an engine that reported an error is not reused by a later iterator
        std::unique_lock<std::mutex> g(g_engine_pool_m);
        if (!it->error && g_engine_pool.size() < 8) g_engine_pool.push_back(it->eng);
        else sta_engine_destroy(it->eng);
    }
    delete it;
}

int sta_bam_plp_push(sta_bam_plp_t it, const bam1_t *b)
{
    if (!it || it->error) return -1;
    if (!b) { it->eof = true; return 0; }
    LiveRead *r = nullptr;
    if (admit(it, b, &r) < 0) return -1;
    if (r) it->pending.push_back(r);
    return 0;
}

const bam_pileup1_t *sta_bam_plp64_next(sta_bam_plp_t it, int *tid, hts_pos_t *pos, int *n_plp)
{
    // push style: columns are produced only from what has been pushed (never through the callback)
    if (!it) { if (n_plp) *n_plp = -1; return nullptr; }
    sta_bam_plp_auto_f f = it->func;
    it->func = nullptr;
    // (a window is built once a full batch plus the record that bounds it, or EOF, has been pushed)
    const bam_pileup1_t *p = next64(it, tid, pos, n_plp);
    it->func = f;
    return p;
}

const bam_pileup1_t *sta_bam_plp_next(sta_bam_plp_t it, int *tid, int *pos, int *n_plp)
{
    hts_pos_t p64 = 0;
    const bam_pileup1_t *p = sta_bam_plp64_next(it, tid, &p64, n_plp);
    if (p) *pos = p64 < INT_MAX ? (int)p64 : INT_MAX;
    return p;
}

const bam_pileup1_t *sta_bam_plp64_auto(sta_bam_plp_t it, int *tid, hts_pos_t *pos, int *n_plp)
{
    if (!it || !it->func) { if (n_plp) *n_plp = -1; return nullptr; }
    return next64(it, tid, pos, n_plp);
}

const bam_pileup1_t *sta_bam_plp_auto(sta_bam_plp_t it, int *tid, int *pos, int *n_plp)
{
    hts_pos_t p64 = 0;
    const bam_pileup1_t *p = sta_bam_plp64_auto(it, tid, &p64, n_plp);
    if (p) *pos = p64 < INT_MAX ? (int)p64 : INT_MAX;
    return p;
}

void sta_bam_plp_set_maxcnt(sta_bam_plp_t it, int maxcnt) { if (it) it->maxcnt = maxcnt; }
void sta_bam_plp_set_batch(sta_bam_plp_t it, int n) { if (it && n > 0) it->batch = n; }
int sta_bam_plp_init_overlaps(sta_bam_plp_t it) { if (!it) return -1; it->overlaps = true; return 0; }
void sta_bam_plp_constructor(sta_bam_plp_t it, int (*func)(void *, const bam1_t *, bam_pileup_cd *)) { if (it) it->ctor = func; }
void sta_bam_plp_destructor(sta_bam_plp_t it, int (*func)(void *, const bam1_t *, bam_pileup_cd *)) { if (it) it->dtor = func; }

void sta_bam_plp_reset(sta_bam_plp_t it)
{
    if (!it) return;
    clear_reads(it);
    it->eof = false; it->error = 0; it->max_tid = -1; it->max_pos = -1;
    it->onames = sta::OverlapNames();
    it->have_win = false; it->prev_tid = -1; it->prev_ce = -1; it->n_new = 0;
    it->offs.clear(); it->ent.clear();
}

int sta_bam_plp_insertion(const bam_pileup1_t *p, kstring_t *ins, int *del_len)
{
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    if (!p || !ins) return -1;
    auto reserve = [&](size_t n) -> int {
        if (ins->m < n) { char *t = (char *)realloc(ins->s, n); if (!t) return -1; ins->s = t; ins->m = n; }
        return 0;
    };
    if (p->indel <= 0) { if (reserve(1) < 0) return -1; ins->l = 0; ins->s[0] = '\0'; return 0; }
    if (del_len) *del_len = 0;
    const uint32_t *cigar = bam_get_cigar(p->b);
    const uint32_t nc = p->b->core.n_cigar;
    int indel = 0;
    for (uint32_t k = (uint32_t)p->cigar_ind + 1; k < nc; ++k) {
        int op = cigar[k] & 0xf;
        if (op == OP_P || op == OP_I) indel += (int)(cigar[k] >> 4); else break;
    }
    if (reserve((size_t)indel + 1) < 0) return -1;
    ins->l = (size_t)indel;
    indel = 0;
    int j = 1;
    for (uint32_t k = (uint32_t)p->cigar_ind + 1; k < nc; ++k) {
        int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
        if (op == OP_P) { for (int l = 0; l < len; ++l) ins->s[indel++] = '*'; }
        else if (op == OP_I) {
            for (int l = 0; l < len; ++l, ++j) {
                int qi = p->qpos + j - (int)p->is_del;
                ins->s[indel++] = qi < p->b->core.l_qseq ? nt16[bam_seqi(bam_get_seq(p->b), qi)] : 'N';
            }
        } else { if (op == OP_D && del_len) *del_len = len; break; }
    }
    ins->s[indel] = '\0';
    return indel;
}

// ---- base modifications through the drop-in names (bam_plcmd.c:86-109, :119, :356-369) ----
// HTSlib's sam_mods.c is absent from the reference tree; these are this library's own equivalents of the four calls mpileup makes
// (hts_base_mod_state_alloc / _free, bam_parse_basemod, bam_mods_at_qpos) plus bam_plp_insertion_mod with a live state, all on the MM / ML
// evaluation of host_mods.cpp that `samtools-amd mpileup --output-mods` itself uses (SAM tags specification 1.7).  The state is created
// per read by the iterator's constructor hook exactly as bam_plcmd.c:356-362 does it.
}  // extern "C"

struct hts_base_mod_state { std::vector<sta::ModHit> hits; bool parsed = false; };

namespace {
int ks_reserve1(kstring_t *ks, size_t n) { if (ks->m < n) { char *t = (char *)realloc(ks->s, n); if (!t) return -1; ks->s = t; ks->m = n; } return 0; }
// aux field `tag` of a record: pointer to its type byte, or NULL (BAM aux encoding, SAM specification 4.2.4).  Every field walked over --
// and the one returned -- is checked to lie inside the record (a Z / H value ends in a NUL before the end, a B array's n elements fit), as
// HTSlib's bam_aux_get does: a truncated or malformed aux block in a caller's bam1_t yields NULL, never a read behind b->data + l_data.
// *vend (if asked for) = the first byte behind the returned field's value.
const uint8_t *aux_find1(const bam1_t *b, const char *tag, const uint8_t **vend)
{
    if (b->core.l_qseq < 0 || b->l_data < 0) return nullptr;
    const uint8_t *p = bam_get_qual(b) + b->core.l_qseq, *end = b->data + b->l_data;
    if (p > end) return nullptr;
    while (end - p >= 3) {
        const bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
        const uint8_t *t = p + 2, *v = t + 1;
        size_t sz = 0;
        switch (*t) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': { const uint8_t *q = v; while (q < end && *q) ++q; if (q >= end) return nullptr; sz = (size_t)(q - v) + 1; break; }
        case 'B': {
            if (end - v < 5) return nullptr;
            const int sub = v[0]; uint32_t n; memcpy(&n, v + 1, 4);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!es || (size_t)n > ((size_t)(end - v) - 5) / es) return nullptr;
            sz = 5 + es * (size_t)n; break;
        }
        default: return nullptr;
        }
        if ((size_t)(end - v) < sz) return nullptr;
        if (hit) { if (vend) *vend = v + sz; return t; }
        p = v + sz;
    }
    return nullptr;
}
// MM before Mm, ML before Ml (HTSlib looks the upper-case tag up first)
const uint8_t *aux_find(const bam1_t *b, const char *t1, const char *t2, const uint8_t **vend = nullptr)
{
    const uint8_t *r = aux_find1(b, t1, vend);
    return r ? r : aux_find1(b, t2, vend);
}
}  // namespace

extern "C" {

hts_base_mod_state *sta_hts_base_mod_state_alloc(void) { return new (std::nothrow) hts_base_mod_state(); }
void sta_hts_base_mod_state_free(hts_base_mod_state *state) { delete state; }

// 0 on success (also when the record has no MM tag), -1 on a malformed MM / ML
int sta_bam_parse_basemod(const bam1_t *b, hts_base_mod_state *state)
{
    if (!b || !state) return -1;
    state->hits.clear(); state->parsed = true;
    const uint8_t *mm = aux_find(b, "MM", "Mm");
    if (!mm) return 0;
    if (*mm != 'Z') return -1;
    const uint8_t *ml = aux_find(b, "ML", "Ml");
    const uint8_t *mlv = nullptr; size_t n_ml = 0;
    if (ml) {
        if (ml[0] != 'B' || (ml[1] != 'C' && ml[1] != 'c')) return -1;
        uint32_t n; memcpy(&n, ml + 2, 4); n_ml = n; mlv = ml + 6;          // (aux_find checked that the n bytes lie inside the record)
    }
    return sta::parse_base_mods(bam_get_seq(b), b->core.l_qseq, (b->core.flag & 16) != 0, (const char *)mm + 1, mlv, n_ml, ml != nullptr, state->hits) ? 0 : -1;
}

// the modifications of query position qpos: fills up to n_mods entries, returns how many there are (possibly more than n_mods), 0 for none
int sta_bam_mods_at_qpos(const bam1_t *b, int qpos, hts_base_mod_state *state, hts_base_mod *mods, int n_mods)
{
    (void)b;
    if (!state || qpos < 0) return -1;
    auto lo = std::lower_bound(state->hits.begin(), state->hits.end(), (uint32_t)qpos, [](const sta::ModHit &h, uint32_t q) { return h.qpos < q; });
    int n = 0;
    for (auto it = lo; it != state->hits.end() && it->qpos == (uint32_t)qpos; ++it, ++n)
        if (mods && n < n_mods) { mods[n].modified_base = it->code; mods[n].canonical_base = it->canonical; mods[n].strand = it->strand; mods[n].qual = it->qual; }
    return n;
}

// bam_plcmd.c:119 calls the _mod form.  m == NULL (no --output-mods, or --no-output-ins-mods) is bam_plp_insertion; with a state every
// inserted base is followed by the "[...]" text of its modifications, as pileup_seq prints them behind an aligned base (:86-109).
// A state that was never handed to bam_parse_basemod is refused (< 0) instead of being read as "no modifications".
int sta_bam_plp_insertion_mod(const bam_pileup1_t *p, hts_base_mod_state *m, kstring_t *ins, int *del_len)
{
    if (!m) return sta_bam_plp_insertion(p, ins, del_len);
    if (!m->parsed) { fprintf(stderr, "[sta_bam_plp_insertion_mod] the modification state was not filled by bam_parse_basemod\n"); return -1; }
    if (!p || !ins) return -1;
    if (del_len) *del_len = 0;
    ins->l = 0;
    if (p->indel <= 0) { if (ks_reserve1(ins, 1) < 0) return -1; ins->s[0] = '\0'; return 0; }
    const uint32_t *cigar = bam_get_cigar(p->b);
    const uint32_t nc = p->b->core.n_cigar;
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    std::string out;
    int indel = 0, j = 1;
    for (uint32_t k = (uint32_t)p->cigar_ind + 1; k < nc; ++k) {
        const int op = (int)(cigar[k] & 0xf), len = (int)(cigar[k] >> 4);
        if (op == 6) { out.append((size_t)len, '*'); indel += len; }                      // P
        else if (op == 1) {                                                              // I
            for (int l = 0; l < len; ++l, ++j, ++indel) {
                const int qi = p->qpos + j - (int)p->is_del;
                out.push_back(qi < p->b->core.l_qseq ? nt16[bam_seqi(bam_get_seq(p->b), qi)] : 'N');
                auto lo = std::lower_bound(m->hits.begin(), m->hits.end(), (uint32_t)qi, [](const sta::ModHit &h, uint32_t q) { return h.qpos < q; });
                size_t n = 0;
                while (lo + (long)n != m->hits.end() && (lo + (long)n)->qpos == (uint32_t)qi) ++n;
                if (n) sta::append_mod_text(&*lo, n, out);
            }
        } else { if (op == 2 && del_len) *del_len = len; break; }                         // D ends the run
    }
    if (ks_reserve1(ins, out.size() + 1) < 0) return -1;
    memcpy(ins->s, out.data(), out.size());
    ins->s[out.size()] = '\0';
    ins->l = out.size();
    return indel;
}

}  // extern "C"

// ---- multi-file iterator ----
struct sta_bam_mplp {
    int n = 0;
    std::vector<sta_bam_plp *> iter;
    std::vector<int> tid, n_plp;
    std::vector<int64_t> pos;
    std::vector<const bam_pileup1_t *> plp;
    int min_tid = INT_MAX; int64_t min_pos = INT64_MAX;
};

extern "C" {

sta_bam_mplp_t sta_bam_mplp_init(int n, sta_bam_plp_auto_f func, void **data)
{
    sta_bam_mplp *m = new sta_bam_mplp();
    m->n = n;
    m->iter.resize((size_t)n); m->tid.assign((size_t)n, INT_MAX); m->pos.assign((size_t)n, INT64_MAX);
    m->n_plp.assign((size_t)n, 0); m->plp.assign((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) m->iter[(size_t)i] = sta_bam_plp_init(func, data[i]);
    return m;
}

void sta_bam_mplp_destroy(sta_bam_mplp_t m)
{
    if (!m) return;
    for (auto *it : m->iter) sta_bam_plp_destroy(it);
    delete m;
}

void sta_bam_mplp_set_maxcnt(sta_bam_mplp_t m, int maxcnt) { if (m) for (auto *it : m->iter) it->maxcnt = maxcnt; }
int sta_bam_mplp_init_overlaps(sta_bam_mplp_t m) { if (!m) return -1; for (auto *it : m->iter) it->overlaps = true; return 0; }
void sta_bam_mplp_constructor(sta_bam_mplp_t m, int (*func)(void *, const bam1_t *, bam_pileup_cd *)) { if (m) for (auto *it : m->iter) it->ctor = func; }
void sta_bam_mplp_destructor(sta_bam_mplp_t m, int (*func)(void *, const bam1_t *, bam_pileup_cd *)) { if (m) for (auto *it : m->iter) it->dtor = func; }

void sta_bam_mplp_reset(sta_bam_mplp_t m)
{
    if (!m) return;
    m->min_tid = INT_MAX; m->min_pos = INT64_MAX;
    for (int i = 0; i < m->n; ++i) {
        sta_bam_plp_reset(m->iter[(size_t)i]);
        m->tid[(size_t)i] = INT_MAX; m->pos[(size_t)i] = INT64_MAX; m->n_plp[(size_t)i] = 0; m->plp[(size_t)i] = nullptr;
    }
}

int sta_bam_mplp64_auto(sta_bam_mplp_t m, int *_tid, hts_pos_t *_pos, int *n_plp, const bam_pileup1_t **plp)
{
    if (!m) return -1;
    int new_min_tid = INT_MAX; int64_t new_min_pos = INT64_MAX;
    for (int i = 0; i < m->n; ++i) {
        size_t u = (size_t)i;
        if (m->pos[u] == m->min_pos && m->tid[u] == m->min_tid) {
            int t = 0; hts_pos_t p = 0;
            m->plp[u] = sta_bam_plp64_auto(m->iter[u], &t, &p, &m->n_plp[u]);
            if (m->n_plp[u] < 0) return -1;
            if (m->plp[u]) { m->tid[u] = t; m->pos[u] = p; }
            else { m->tid[u] = INT_MAX; m->pos[u] = INT64_MAX; }
        }
        if (m->plp[u]) {
            if (m->tid[u] < new_min_tid) { new_min_tid = m->tid[u]; new_min_pos = m->pos[u]; }
            else if (m->tid[u] == new_min_tid && m->pos[u] < new_min_pos) new_min_pos = m->pos[u];
        }
    }
    m->min_pos = new_min_pos; m->min_tid = new_min_tid;
    if (new_min_pos == INT64_MAX) return 0;
    *_tid = new_min_tid; *_pos = new_min_pos;
    int ret = 0;
    for (int i = 0; i < m->n; ++i) {
        size_t u = (size_t)i;
        if (m->pos[u] == m->min_pos && m->tid[u] == m->min_tid) { n_plp[i] = m->n_plp[u]; plp[i] = m->plp[u]; ++ret; }
        else { n_plp[i] = 0; plp[i] = nullptr; }
    }
    return ret;
}

int sta_bam_mplp_auto(sta_bam_mplp_t m, int *_tid, int *_pos, int *n_plp, const bam_pileup1_t **plp)
{
    hts_pos_t p64 = 0;
    int ret = sta_bam_mplp64_auto(m, _tid, &p64, n_plp, plp);
    if (ret > 0) *_pos = p64 < INT_MAX ? (int)p64 : INT_MAX;
    return ret;
}

}  // extern "C"

// ---- bam_plbuf (bam_plbuf.c:40-69) ----
struct sta_bam_plbuf {
    sta_bam_plp_t iter;
    sta_bam_pileup_f func;
    void *data;
};

extern "C" {

sta_bam_plbuf_t *sta_bam_plbuf_init(sta_bam_pileup_f func, void *data)
{
    sta_bam_plbuf *buf = new sta_bam_plbuf();
    buf->iter = sta_bam_plp_init(nullptr, nullptr);
    buf->func = func; buf->data = data;
    return buf;
}

void sta_bam_plbuf_destroy(sta_bam_plbuf_t *buf)
{
    if (!buf) return;
    sta_bam_plp_destroy(buf->iter);
    delete buf;
}

void sta_bam_plbuf_reset(sta_bam_plbuf_t *buf) { if (buf) sta_bam_plp_reset(buf->iter); }

int sta_bam_plbuf_push(const bam1_t *b, sta_bam_plbuf_t *buf)
{
    int ret, n_plp = 0, tid = 0;
    hts_pos_t pos = 0;
    const bam_pileup1_t *plp;
    ret = sta_bam_plp_push(buf->iter, b);
    if (ret < 0) return ret;
    while ((plp = sta_bam_plp64_next(buf->iter, &tid, &pos, &n_plp)) != 0)
        buf->func((uint32_t)tid, pos, n_plp, plp, buf->data);
    return n_plp < 0 ? -1 : 0;
}

}  // extern "C"
