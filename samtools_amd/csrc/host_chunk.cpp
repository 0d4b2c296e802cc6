// host_chunk.cpp -- see host_chunk.h
#include "host_chunk.h"
#include "host_inflate.h"
#include <chrono>
#include <algorithm>
#include <climits>
#include <cstring>

namespace sta {

// ------------------------------------------------------------------------------------------------ Chunk
void Chunk::append(const Rec &r)
{
    if (cig_off.empty()) { cig_off.push_back(0); base_off8.push_back(0); name_off.push_back(0); }
    tid.push_back(r.tid); pos.push_back(r.pos); flag.push_back(r.flag); mapq.push_back(r.mapq);
    l_qseq.push_back(r.l_qseq); mtid.push_back(r.mtid); mpos.push_back(r.mpos); isize.push_back(r.isize);
    rlen.push_back((int32_t)std::min<int64_t>(r.rlen, INT32_MAX));
    const bool bq_ok = r.has_bq && (int32_t)r.bq.size() >= r.l_qseq;
    aux.push_back((uint8_t)((bq_ok ? STA_AUX_HAS_BQ : 0) | (r.has_zq ? STA_AUX_HAS_ZQ : 0)));
    cigar.insert(cigar.end(), r.cigar.begin(), r.cigar.end());
    const size_t b0 = qual.size(), padded = ((size_t)r.l_qseq + 7) & ~(size_t)7;
    qual.resize(b0 + padded, 0);
    if (r.l_qseq) memcpy(&qual[b0], r.qual.data(), (size_t)r.l_qseq);
    seq.resize((b0 + padded) / 2, 0);
    if (r.l_qseq) memcpy(&seq[b0 / 2], r.seq.data(), ((size_t)r.l_qseq + 1) / 2);
    if (bq_ok && !has_bq_pool) { has_bq_pool = true; bq.assign(b0, 64); }
    if (has_bq_pool) { bq.resize(b0 + padded, 64); if (bq_ok) memcpy(&bq[b0], r.bq.data(), (size_t)r.l_qseq); }
    names.insert(names.end(), r.qname.begin(), r.qname.end());
    names.push_back('\0');
    name_h.push_back(qname_hash64(r.qname.data(), r.qname.size()));
    if (!r.tagtext.empty()) {
        if (tag_off.empty()) { n_tags = (int)r.tagtext.size(); tag_off.push_back(0); }
        for (size_t t = 0; t < r.tagtext.size(); ++t) {
            const bool has = t < r.tag_has.size() && r.tag_has[t];
            if (has) tag_text.insert(tag_text.end(), r.tagtext[t].begin(), r.tagtext[t].end());
            tag_has.push_back((char)has);
            tag_off.push_back((uint32_t)tag_text.size());
        }
    }
    cig_off.push_back((uint32_t)cigar.size());
    base_off8.push_back((uint32_t)(qual.size() >> 3));
    name_off.push_back((uint32_t)names.size());
}

void Chunk::reset()
{
    tid.clear(); l_qseq.clear(); mtid.clear(); rlen.clear(); pos.clear(); mpos.clear(); isize.clear(); flag.clear(); mapq.clear(); aux.clear();
    cig_off.clear(); base_off8.clear(); name_off.clear(); cigar.clear(); seq.clear(); qual.clear(); bq.clear(); has_bq_pool = false; names.clear();
    n_tags = 0; tag_off.clear(); tag_text.clear(); tag_has.clear();
    raw.reset(); rec_off.clear(); raw_ok = true;
    name_h.clear(); t_id.clear(); t_a.clear(); t_b.clear(); t_mode = 0;
}

void Chunk::close()
{
    if (cig_off.empty()) { cig_off.push_back(0); base_off8.push_back(0); name_off.push_back(0); }
}

void Chunk::to_rec(int64_t i, Rec &r) const
{
    const size_t k = (size_t)i;
    r = Rec();
    r.tid = tid[k]; r.mtid = mtid[k]; r.pos = pos[k]; r.mpos = mpos[k]; r.isize = isize[k];
    r.flag = flag[k]; r.mapq = mapq[k]; r.l_qseq = l_qseq[k]; r.rlen = rlen[k];
    r.qname.assign(names.data() + name_off[k]);
    r.name_h = name_h[k];
    if (t_id.size() == pos.size()) { r.id = t_id[k]; if (t_mode == 1) r.clip = t_a[k]; else r.mate_id = t_a[k]; r.mate_end = t_b[k]; }
    r.cigar.assign(cigar.begin() + cig_off[k], cigar.begin() + cig_off[k + 1]);
    const size_t b0 = (size_t)base_off8[k] << 3, l = (size_t)l_qseq[k];
    r.seq.assign(seq.begin() + (long)(b0 / 2), seq.begin() + (long)(b0 / 2 + (l + 1) / 2));
    r.qual.assign(qual.begin() + (long)b0, qual.begin() + (long)(b0 + l));
    r.has_bq = (aux[k] & STA_AUX_HAS_BQ) != 0; r.has_zq = (aux[k] & STA_AUX_HAS_ZQ) != 0;
    if (r.has_bq) r.bq.assign(bq.begin() + (long)b0, bq.begin() + (long)(b0 + l));
    if (n_tags > 0) {
        r.tagtext.assign((size_t)n_tags, std::string()); r.tag_has.assign((size_t)n_tags, 0);
        for (int t = 0; t < n_tags; ++t) {
            const size_t e = k * (size_t)n_tags + (size_t)t;
            r.tag_has[(size_t)t] = tag_has[e];
            r.tagtext[(size_t)t].assign(tag_text.data() + tag_off[e], tag_text.data() + tag_off[e + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ ChunkReader
static constexpr size_t GROUP_BYTES = 1 << 20;

static constexpr size_t GROUP_HEAD = 1 << 16;      // free bytes in front of a mapped group's data: room for the record carried in

ChunkReader::ChunkReader(AlnReader *rd, int threads, bool keep_raw, int gpu_device) : rd_(rd), keep_raw_(keep_raw)
{
    if (threads < 1) threads = 1;
    max_ahead_ = (size_t)threads * 2 + 2;
    // a BAM file on disk is read through a mapping (work_mapped); anything else -- SAM text, stdin, STA_CHUNK_MAP=0 -- through the
    // reader's byte stream (work)
    std::string path, err; uint64_t coff = 0, consumed = 0;
    const char *ev = getenv("STA_CHUNK_MAP");
    if (!(ev && atoi(ev) == 0) && rd->record_stream_position(&path, &coff, &consumed)) {
        map_ = BgzfMap::open(path, &err);
        if (map_) {
            cut_off_ = coff;
            Link l0; l0.skip = consumed;
            links_[0] = std::move(l0);
            rd->release_source();
        }
    }
    // the device decoder comes up on a thread of its own: until it is there the feeder hands single groups to the parsers, which inflate
    // them themselves -- the first windows do not wait for the HIP runtime
    const char *gi = getenv("STA_GPU_INFLATE"), *fk = getenv("STA_FAKE_GPU_INFLATE");
    if (map_ && gpu_device >= 0 && ((gi && atoi(gi) != 0) || (fk && atoi(fk) != 0))) {
        gpu_device_ = gpu_device;
        gpu_init_ = std::thread([this] {
            std::unique_ptr<GpuInflater> p = make_gpu_inflater(gpu_device_);
            const bool ok = p != nullptr;
            gpu_ = std::move(p);
            gpu_state_.store(ok ? 1 : -1, std::memory_order_release);
        });
        // (two batches of groups in flight on the device besides what the parsers hold)
        const char *e = getenv("STA_GPU_INFLATE_BATCH"); const int bg = e ? atoi(e) : 48;
        max_ahead_ = (size_t)(bg < 1 ? 1 : bg > 512 ? 512 : bg) * 2 + (size_t)threads * 2 + 2;
        th_.emplace_back([this] { work_gpu_feeder(); });
        for (int i = 0; i < threads; ++i) th_.emplace_back([this] { work_gpu_parse(); });
        return;
    }
    for (int i = 0; i < threads; ++i) th_.emplace_back([this] { if (map_) work_mapped(); else work(); });
}

ChunkReader::~ChunkReader()
{
    { std::lock_guard<std::mutex> g(out_m_); stop_ = true; }
    cv_room_.notify_all(); cv_out_.notify_all(); cv_link_.notify_all(); cv_ready_.notify_all();
    for (auto &t : th_) if (t.joinable()) t.join();
    if (gpu_init_.joinable()) gpu_init_.join();
}

void ChunkReader::publish_link(uint64_t seq, Link &&l)
{
    std::lock_guard<std::mutex> g(out_m_);
    links_[seq] = std::move(l);
    cv_link_.notify_all();
}

// the next group's blocks (under io_m_): true = a group (seq, blocks, total filled in); false = nothing more to cut, *end_status =
// 0 at the clean end of the file / of the region, -1 at a damaged block
bool ChunkReader::cut_group(MGroup &g, int *end_status)
{
    std::lock_guard<std::mutex> lk(io_m_);
    g.blocks.clear(); g.total = 0; g.bad = false; g.verify = false; g.keep.reset();
    *end_status = 0;
    if (io_end_.load()) return false;
    int st = 1;
    for (;;) {
        BgzfMap::Block b; uint64_t o = cut_off_;
        st = map_->block_at(&o, &b);
        if (st <= 0) break;
        if (g.total && g.total + b.isize > GROUP_BYTES) break;          // (stays for the next group)
        cut_off_ = o;
        if (b.isize) { g.blocks.push_back(b); g.total += b.isize; }
    }
    if (g.blocks.empty()) { *end_status = st < 0 ? -1 : 0; return false; }
    std::lock_guard<std::mutex> g2(out_m_);
    g.seq = next_in_++;
    return true;
}

// the end of the file (st == 0) or a damaged block (-1).  A clean end must not leave a record unfinished: what the last group handed
// on has to be empty
void ChunkReader::finish_stream(int st)
{
    std::unique_lock<std::mutex> lk(out_m_);
    if (io_end_.load()) return;                      // (the region's end was seen first: groups cut mid-record there are not an error)
    const uint64_t end_seq = next_in_;
    cv_link_.wait(lk, [&] { return stop_ || io_end_.load() || links_.count(end_seq) != 0; });
    if (stop_ || io_end_.load()) return;
    const Link &l = links_[end_seq];
    io_status_.store(st < 0 ? -1 : (l.bad || !l.carry.empty()) ? -2 : 0);
    io_end_.store(true);
    cv_out_.notify_all(); cv_link_.notify_all();
}

// one inflated group -> its chunk: the record carried in from the group in front, the walk over the block_size fields, the hand-over to
// the next group, the records
void ChunkReader::process_group(MGroup &g, pvector<uint8_t> &raw, Rec &r, std::string &scratch)
{
    const uint64_t seq = g.seq; const size_t total = g.total;
    bool bad = g.bad;
    if (g.verify && !bad) {
        // blocks that came back from the device: their CRC-32 is checked here; a mismatch (or a block the device gave up on, already
        // redone by the feeder) goes through the host decoder, whose verdict counts
        size_t off = GROUP_HEAD;
        for (const BgzfMap::Block &b : g.blocks) {
            if (fast_crc32(raw.data() + off, b.isize) != b.crc && !bgzf_inflate_block(b, raw.data() + off)) { bad = true; break; }
            off += b.isize;
        }
    }
    Link in;
    {
        std::unique_lock<std::mutex> lk(out_m_);
        cv_link_.wait(lk, [&] { return stop_ || links_.count(seq) != 0; });
        if (stop_) return;
        in = std::move(links_[seq]);
        links_.erase(seq);
    }
    auto c = new_chunk();
    Link out;
    size_t beg = GROUP_HEAD, end = GROUP_HEAD + total, q = end;       // records: raw[beg, q); raw[q, end) goes on to the next group
    bool records = false;
    if (bad || in.bad) { bad = true; out.bad = true; }
    else if (in.skip >= total) out.skip = in.skip - total;               // (still inside the header)
    else {
        beg += (size_t)in.skip;
        bool whole = true;                                                // the record carried in ends inside this group
        if (!in.carry.empty()) {
            const size_t cl = in.carry.size();
            uint8_t h[4];
            for (size_t k = 0; k < 4; ++k) h[k] = k < cl ? in.carry[k] : (k - cl < total ? raw[GROUP_HEAD + (k - cl)] : 0);
            int32_t bs = 0; memcpy(&bs, h, 4);
            if (cl + total < 4) whole = false;
            else if (bs < 32) bad = true;
            else if (4 + (size_t)bs - cl > total) whole = false;
            if (!bad) {
                if (!whole) { out.carry = std::move(in.carry); out.carry.insert(out.carry.end(), raw.begin() + (long)GROUP_HEAD, raw.begin() + (long)end); }
                else if (cl <= GROUP_HEAD) { memcpy(raw.data() + GROUP_HEAD - cl, in.carry.data(), cl); beg = GROUP_HEAD - cl; }
                else {
                    // a carry larger than the free head (a record of more than 64 KiB): the data moves up behind it
                    raw.resize(cl + total);
                    memmove(raw.data() + cl, raw.data() + GROUP_HEAD, total);
                    memcpy(raw.data(), in.carry.data(), cl);
                    beg = 0; end = cl + total;
                }
            }
        }
        if (!bad && whole) {
            // the walk over the block_size fields: where the last whole record ends
            q = beg;
            while (end - q >= 4) {
                int32_t bs; memcpy(&bs, &raw[q], 4);
                if (bs < 32) { bad = true; break; }
                if ((size_t)bs + 4 > end - q) break;
                q += (size_t)bs + 4;
            }
            if (!bad) { out.carry.assign(raw.begin() + (long)q, raw.begin() + (long)end); records = true; }
        }
        if (bad) { out = Link(); out.bad = true; }
    }
    publish_link(seq + 1, std::move(out));
    if (records) {
        size_t o = beg;
        while (o < q) {
            size_t used = 0;
            int st = rd_->parse_raw(raw.data() + o, q - o, &used, r, scratch);
            if (st <= 0) { bad = true; break; }
            o += used;
            r.rlen = 0;
            for (uint32_t cg : r.cigar) { int op = (int)(cg & 0xf); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r.rlen += (cg >> 4); }
            if (rd_->past_region(r) && !io_end_.load()) { std::lock_guard<std::mutex> g2(out_m_); io_status_.store(0); io_end_.store(true); cv_out_.notify_all(); cv_link_.notify_all(); }
            if (!rd_->in_region(r)) continue;
            c->append(r);
            if (keep_raw_ && g.keep) { c->rec_off.push_back((uint32_t)(o - used + 4)); if (r.cigar_from_tag) c->raw_ok = false; }
        }
    }
    c->close();
    if (g.keep && keep_raw_) { raw.resize(q); c->raw = std::move(g.keep); }    // (a window uploads up to the end of the last whole record)
    std::lock_guard<std::mutex> lk(out_m_);
    if (bad && seq < bad_seq_) bad_seq_ = seq;
    done_[seq] = std::move(c);
    cv_out_.notify_all();
}

void ChunkReader::work_mapped()
{
    pvector<uint8_t> own_raw;
    Rec r; std::string scratch;
    MGroup g;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(out_m_);
            cv_room_.wait(lk, [this] { return stop_ || next_in_ - next_out_ < max_ahead_; });
            if (stop_) return;
        }
        int end_status = 0;
        if (!cut_group(g, &end_status)) { finish_stream(end_status); return; }
        g.keep = keep_raw_ ? get_buf() : nullptr;
        pvector<uint8_t> &raw = g.keep ? *g.keep : own_raw;
        raw.resize(GROUP_HEAD + g.total);
        size_t off = GROUP_HEAD;
        for (const BgzfMap::Block &b : g.blocks) { if (!bgzf_inflate_block(b, raw.data() + off)) { g.bad = true; break; } off += b.isize; }
        process_group(g, raw, r, scratch);
    }
}

// ---- the same with the device's decoder (host_gpu_inflate.h): one feeder thread cuts groups by the batch, sends their blocks through
// the device and queues the inflated groups; the parser threads check the CRCs and parse ----
void ChunkReader::work_gpu_feeder()
{
    static const size_t batch_groups = [] { const char *e = getenv("STA_GPU_INFLATE_BATCH"); const int v = e ? atoi(e) : 48; return (size_t)(v < 1 ? 1 : v > 512 ? 512 : v); }();
    struct Batch { std::vector<MGroup> groups; std::vector<GpuInflateJob> jobs; int ticket = -1; };
    Batch bt[2];
    std::vector<uint32_t> status;
    int end_status = 0; bool ended = false;
    auto finalize = [&](Batch &b) {
        if (b.groups.empty()) return;
        bool dev_ok = b.ticket >= 0 && gpu_->wait(b.ticket, status) && status.size() == b.jobs.size();
        size_t j = 0;
        for (MGroup &g : b.groups) {
            size_t off = GROUP_HEAD;
            for (const BgzfMap::Block &blk : g.blocks) {
                // a block the device gave up on (or a whole batch it could not take) is inflated here; CRCs of the others: the parsers
                if ((!dev_ok || status[j] != 0) && !bgzf_inflate_block(blk, g.keep->data() + off)) g.bad = true;
                off += blk.isize; ++j;
            }
            g.verify = dev_ok;
        }
        {
            std::lock_guard<std::mutex> lk(out_m_);
            for (MGroup &g : b.groups) ready_q_.push_back(std::move(g));
        }
        cv_ready_.notify_all();
        b.groups.clear(); b.jobs.clear(); b.ticket = -1;
    };
    // leaving early (the reader is being torn down): the device may still be copying into the page-locked buffers of the batches in flight;
    // they must not go back to the pool (or be freed) before those copies have landed
    auto drain = [&] { for (Batch &b : bt) if (b.ticket >= 0 && gpu_) { gpu_->wait(b.ticket, status); b.ticket = -1; } };
    for (int cur = 0; !ended; cur ^= 1) {
        Batch &b = bt[cur];
        finalize(b);                                        // (the batch that used this slot two rounds ago)
        if (gpu_state_.load(std::memory_order_acquire) != 1) {
            // no device decoder (yet): one group at a time, inflated by the parser that takes it.  Groups must reach the parsers in cut
            // order, so the other slot's batch -- submitted when the decoder was there -- goes first
            finalize(bt[cur ^ 1]);
            {
                std::unique_lock<std::mutex> lk(out_m_);
                cv_room_.wait(lk, [this] { return stop_ || next_in_ - next_out_ < max_ahead_; });
                if (stop_) { drain(); return; }
            }
            MGroup g;
            if (!cut_group(g, &end_status)) { ended = true; break; }
            g.keep = get_buf();
            g.keep->resize(GROUP_HEAD + g.total);
            g.inflate_here = true;
            { std::lock_guard<std::mutex> lk(out_m_); ready_q_.push_back(std::move(g)); }
            cv_ready_.notify_one();
            continue;
        }
        while (b.groups.size() < batch_groups) {
            {
                std::unique_lock<std::mutex> lk(out_m_);
                // (a started batch goes out when the window of groups in flight is full: its groups are what the consumer waits for)
                if (next_in_ - next_out_ >= max_ahead_ && !b.groups.empty()) break;
                cv_room_.wait(lk, [this] { return stop_ || next_in_ - next_out_ < max_ahead_; });
                if (stop_) { drain(); return; }
            }
            MGroup g;
            if (!cut_group(g, &end_status)) { ended = true; break; }
            g.keep = get_buf();
            g.keep->resize(GROUP_HEAD + g.total);
            size_t off = GROUP_HEAD;
            for (const BgzfMap::Block &blk : g.blocks) { b.jobs.push_back(GpuInflateJob{ blk.comp, blk.clen, blk.isize, g.keep->data() + off }); off += blk.isize; }
            b.groups.push_back(std::move(g));
        }
        if (!b.groups.empty()) b.ticket = gpu_->submit(b.jobs.data(), b.jobs.size());
    }
    // the two batches still in flight, the older one first: groups must reach the parsers in cut order (a parser holding group k waits
    // for the hand-over from k - 1; were the later batch queued first, every parser could end up waiting for groups nobody is left to take)
    {
        const int older = bt[0].groups.empty() ? 1 : bt[1].groups.empty() ? 0 : (bt[0].groups.front().seq < bt[1].groups.front().seq ? 0 : 1);
        finalize(bt[older]); finalize(bt[older ^ 1]);
    }
    {
        std::lock_guard<std::mutex> lk(out_m_);
        feed_end_ = true;
    }
    cv_ready_.notify_all();
    finish_stream(end_status);
}

void ChunkReader::work_gpu_parse()
{
    Rec r; std::string scratch;
    for (;;) {
        MGroup g;
        {
            std::unique_lock<std::mutex> lk(out_m_);
            cv_ready_.wait(lk, [this] { return stop_ || !ready_q_.empty() || feed_end_; });
            if (stop_) return;
            if (ready_q_.empty()) return;                   // (the feeder is done)
            g = std::move(ready_q_.front()); ready_q_.pop_front();
        }
        if (g.inflate_here) {
            size_t off = GROUP_HEAD;
            for (const BgzfMap::Block &b : g.blocks) { if (!bgzf_inflate_block(b, g.keep->data() + off)) { g.bad = true; break; } off += b.isize; }
        }
        process_group(g, *g.keep, r, scratch);
    }
}

std::shared_ptr<Chunk> ChunkReader::new_chunk()
{
    std::unique_ptr<Chunk> c;
    {
        std::lock_guard<std::mutex> g(cpool_->m);
        if (!cpool_->free.empty()) { c = std::move(cpool_->free.back()); cpool_->free.pop_back(); }
    }
    if (!c) c.reset(new Chunk());
    std::shared_ptr<ChunkPool> pool = cpool_;       // (held by the deleter: windows may keep chunks beyond the reader's life)
    return std::shared_ptr<Chunk>(c.release(), [pool](Chunk *p) {
        p->reset();
        std::lock_guard<std::mutex> g(pool->m);
        if (pool->free.size() < 256) pool->free.emplace_back(p); else delete p;
    });
}

std::shared_ptr<pvector<uint8_t>> ChunkReader::get_buf()
{
    std::unique_ptr<pvector<uint8_t>> b;
    {
        std::lock_guard<std::mutex> g(pool_->m);
        if (!pool_->free.empty()) { b = std::move(pool_->free.back()); pool_->free.pop_back(); }
    }
    if (!b) { b.reset(new pvector<uint8_t>()); b->reserve(GROUP_BYTES + (GROUP_BYTES >> 3)); }
    std::shared_ptr<RawPool> pool = pool_;
    return std::shared_ptr<pvector<uint8_t>>(b.release(), [pool](pvector<uint8_t> *p) {
        std::lock_guard<std::mutex> g(pool->m);
        if (pool->free.size() < 256) pool->free.emplace_back(p); else delete p;
    });
}

void ChunkReader::work()
{
    pvector<uint8_t> own_raw;
    Rec r; std::string scratch;
    for (;;) {
        uint64_t seq;
        {
            // room first (bounded read-ahead), then the next group under the I/O lock: groups are cut strictly in file order
            std::unique_lock<std::mutex> lk(out_m_);
            cv_room_.wait(lk, [this] { return stop_ || next_in_ - next_out_ < max_ahead_; });
            if (stop_) return;
        }
        int64_t nrec = 0;
        std::shared_ptr<pvector<uint8_t>> keep = keep_raw_ ? get_buf() : nullptr;
        pvector<uint8_t> &raw = keep ? *keep : own_raw;
        {
            std::lock_guard<std::mutex> g(io_m_);
            if (io_end_.load()) return;
            int st = rd_->raw_group(raw, GROUP_BYTES, &nrec);
            if (st <= 0) {
                std::lock_guard<std::mutex> g2(out_m_);
                io_status_.store(st); io_end_.store(true);
                cv_out_.notify_all();
                return;
            }
            std::lock_guard<std::mutex> g2(out_m_);
            seq = next_in_++;
        }
        auto c = new_chunk();
        size_t o = 0; bool bad = false;
        while (o < raw.size()) {
            size_t used = 0;
            int st = rd_->parse_raw(raw.data() + o, raw.size() - o, &used, r, scratch);
            if (st <= 0) { bad = true; break; }
            o += used;
            r.rlen = 0;
            for (uint32_t cg : r.cigar) { int op = (int)(cg & 0xf); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r.rlen += (cg >> 4); }
            // (a record beyond the region of a sorted file: the groups behind this one are not cut any more; the ones already
            // in flight filter their records as before)
            if (rd_->past_region(r) && !io_end_.load()) { std::lock_guard<std::mutex> g2(out_m_); io_status_.store(0); io_end_.store(true); cv_out_.notify_all(); }
            if (!rd_->in_region(r)) continue;
            c->append(r);
            if (keep) { c->rec_off.push_back((uint32_t)(o - used + 4)); if (r.cigar_from_tag) c->raw_ok = false; }
        }
        c->close();
        if (keep) c->raw = std::move(keep);
        std::lock_guard<std::mutex> g(out_m_);
        if (bad && seq < bad_seq_) bad_seq_ = seq;
        done_[seq] = std::move(c);
        cv_out_.notify_all();
    }
}

std::shared_ptr<Chunk> ChunkReader::next()
{
    std::unique_lock<std::mutex> lk(out_m_);
    for (;;) {
        auto it = done_.find(next_out_);
        if (it != done_.end()) {
            std::shared_ptr<Chunk> c = std::move(it->second);
            done_.erase(it);
            const bool bad = next_out_ >= bad_seq_;
            ++next_out_;
            cv_room_.notify_all();
            if (bad) { status_ = -2; return nullptr; }       // records before the malformed one in that group are dropped too
            return c;
        }
        if (io_end_.load() && next_out_ >= next_in_) { const int st = io_status_.load(); status_ = st < 0 ? st : 0; return nullptr; }
        cv_out_.wait(lk);
    }
}

// ------------------------------------------------------------------------------------------------ ChunkPump
static double mono_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
ChunkPump::ChunkPump(std::vector<std::unique_ptr<AlnReader>> &readers, const PumpConfig &cfg, int threads) : cfg_(cfg)
{
    f_.resize(readers.size());
    // STA_STAGE_DEVICE: 1 (default for BAM input) = a window's new reads reach the device as raw alignment records and their pools are
    // built there (kernels_stage.hip); 0 = pools copied on the host (round-3 path); 2 = both, compared on the device (tests)
    raw_mode_ = cfg.device_pools ? 1 : 0;
    if (const char *ev = getenv("STA_STAGE_DEVICE")) if (cfg.device_pools) raw_mode_ = atoi(ev);
    if (cfg.xs_n_tags > 0) raw_mode_ = 0;                           // tag text columns are formatted per record on the host
    for (size_t i = 0; i < readers.size(); ++i) f_[i].rd.reset(new ChunkReader(readers[i].get(), threads, raw_mode_ != 0 && readers[i]->is_bam(), cfg.inflate_device));
    // threads that copy a window's chunk slices into the staging arrays (STA_STAGE_THREADS; 1 = the producer thread alone)
    stage_threads_ = threads >= 8 ? 4 : threads >= 4 ? 2 : 1;
    if (const char *ev = getenv("STA_STAGE_THREADS")) { const int v = atoi(ev); if (v >= 1 && v <= 64) { stage_threads_ = v; stage_min_bytes_ = 0; } }   // set explicitly: for every window, however small
}

bool ChunkPump::settle(File &f)
{
    for (;;) {
        if (f.eof) return false;
        if (!f.cur || f.idx >= f.cur->n()) {
            { const double t0 = mono_s(); f.cur = f.rd->next(); stats_.wait_s += mono_s() - t0; }
            f.idx = 0;
            if (!f.cur) {
                f.eof = true;
                if (f.rd->status() < 0 && !err_) { err_ = -1; errtxt_ = "error reading from input file"; }
                return false;
            }
            continue;
        }
        const Chunk &c = *f.cur; const size_t k = (size_t)f.idx;
        if (c.tid[k] < 0) { ++f.idx; continue; }                    // unplaced reads never reach the engines
        if (c.tid[k] >= cfg_.nref_limit) { f.eof = true; err_ = -3; errtxt_ = "a record names a reference sequence that is not in the first input's header"; return false; }
        const bool before = c.tid[k] < f.last_tid || (c.tid[k] == f.last_tid && c.pos[k] < f.last_pos);
        if (!(c.flag[k] & 4)) {
            if (before) { f.eof = true; err_ = -2; errtxt_ = "the input is not position sorted"; return false; }
            f.last_tid = c.tid[k]; f.last_pos = c.pos[k];
        } else if (before) { ++f.idx; continue; }                  // out-of-order unmapped-flagged record: filtered anyway
        return true;
    }
}

int ChunkPump::next_tid()
{
    int best = INT_MAX;
    for (auto &f : f_) {
        if (settle(f)) best = std::min(best, (int)f.cur->tid[(size_t)f.idx]);
        if (!f.carry.empty()) best = std::min(best, (int)f.carry.front().tid);
    }
    return best == INT_MAX ? -1 : best;
}

int64_t ChunkPump::next_pos(int tid)
{
    int64_t best = INT64_MAX;
    for (auto &f : f_) if (settle(f) && f.cur->tid[(size_t)f.idx] == tid) best = std::min(best, f.cur->pos[(size_t)f.idx]);
    return best;
}

bool ChunkPump::has_carry() const
{
    for (auto &f : f_) if (!f.carry.empty()) return true;
    return false;
}

int64_t ChunkPump::carry_next_covered(int64_t cursor) const
{
    int64_t best = INT64_MAX;
    for (auto &f : f_) for (auto &r : f.carry) if (span_end(r) > cursor) best = std::min(best, std::max(r.pos, cursor));
    return best;
}

int64_t ChunkPump::carry_max_end() const
{
    int64_t m = INT64_MIN;
    for (auto &f : f_) {
        for (auto &r : f.carry) m = std::max(m, span_end(r));
        for (auto &g : f.fresh) for (int64_t i = g.i0; i < g.i1; ++i) m = std::max(m, span_end(*g.c, i));
    }
    return m;
}

int64_t ChunkPump::fill_window(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> *staged_out)
{
    int64_t ce = ce_target;
    const int tpl = cfg_.tpl;
    auto take = [&](File &f) {
        // append the settled record to the window's slices
        if (!f.fresh.empty() && f.fresh.back().c == f.cur && f.fresh.back().i1 == f.idx) f.fresh.back().i1++;
        else f.fresh.push_back(Range{ f.cur, f.idx, f.idx + 1 });
        if (tpl) {
            Chunk &c = *f.cur; const size_t k = (size_t)f.idx;
            c.tpl_touch(tpl);
            c.t_id[k] = f.next_id;
            if (tpl == PumpConfig::TPL_DEPTH) {
                // bam2depth.c:598-623, in file order: a record that passes the read filters visits the name hash
                c.t_a[k] = 0;
                if (cfg_.depth_filter.passes(c.flag[k], c.mapq[k], c.l_qseq[k], c.cigar.data() + c.cig_off[k], c.cig_off[k + 1] - c.cig_off[k]))
                    c.t_a[k] = f.dclip.visit(c.names.data() + c.name_off[k], c.flag[k], c.tid[k], c.endpos((int64_t)k), c.mtid[k], c.mpos[k]);
            }
        }
        ++f.next_id;
        ++f.idx;
    };
    for (size_t fi = 0; fi < f_.size(); ++fi) {
        File &f = f_[fi];
        f.fresh.clear(); f.dropped.clear();
        f.first_fresh_id = f.next_id; f.n_fresh_paired = 0;
        int64_t count = 0;
        while (settle(f) && f.cur->tid[(size_t)f.idx] == tid && f.cur->pos[(size_t)f.idx] < ce) {
            const int64_t p = f.cur->pos[(size_t)f.idx];
            take(f);
            if (++count >= cfg_.max_reads && fi == 0 && p >= cb) {
                // cut the window after this start position (all reads sharing it stay together)
                while (settle(f) && f.cur->tid[(size_t)f.idx] == tid && f.cur->pos[(size_t)f.idx] == p) take(f);
                if (p + 1 > cb) ce = std::min(ce, p + 1);
                break;
            }
        }
    }
    if (tpl == PumpConfig::TPL_MPLP) {
        // Lookahead.  HTSlib hands out column c once a read starting beyond c was pushed, and a pair is resolved when its second mate is
        // pushed: the placeholders of a deletion / reference skip that straddles the mate's start show the RESOLVED quality of the next
        // query base from the column of the last push in front of the mate on (kernels_overlap.hip fix_y; dev_util.h placeholder_qual).
        // So a window also stages the reads that start at or after its end, up to the first one that reaches bam_plp_push -- the trigger
        // of the window's last columns -- as far as they can still be the mate of a staged read (start < largest staged end).  They add no
        // column here and stay carried.  Where the host does not decide who is pushed (pushed_on_device) the lookahead runs on to that end.
        Rec probe;
        for (auto &f : f_) {
            int64_t me = INT64_MIN;
            for (auto &r : f.carry) me = std::max(me, span_end(r));
            for (auto &g : f.fresh) for (int64_t i = g.i0; i < g.i1; ++i) me = std::max(me, span_end(*g.c, i));
            bool sure = false;
            if (!cfg_.pushed_on_device) for (auto &r : f.carry) if (r.pos >= ce && (!cfg_.pushed || cfg_.pushed(r))) { sure = true; break; }
            while (!sure && settle(f) && f.cur->tid[(size_t)f.idx] == tid && f.cur->pos[(size_t)f.idx] < me) {
                const size_t k = (size_t)f.idx;
                if (!cfg_.pushed_on_device) {
                    probe.tid = f.cur->tid[k]; probe.pos = f.cur->pos[k]; probe.flag = f.cur->flag[k]; probe.mapq = f.cur->mapq[k]; probe.rlen = f.cur->rlen[k];
                    sure = !cfg_.pushed || cfg_.pushed(probe);
                }
                take(f);
            }
        }
    }
    if (!staged_out) {
        // a window the lane only passes over: its reads visit the overlap hash at once (the host's own verdict on who is pushed)
        for (auto &f : f_) { f.n_carry_staged = f.carry.size(); if (tpl == PumpConfig::TPL_MPLP) pair_fresh(f, nullptr, 0); }
        return ce;
    }
    std::vector<StagedFile> &staged = *staged_out;
    const double t_stage0 = mono_s();
    std::vector<StagedFile::Slice> slices;
    staged.resize(f_.size());
    for (size_t fi = 0; fi < f_.size(); ++fi) {
        File &f = f_[fi];
        StagedFile &s = staged[fi];
        s.clear();
        XcolSpec xs; xs.n_tags = cfg_.xs_n_tags; xs.empty = cfg_.xs_empty;
        const XcolSpec *xp = xs.n_tags > 0 ? &xs : nullptr;
        for (auto &r : f.carry) s.add(r, cb, nullptr, xp);
        f.n_carry_staged = f.carry.size();
        slices.clear();
        for (auto &g : f.fresh) slices.push_back(StagedFile::Slice{ g.c.get(), g.i0, g.i1 });
        s.add_ranges(slices.data(), slices.size(), cb, xp, stage_threads_, stage_min_bytes_, &f.high_water, raw_mode_);
        s.finish();
        if (tpl == PumpConfig::TPL_DEPTH) {
            s.tpl = 1;
            s.clip.resize((size_t)s.n());
            size_t i = 0;
            for (auto &r : f.carry) s.clip[i++] = r.clip;
            for (auto &g : f.fresh) for (int64_t k = g.i0; k < g.i1; ++k) s.clip[i++] = g.c->t_a[(size_t)k];
        }
    }
    stats_.stage_s += mono_s() - t_stage0;
    return ce;
}

// ---- mpileup's overlap hash (host_names.h) ----
ChunkPump::Loc ChunkPump::locate(File &f, int64_t id)
{
    Loc l;
    if (id < 0) return l;
    if (id >= f.first_fresh_id) {
        int64_t o = id - f.first_fresh_id;                      // the window's new reads were taken one after the other
        for (auto &g : f.fresh) { if (o < g.i1 - g.i0) { l.c = g.c.get(); l.k = g.i0 + o; return l; } o -= g.i1 - g.i0; }
        return l;
    }
    auto it = std::lower_bound(f.carry.begin(), f.carry.end(), id, [](const Rec &r, int64_t v) { return r.id < v; });
    if (it != f.carry.end() && it->id == id) l.r = &*it;
    return l;
}

// the window's new reads that have not been there yet visit the overlap hash, in file order.  info == nullptr: PumpConfig::pushed decides
// who reaches bam_plp_push; otherwise the device's RI_* words of the staged reads do (bit 0 pushed, bit 1 in the pileup)
void ChunkPump::pair_fresh(File &f, const uint32_t *info, int64_t n_info)
{
    Rec probe;
    int64_t o = 0;
    for (auto &g : f.fresh)
        for (int64_t k = g.i0; k < g.i1; ++k, ++o) {
            if (o < f.n_fresh_paired) continue;
            Chunk &c = *g.c; const size_t q = (size_t)k;
            bool pushed, dropped = false;
            if (info) {
                const int64_t si = (int64_t)f.n_carry_staged + o;
                const uint32_t w = si < n_info ? info[si] : 0;
                pushed = (w & 1u) != 0;
                dropped = pushed && !(w & 2u) && c.rlen[q] > 0;
            } else {
                probe.tid = c.tid[q]; probe.pos = c.pos[q]; probe.flag = c.flag[q]; probe.mapq = c.mapq[q]; probe.rlen = c.rlen[q];
                pushed = cfg_.pushed ? cfg_.pushed(probe) : !(c.flag[q] & 4);
            }
            if (!pushed) continue;
            OverlapNames::Read r;
            r.flag = c.flag[q]; r.tid = c.tid[q]; r.pos = c.pos[q]; r.end = c.end(k);
            if (!(r.flag & 2u) && f.onames.plain_case(dropped, false)) { f.onames.push_plain(c.name_h[q], r.tid, r.pos, r.end); continue; }
            r.mtid = c.mtid[q]; r.l_qseq = c.l_qseq[q]; r.mpos = c.mpos[q]; r.isize = c.isize[q];
            if (f.onames.plain_case(dropped, OverlapNames::eligible(r.flag, r.tid, r.mtid, r.l_qseq, r.end, r.mpos, r.isize))) { f.onames.push_plain(c.name_h[q], r.tid, r.pos, r.end); continue; }
            r.h = c.name_h[q]; r.qname = c.names.data() + c.name_off[q]; r.l_qname = c.name_off[q + 1] - c.name_off[q] - 1; r.id = c.t_id[q];
            const int64_t holder = f.onames.push(r, dropped);
            if (holder < 0) continue;
            c.t_a[q] = holder;
            // the two stay staged together while either can touch a column
            Loc h = locate(f, holder);
            if (h.r) { c.t_b[q] = h.r->end(); h.r->mate_end = r.end; }
            else if (h.c) { c.t_b[q] = h.c->end(h.k); h.c->t_b[(size_t)h.k] = r.end; }
        }
    f.n_fresh_paired = o;
}

// mate[i] = staged index of the record whose entry staged read i found (-1: none, or that record is no longer staged -- it ended
// before this window and shares no column with read i)
void ChunkPump::fill_mates(File &f, int32_t *mate, int64_t n) const
{
    auto index_of = [&](int64_t id) -> int32_t {
        if (id < 0) return -1;
        if (id >= f.first_fresh_id) return (int32_t)((int64_t)f.n_carry_staged + (id - f.first_fresh_id));
        auto it = std::lower_bound(f.carry.begin(), f.carry.end(), id, [](const Rec &r, int64_t v) { return r.id < v; });
        return it != f.carry.end() && it->id == id ? (int32_t)(it - f.carry.begin()) : -1;
    };
    int64_t i = 0;
    for (auto &r : f.carry) { if (i >= n) return; mate[i++] = index_of(r.mate_id); }
    for (auto &g : f.fresh) for (int64_t k = g.i0; k < g.i1; ++k) { if (i >= n) return; mate[i++] = index_of(g.c->t_a[(size_t)k]); }
}

void ChunkPump::pair_staged(std::vector<StagedFile> &staged)
{
    if (cfg_.tpl != PumpConfig::TPL_MPLP) return;
    for (size_t fi = 0; fi < f_.size() && fi < staged.size(); ++fi) {
        File &f = f_[fi];
        pair_fresh(f, nullptr, 0);
        StagedFile &s = staged[fi];
        s.tpl = 2;
        s.mate.resize((size_t)s.n());
        fill_mates(f, s.mate.data(), s.n());
    }
}

void ChunkPump::pair_from_info(size_t fi, const uint32_t *info, int64_t n, int32_t *mate_out)
{
    for (int64_t i = 0; i < n; ++i) mate_out[i] = -1;
    if (cfg_.tpl != PumpConfig::TPL_MPLP || fi >= f_.size()) return;
    File &f = f_[fi];
    pair_fresh(f, info, n);
    fill_mates(f, mate_out, n);
}

bool ChunkPump::staged_has_span(size_t fi, size_t i) const
{
    if (fi >= f_.size()) return false;
    const File &f = f_[fi];
    if (i < f.n_carry_staged) return i < f.carry.size() && f.carry[i].rlen > 0;
    int64_t k = (int64_t)(i - f.n_carry_staged);
    for (auto &g : f.fresh) { if (k < g.i1 - g.i0) return g.c->rlen[(size_t)(g.i0 + k)] > 0; k -= g.i1 - g.i0; }
    return false;
}

int64_t ChunkPump::staged_max_span(size_t fi) const
{
    if (fi >= f_.size()) return 0;
    const File &f = f_[fi];
    int64_t m = 0;
    size_t i = 0;
    for (auto &r : f.carry) { if (i++ >= f.n_carry_staged) break; if ((int64_t)r.rlen > m) m = (int64_t)r.rlen; }
    for (auto &g : f.fresh) for (int64_t k = g.i0; k < g.i1; ++k) if ((int64_t)g.c->rlen[(size_t)k] > m) m = (int64_t)g.c->rlen[(size_t)k];
    return m;
}

void ChunkPump::drop(size_t fi, const std::vector<char> &dropped)
{
    File &f = f_[fi];
    f.dropped = dropped;             // consulted by retire() for the window's new reads
    std::deque<Rec> keep;
    size_t i = 0;
    for (auto &r : f.carry) { if (!(i < dropped.size() && dropped[i])) keep.push_back(std::move(r)); ++i; }
    // (n_carry_staged keeps the ORIGINAL count: `dropped` is indexed in staged order)
    f.carry.swap(keep);
}

void ChunkPump::retire(int64_t ce)
{
    // What stays staged for the next window: a record whose span reaches beyond the cut, and -- mpileup with overlap detection -- a record
    // whose partner in the overlap hash does (tweak_overlap_quality rewrites both, from both, so the two are staged together for as long as
    // either can touch a column).  Nothing else: which record found which is the lane's own state (host_names.h), not a replay.
    const bool partners = cfg_.tpl == PumpConfig::TPL_MPLP;
    for (auto &f : f_) {
        auto is_dropped = [&](size_t staged_index) { return staged_index < f.dropped.size() && f.dropped[staged_index]; };
        std::deque<Rec> keep;
        for (auto &r : f.carry) if (span_end(r) > ce || (partners && r.mate_end > ce)) keep.push_back(std::move(r));
        size_t si = f.n_carry_staged;
        for (auto &g : f.fresh)
            for (int64_t i = g.i0; i < g.i1; ++i, ++si) {
                if (is_dropped(si)) continue;
                if (span_end(*g.c, i) > ce || (partners && g.c->t_id.size() == g.c->pos.size() && g.c->t_b[(size_t)i] > ce)) { keep.emplace_back(); g.c->to_rec(i, keep.back()); }
            }
        f.carry.swap(keep);
        for (auto &r : f.carry) r.accepted = true;      // what stays was accepted by this window's -d replay
        f.fresh.clear(); f.dropped.clear(); f.n_carry_staged = 0;
        f.first_fresh_id = f.next_id; f.n_fresh_paired = 0;
    }
}

void ChunkPump::drop_tid_carry()
{
    for (auto &f : f_) { f.carry.clear(); f.fresh.clear(); f.dropped.clear(); f.n_carry_staged = 0; }
}

}  // namespace sta
