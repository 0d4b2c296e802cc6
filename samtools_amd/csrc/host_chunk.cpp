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
    auto take = [](File &f) {
        // append the settled record to the window's slices
        if (!f.fresh.empty() && f.fresh.back().c == f.cur && f.fresh.back().i1 == f.idx) f.fresh.back().i1++;
        else f.fresh.push_back(Range{ f.cur, f.idx, f.idx + 1 });
        ++f.idx;
    };
    for (size_t fi = 0; fi < f_.size(); ++fi) {
        File &f = f_[fi];
        f.fresh.clear(); f.dropped.clear();
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
    if (cfg_.surely_pushed) {
        Rec probe;
        for (auto &f : f_) {
            int64_t me = INT64_MIN;
            for (auto &r : f.carry) me = std::max(me, span_end(r));
            for (auto &g : f.fresh) for (int64_t i = g.i0; i < g.i1; ++i) me = std::max(me, span_end(*g.c, i));
            bool sure = false;
            for (auto &r : f.carry) if (r.pos >= ce && cfg_.surely_pushed(r)) { sure = true; break; }
            while (!sure && settle(f) && f.cur->tid[(size_t)f.idx] == tid && f.cur->pos[(size_t)f.idx] < me) {
                const size_t k = (size_t)f.idx;
                probe.tid = f.cur->tid[k]; probe.pos = f.cur->pos[k]; probe.flag = f.cur->flag[k]; probe.mapq = f.cur->mapq[k];
                sure = cfg_.surely_pushed(probe);
                take(f);
            }
        }
    }
    if (!staged_out) { for (auto &f : f_) f.n_carry_staged = f.carry.size(); return ce; }
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
    }
    stats_.stage_s += mono_s() - t_stage0;
    return ce;
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
    struct Stay { int64_t pos; const char *qname; };
    const bool proper_only = cfg_.mates_proper_only;
    auto paired_ok = [proper_only](unsigned flag) { return (flag & 1) && ((flag & 2) || !proper_only) && !(flag & 8); };
    for (auto &f : f_) {
        auto is_dropped = [&](size_t staged_index) { return staged_index < f.dropped.size() && f.dropped[staged_index]; };
        // pass 1: who stays because its span reaches beyond ce (position sorted: carried reads first, then the new ones)
        std::vector<Stay> stay;
        if (cfg_.keep_mates) {
            for (auto &r : f.carry) if (span_end(r) > ce && paired_ok(r.flag)) stay.push_back(Stay{ r.pos, r.qname.c_str() });
            size_t si = f.n_carry_staged;
            for (auto &g : f.fresh)
                for (int64_t i = g.i0; i < g.i1; ++i, ++si)
                    if (!is_dropped(si) && span_end(*g.c, i) > ce && paired_ok(g.c->flag[(size_t)i]))
                        stay.push_back(Stay{ g.c->pos[(size_t)i], g.c->names.data() + g.c->name_off[(size_t)i] });
        }
        auto mate_stays = [&](unsigned flag, int32_t tid, int32_t mtid, int64_t mpos, const char *qname) {
            if (stay.empty() || !paired_ok(flag) || mtid != tid) return false;
            auto lo = std::lower_bound(stay.begin(), stay.end(), mpos, [](const Stay &s, int64_t p) { return s.pos < p; });
            for (; lo != stay.end() && lo->pos == mpos; ++lo) if (lo->qname != qname && !strcmp(lo->qname, qname)) return true;
            return false;
        };
        // A record that stays ONLY for its mate's sake holds nothing any more once a read beyond its end was pushed before that mate
        // (bam_plp_next frees it and overlap_remove takes the entry of its name along): see Pump::retire
        // ("that mate" = the NEXT pushed record of its template, which need not be the one at its mate position: see Pump::retire.  `self`
        // names the record asked about: a carried Rec, or (chunk, index) of a new one.)
        auto freed_before_mate = [&](int64_t end, int64_t mpos, const char *qname, const void *self, int64_t self_i) {
            if (!cfg_.surely_pushed) return false;
            bool behind = false;
            for (auto &q : f.carry) {
                if ((const void *)&q == self) { behind = true; continue; }
                if (!behind) continue;
                if (q.pos > mpos) return false;       // (a record AT the mate position in front of the mate in the file frees it too)
                if (!cfg_.surely_pushed(q)) continue;
                if (!strcmp(q.qname.c_str(), qname)) return false;
                if (q.pos > end) return true;
            }
            Rec probe;
            size_t sj = f.n_carry_staged;
            for (auto &g : f.fresh)
                for (int64_t i = g.i0; i < g.i1; ++i, ++sj) {
                    const size_t k = (size_t)i;
                    if ((const void *)g.c.get() == self && i == self_i) { behind = true; continue; }
                    if (!behind) continue;
                    if (g.c->pos[k] > mpos) return false;
                    if (is_dropped(sj)) continue;
                    probe.tid = g.c->tid[k]; probe.pos = g.c->pos[k]; probe.flag = g.c->flag[k]; probe.mapq = g.c->mapq[k];
                    if (!cfg_.surely_pushed(probe)) continue;
                    if (!strcmp(g.c->names.data() + g.c->name_off[k], qname)) return false;
                    if (g.c->pos[k] > end) return true;
                }
            return false;
        };
        // pass 2: decide for the carried reads before anything moves (`stay` points into them), then keep / materialise
        std::vector<char> keepc(f.carry.size(), 0);
        // (where the host cannot tell who is pushed -- -l, -G, -C, --min-read-len -- the record stays together with every record that
        // starts between its end and its mate: the replay sees from their RI_PUSHED whether one of them freed it)
        struct Ctx { int64_t pos, end, mpos; std::string qname; };
        std::vector<Ctx> ctx;
        auto for_mate_only = [&](int64_t pos, int64_t end, unsigned flag, int32_t tid, int32_t mtid, int64_t mpos, const char *qname, const void *self, int64_t self_i) {
            if (!mate_stays(flag, tid, mtid, mpos, qname) || freed_before_mate(end, mpos, qname, self, self_i)) return false;
            ctx.push_back(Ctx{ pos, end, mpos, qname });
            return true;
        };
        // (by position, or as another record of the template: one that starts inside the kept record's span is no context record by position)
        auto in_ctx = [&](int64_t pos, const char *qname) {
            for (auto &iv : ctx) if (pos <= iv.mpos && (pos > iv.end || (pos >= iv.pos && !strcmp(iv.qname.c_str(), qname)))) return true;
            return false;
        };
        // (1) A record whose span ends at the cut is still in the reference's buffer while no pushed read has started beyond its end
        // (Pump::retire): max_start = the last pushed start in front of the cut (the lists are position sorted: looked for from the back).
        // Only while the contig has reads to come: at its end the reference flushes its buffer, and a record kept here for ever would keep
        // the window loop going for ever.
        int64_t max_start = INT64_MIN;
        int cur_tid = -1;
        if (!f.carry.empty()) cur_tid = f.carry.front().tid;
        else for (auto &g : f.fresh) if (g.i1 > g.i0) { cur_tid = g.c->tid[(size_t)g.i0]; break; }
        if (cfg_.keep_mates && cur_tid >= 0 && next_pos(cur_tid) != INT64_MAX) {
            Rec probe;
            bool found = false;
            size_t sj = f.n_carry_staged;
            for (auto &g : f.fresh) sj += (size_t)(g.i1 - g.i0);
            for (auto g = f.fresh.rbegin(); g != f.fresh.rend() && !found; ++g)
                for (int64_t i = g->i1 - 1; i >= g->i0; --i) {
                    --sj;
                    const size_t k = (size_t)i;
                    if (is_dropped(sj) || g->c->pos[k] >= ce) continue;
                    if (cfg_.surely_pushed && !cfg_.pushed_unknown) { probe.tid = g->c->tid[k]; probe.pos = g->c->pos[k]; probe.flag = g->c->flag[k]; probe.mapq = g->c->mapq[k]; if (!cfg_.surely_pushed(probe)) continue; }
                    max_start = g->c->pos[k]; found = true; break;
                }
            if (!found)
                for (auto r = f.carry.rbegin(); r != f.carry.rend(); ++r)
                    if (r->pos < ce && (!cfg_.surely_pushed || cfg_.pushed_unknown || cfg_.surely_pushed(*r))) { max_start = r->pos; break; }
        }
        const bool alive_rule = cfg_.keep_mates && max_start != INT64_MIN;
        std::vector<const char *> multi;          // names of secondary / supplementary records: templates with more than two records
        { size_t i = 0; for (auto &r : f.carry) {
            const int64_t e = span_end(r);
            if (cfg_.keep_mates && (r.flag & 0x900)) multi.push_back(r.qname.c_str());
            keepc[i++] = e > ce || (alive_rule && e >= max_start) || for_mate_only(r.pos, e, r.flag, r.tid, r.mtid, r.mpos, r.qname.c_str(), &r, 0); } }
        // the window's new reads, in staged order (dropped ones never kept)
        size_t n_fresh = 0;
        for (auto &g : f.fresh) n_fresh += (size_t)(g.i1 - g.i0);
        std::vector<char> keepf(n_fresh, 0);
        auto each_fresh = [&](auto &&fn) {          // fn(chunk, index in chunk, index in keepf); dropped records skipped
            size_t si = f.n_carry_staged, j = 0;
            for (auto &g : f.fresh)
                for (int64_t i = g.i0; i < g.i1; ++i, ++si, ++j) if (!is_dropped(si)) fn(*g.c, i, j);
        };
        auto name_of = [](const Chunk &c, int64_t i) { return c.names.data() + c.name_off[(size_t)i]; };
        each_fresh([&](const Chunk &c, int64_t i, size_t j) {
            const size_t k = (size_t)i;
            const int64_t e = span_end(c, i);
            if (cfg_.keep_mates && (c.flag[k] & 0x900)) multi.push_back(name_of(c, i));
            keepf[j] = e > ce || (alive_rule && e >= max_start) || for_mate_only(c.pos[k], e, c.flag[k], c.tid[k], c.mtid[k], c.mpos[k], name_of(c, i), &c, i);
        });
        if (cfg_.keep_mates) {
            // (2) the records of a template with more than two records all stay while one of them does: see Pump::retire
            if (!multi.empty()) {
                std::vector<Ctx> tpl;                  // (first position, -, last position, name) of such a template with a record that stays
                auto note = [&](const char *qn) {
                    bool is_multi = false;
                    for (const char *m : multi) if (!strcmp(m, qn)) { is_multi = true; break; }
                    if (!is_multi) return;
                    for (auto &t : tpl) if (!strcmp(t.qname.c_str(), qn)) return;
                    tpl.push_back(Ctx{ INT64_MAX, 0, INT64_MIN, qn });
                };
                { size_t i = 0; for (auto &r : f.carry) { if (keepc[i]) note(r.qname.c_str()); ++i; } }
                each_fresh([&](const Chunk &c, int64_t i, size_t j) { if (keepf[j]) note(name_of(c, i)); });
                if (!tpl.empty()) {
                    auto span = [&](const char *qn, int64_t pos) { for (auto &t : tpl) if (!strcmp(t.qname.c_str(), qn)) { t.pos = std::min(t.pos, pos); t.mpos = std::max(t.mpos, pos); } };
                    for (auto &r : f.carry) span(r.qname.c_str(), r.pos);
                    each_fresh([&](const Chunk &c, int64_t i, size_t) { span(name_of(c, i), c.pos[(size_t)i]); });
                    auto of_tpl = [&](const char *qn, int64_t pos, int64_t end) {
                        for (auto &t : tpl) if (!strcmp(t.qname.c_str(), qn)) { ctx.push_back(Ctx{ pos, end, t.mpos, qn }); return true; }
                        return false;
                    };
                    { size_t i = 0; for (auto &r : f.carry) { if (!keepc[i] && of_tpl(r.qname.c_str(), r.pos, span_end(r))) keepc[i] = 1; ++i; } }
                    each_fresh([&](const Chunk &c, int64_t i, size_t j) { if (!keepf[j] && of_tpl(name_of(c, i), c.pos[(size_t)i], span_end(c, i))) keepf[j] = 1; });
                }
            }
        }
        if (!ctx.empty()) {          // the context records
            { size_t i = 0; for (auto &r : f.carry) { if (!keepc[i] && in_ctx(r.pos, r.qname.c_str())) keepc[i] = 1; ++i; } }
            each_fresh([&](const Chunk &c, int64_t i, size_t j) { if (!keepf[j] && in_ctx(c.pos[(size_t)i], name_of(c, i))) keepf[j] = 1; });
        }
        std::vector<std::pair<const Chunk *, int64_t>> fresh_keep;       // in position order (carried first, then the window's new reads)
        each_fresh([&](const Chunk &c, int64_t i, size_t j) { if (keepf[j]) fresh_keep.emplace_back(&c, i); });
        std::deque<Rec> keep;
        { size_t i = 0; for (auto &r : f.carry) { if (keepc[i++]) keep.push_back(std::move(r)); } }
        for (auto &pr : fresh_keep) { keep.emplace_back(); pr.first->to_rec(pr.second, keep.back()); }
        f.carry.swap(keep);
        for (auto &r : f.carry) r.accepted = true;      // what stays was accepted by this window's -d replay
        f.fresh.clear(); f.dropped.clear(); f.n_carry_staged = 0;
    }
}

void ChunkPump::drop_tid_carry()
{
    for (auto &f : f_) { f.carry.clear(); f.fresh.clear(); f.dropped.clear(); f.n_carry_staged = 0; }
}

}  // namespace sta
