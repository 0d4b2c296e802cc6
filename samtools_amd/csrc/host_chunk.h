// host_chunk.h -- the drivers' fast input lane: records are decoded on several threads straight into structure-of-arrays
// chunks whose pools have the staging layout, so that a window is assembled by concatenating chunk slices (bulk copies +
// offset rebasing) instead of one decoded record at a time.  Same role as host_pump.h (the reference's "pull reads while they
// can still touch the column" loop, bam_plcmd.c:607 / bam2depth.c:578-663), same window semantics -- host_scan.cpp checks that
// both lanes stage byte-identical windows -- but per-record work is left only for the few reads that straddle a window end.
// Used when no per-record host formatting is asked for (no RNEXT column, no --output-mods, no -G read-group list); wanted aux
// tags travel with the chunks as text.
#pragma once
#include "host_io.h"
#include "host_pump.h"
#include "host_stage.h"
#include "host_bgzf.h"
#include "host_gpu_inflate.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace sta {

// consecutive records of one file in file order; pools laid out like StagedFile's (8-padded bases, NUL-terminated names)
struct Chunk {
    std::vector<int32_t> tid, l_qseq, mtid, rlen;
    std::vector<int64_t> pos, mpos, isize;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq, aux;               // aux: STA_AUX_HAS_BQ / STA_AUX_HAS_ZQ
    std::vector<uint32_t> cig_off, base_off8, name_off;      // n + 1 entries each
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq, qual, bq;           // bq maintained only once some record carries BQ:Z
    bool has_bq_pool = false;
    std::vector<char> names;
    std::vector<uint64_t> name_h;                 // qname_hash64 of every name (host_names.h): hashed here, on the decode threads
    // written by the window producer when it takes a record (PumpConfig::tpl, host_names.h): the lane's running number of the record; depth
    // -s: its clip column / mpileup: the record whose overlap-hash entry it found (-1); mpileup: its partner's end (INT64_MIN: no partner)
    std::vector<int64_t> t_id, t_a, t_b; int t_mode = 0;      // t_mode: PumpConfig::TPL_*
    void tpl_touch(int mode) { if (t_id.size() != pos.size()) { t_mode = mode; t_id.assign(pos.size(), -1); t_a.assign(pos.size(), mode == 1 ? 0 : -1); t_b.assign(pos.size(), INT64_MIN); } }
    // wanted aux tags (AlnReader::set_wanted_tags) as text, n_tags entries per record: the staging layer turns them into text
    // columns (--output-extra tags of mpileup; MD:Z for the consensus path)
    int n_tags = 0;
    std::vector<uint32_t> tag_off; std::vector<char> tag_text, tag_has;
    // BAM input with device staging on (host_stage.h add_ranges): the group's inflated bytes as they came off the stream, page-locked,
    // and for every record kept here the offset of its refID field in them.  The window producer then uploads these bytes instead of
    // copying the pools above; raw_ok = false when a record's CIGAR came out of a CG tag (its own CIGAR field is a placeholder).
    std::shared_ptr<pvector<uint8_t>> raw;
    std::vector<uint32_t> rec_off;
    bool raw_ok = true;
    int64_t n() const { return (int64_t)pos.size(); }
    int64_t end(int64_t i) const { return pos[(size_t)i] + rlen[(size_t)i]; }
    int64_t endpos(int64_t i) const { int64_t l = (flag[(size_t)i] & 4) ? 0 : rlen[(size_t)i]; return pos[(size_t)i] + (l > 0 ? l : 1); }
    void append(const Rec &r);
    void close();                                 // final offset entries
    void reset();                                 // empty again, capacities kept (chunks go round: ChunkReader::new_chunk)
    void to_rec(int64_t i, Rec &r) const;         // materialise one record (reads that stay carried across windows)
};

// decodes one input on `threads` parser threads; chunks come out in file order
class ChunkReader {
public:
    // gpu_device >= 0: BAM files on disk are inflated by that device's decoder (host_gpu_inflate.h) when it is usable
    ChunkReader(AlnReader *rd, int threads, bool keep_raw = false, int gpu_device = -1);
    ~ChunkReader();
    // next chunk in file order; nullptr at end of data or after an error (status() tells which)
    std::shared_ptr<Chunk> next();
    int status() const { return status_; }        // 0 = fine / clean end, <0 = decode error
private:
    AlnReader *rd_;
    std::vector<std::thread> th_;
    std::mutex io_m_, out_m_;
    std::condition_variable cv_out_, cv_room_;
    std::map<uint64_t, std::shared_ptr<Chunk>> done_;
    uint64_t next_in_ = 0, next_out_ = 0;
    std::atomic<bool> io_end_{ false }; std::atomic<int> io_status_{ 0 };
    bool stop_ = false;
    int status_ = 0;
    uint64_t bad_seq_ = UINT64_MAX;               // first group that failed to parse
    size_t max_ahead_;
    bool keep_raw_ = false;
    // page-locked group buffers go round: a chunk hands its buffer back when the last window that uploads from it is done
    // (the pool is shared with the buffers' deleters: staged windows may still hold buffers when the reader has gone)
    struct RawPool { std::mutex m; std::vector<std::unique_ptr<pvector<uint8_t>>> free; };
    std::shared_ptr<RawPool> pool_ = std::make_shared<RawPool>();
    std::shared_ptr<pvector<uint8_t>> get_buf();
    // chunks go round as well: a fresh Chunk per group meant ~20 growing vectors each, i.e. allocator calls, page faults and unmaps by
    // the hundred thousand per second of input -- which do not scale beyond a few threads (address-space lock, TLB shootdowns)
    struct ChunkPool { std::mutex m; std::vector<std::unique_ptr<Chunk>> free; };
    std::shared_ptr<ChunkPool> cpool_ = std::make_shared<ChunkPool>();
    std::shared_ptr<Chunk> new_chunk();
    void work();
    // BAM files on disk: the file is mapped, every parser thread cuts the next group of BGZF blocks (under io_m_, headers only), inflates
    // them itself straight into the group's buffer and parses it -- no inflated byte passes through a shared stream.  Groups are whole
    // BGZF blocks, so a record may straddle two of them: a group hands the bytes of its unfinished last record (`carry`) to the next
    // one, which puts them in front of its own data (the buffers keep HEAD bytes free for that).  That hand-over is the only serial
    // step: a walk over the group's block_size fields.
    struct Link { std::vector<uint8_t> carry; uint64_t skip = 0; bool bad = false; };      // skip: inflated bytes in front of the first record (the header)
    std::unique_ptr<BgzfMap> map_;
    uint64_t cut_off_ = 0;                         // compressed offset of the next block to cut (guarded by io_m_)
    std::map<uint64_t, Link> links_;               // what group seq receives from group seq - 1 (guarded by out_m_)
    std::condition_variable cv_link_;
    void publish_link(uint64_t seq, Link &&l);
    struct MGroup {                                // a group of whole BGZF blocks on its way to a chunk
        uint64_t seq = 0; size_t total = 0;
        std::vector<BgzfMap::Block> blocks;
        std::shared_ptr<pvector<uint8_t>> keep;    // its buffer: GROUP_HEAD free bytes, then the inflated blocks
        bool bad = false, verify = false;          // verify: the bytes came back from the device, their CRC-32s are still to be checked
        bool inflate_here = false;                 // the parser inflates the blocks itself (the device decoder was not up yet)
    };
    bool cut_group(MGroup &g, int *end_status);
    void finish_stream(int st);
    void process_group(MGroup &g, pvector<uint8_t> &raw, Rec &r, std::string &scratch);
    void work_mapped();
    // with the device's decoder: a feeder thread and a queue of inflated groups in front of the parsers
    std::unique_ptr<GpuInflater> gpu_;
    int gpu_device_ = -1;
    std::atomic<int> gpu_state_{ 0 };              // 0 = coming up (its own thread: the HIP runtime takes a quarter of a second), 1 = ready, -1 = none
    std::thread gpu_init_;
    std::deque<MGroup> ready_q_; std::condition_variable cv_ready_; bool feed_end_ = false;     // (guarded by out_m_)
    void work_gpu_feeder();
    void work_gpu_parse();
};

// window source over chunked readers
class ChunkPump : public WindowSource {
public:
    ChunkPump(std::vector<std::unique_ptr<AlnReader>> &readers, const PumpConfig &cfg, int threads);
    int next_tid() override;
    int64_t next_pos(int tid) override;
    bool has_carry() const override;
    int64_t carry_next_covered(int64_t cursor) const override;
    int64_t carry_max_end() const override;
    int64_t fill_staged(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> &staged) override { return fill_window(tid, cb, ce_target, &staged); }
    int64_t fill_unstaged(int tid, int64_t cb, int64_t ce_target) override { return fill_window(tid, cb, ce_target, nullptr); }
    bool staged_has_span(size_t f, size_t i) const override;
    int64_t staged_max_span(size_t f) const override;
    void drop(size_t f, const std::vector<char> &dropped) override;
    void pair_staged(std::vector<StagedFile> &staged) override;
    void pair_from_info(size_t f, const uint32_t *info, int64_t n, int32_t *mate_out) override;
    void retire(int64_t ce) override;
    void drop_tid_carry() override;
    int error() const override { return err_; }
    const char *error_text() const override { return errtxt_.c_str(); }
    // where the producer's time goes (seconds): waiting for the decode threads / assembling the staging arrays
    struct Stats { double wait_s = 0, stage_s = 0; };
    const Stats &stats() const { return stats_; }
    void producer_split(double *decode_wait, double *stage_copy) const override { *decode_wait = stats_.wait_s; *stage_copy = stats_.stage_s; }
private:
    Stats stats_;
    int stage_threads_ = 1; size_t stage_min_bytes_ = (size_t)4 << 20;
    int raw_mode_ = 0;       // 0: pools copied on the host; 1: pools of a window's new reads built on the device out of the raw records; 2: both, compared (tests)
    struct Range { std::shared_ptr<Chunk> c; int64_t i0, i1; };
    struct File {
        std::unique_ptr<ChunkReader> rd;
        std::shared_ptr<Chunk> cur; int64_t idx = 0;      // next unread record
        bool eof = false;
        int64_t last_pos = -1; int last_tid = -1;
        std::deque<Rec> carry;                            // reads carried from earlier windows (materialised)
        std::vector<Range> fresh;                         // the current window's new reads, as chunk slices
        size_t n_carry_staged = 0;                        // carried reads at the front of the staged order
        std::vector<char> dropped;                        // per staged read: removed by the -d cap in this window
        StagedFile::PoolSizes high_water;                 // largest staging pools a window of this input needed so far
        // template state in file order (host_names.h)
        int64_t next_id = 0, first_fresh_id = 0;          // running number of the next record taken / of the window's first new read
        int64_t n_fresh_paired = 0;                       // new reads of the window that have visited the overlap hash
        DepthMateClip dclip; OverlapNames onames;
    };
    struct Loc { Rec *r = nullptr; Chunk *c = nullptr; int64_t k = 0; };
    Loc locate(File &f, int64_t id);                      // a staged record by its running number (carried or new), or nothing
    void pair_fresh(File &f, const uint32_t *info, int64_t n_info);
    void fill_mates(File &f, int32_t *mate, int64_t n) const;
    PumpConfig cfg_;
    std::vector<File> f_;
    int err_ = 0; std::string errtxt_;
    int64_t fill_window(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> *staged);     // staged == nullptr: bookkeeping only
    bool settle(File &f);                                 // positions f.cur/f.idx on the next usable record; false at end
    int64_t span_end(const Chunk &c, int64_t i) const { return cfg_.use_endpos ? c.endpos(i) : c.end(i); }
    int64_t span_end(const Rec &r) const { return cfg_.use_endpos ? r.endpos() : r.end(); }
};

}  // namespace sta
