// host_pump.h -- streams position-sorted records of N input files into column windows.
// Host-side counterpart of the reference's "pull reads while they can still touch the current
// column" loop (bam_mplp64_auto -> mplp_func, bam_plcmd.c:607; fastdepth_core merge,
// bam2depth.c:578-663), restated for window-at-a-time processing on the device: a window
// [cb, ce) of one contig receives every read whose span can touch it, i.e. the reads carried over
// from the previous window plus all unread records that start before ce.
#pragma once
#include "host_io.h"
#include "host_stage.h"
#include "host_names.h"
#include <deque>
#include <functional>
#include <set>

namespace sta {

struct PumpConfig {
    int64_t window_cols = 1 << 20;     // target columns per window
    int64_t max_reads = 4 << 20;       // soft cap of new reads per file per window
    bool use_endpos = false;           // carry criterion: bam_endpos (depth) instead of pos + rlen (mpileup)
    // Template state (host_names.h): the reference's name hashes are sequential over the whole file, so the lane runs them itself while it
    // takes records off the inputs and stages what every record found.
    //   TPL_DEPTH  depth -s (bam2depth.c:598-623): every record that passes `depth_filter` visits the per-file name -> end hash; the
    //              staged arrays carry its clip column (sta_reads.olap_clip).
    //   TPL_MPLP   mpileup's overlap hash: every record that reaches bam_plp_push -- `pushed` decides, EXACTLY as
    //              k_prep_reads does on the device -- visits the per-file hash; the staged arrays carry the staged index of the record
    //              whose entry it found (sta_reads.olap_mate), and the two partners stay staged while either can touch a column.
    //              A window the lane only passes over (fill_unstaged) is paired at once; a staged window is paired when the driver
    //              calls pair_staged() -- or, where the device decides who is pushed (-C) or turned away (a -d cap that triggers),
    //              pair_from_info() with the window's RI_* words (driver_mpileup.cpp: the engine's resolver callback).
    enum { TPL_NONE = 0, TPL_DEPTH = 1, TPL_MPLP = 2 };
    int tpl = TPL_NONE;
    DepthReadFilter depth_filter;
    std::function<bool(const Rec &)> pushed;          // TPL_MPLP: tid, pos, rlen, flag, mapq (and rg in the record lane) are filled in
    bool pushed_on_device = false;                    // TPL_MPLP: every staged window goes through pair_from_info() (-C): `pushed` only serves windows passed over
    // staging options of fill_staged(): -G read groups to mark STA_AUX_SKIP, --output-extra columns formatted on the host
    const std::set<std::string> *rg_excl = nullptr;
    bool xs_rnext = false; int xs_n_tags = 0; char xs_empty = '*'; bool xs_mods = false;
    // contigs of the header the driver prints from (the first input's): a record of any input naming a later one is an error
    // instead of an out-of-range name / length lookup
    int nref_limit = INT32_MAX;
    // chunk lane, BAM input: hand the engine the raw alignment records of a window's new reads and let it cut their CIGAR / bases /
    // qualities / names out on the device (host_stage.h add_ranges raw_mode, kernels_stage.hip) instead of copying them into the
    // staging pools here.  Set by drivers whose windows go to an engine through sta_stage_window(); STA_STAGE_DEVICE=0 turns it off.
    bool device_pools = false;
    // chunk lane, BAM files on disk: the device whose decoder inflates the BGZF blocks (kernels_inflate.hip through host_gpu_inflate.h),
    // -1 = the reader's own threads do.  Set by the drivers next to device_pools; used only with STA_GPU_INFLATE=1 (see host_gpu_inflate.cpp).
    int inflate_device = -1;
};

// What the window loops of the drivers need from an input lane (Pump below: one decoded record at a time; ChunkPump in
// host_chunk.h: chunk slices).  A window [cb, ce) of one contig receives every read whose span can touch it.
class WindowSource {
public:
    virtual ~WindowSource() {}
    virtual int next_tid() = 0;                            // smallest tid with unread or carried reads; -1 at the end
    virtual int64_t next_pos(int tid) = 0;                 // first unread position on tid (INT64_MAX if none)
    virtual bool has_carry() const = 0;
    virtual int64_t carry_next_covered(int64_t cursor) const = 0;
    virtual int64_t carry_max_end() const = 0;
    // consume the records of `tid` starting before ce_target (the read cap may cut the window short) and stage them,
    // carried reads first, relative to cb; returns the actual window end
    virtual int64_t fill_staged(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> &staged) = 0;
    // the same window bookkeeping without building the staging arrays: lets a driver pass over columns it does not own
    // (a rank of a sharded run walking up to its block) with exactly the carry state the unsharded run would have there
    virtual int64_t fill_unstaged(int tid, int64_t cb, int64_t ce_target) = 0;
    // consume everything of `tid` before `pos` in steps of `step` columns (fill_unstaged + retire); returns pos
    int64_t skip_to(int tid, int64_t from, int64_t pos, int64_t step)
    {
        int64_t cur = from;
        while (cur < pos && !error()) {
            if (next_pos(tid) == INT64_MAX && !has_carry()) break;
            int64_t ce = fill_unstaged(tid, cur, pos - cur > step ? cur + step : pos);
            retire(ce);
            cur = ce > cur ? ce : cur + 1;
        }
        return pos;
    }
    virtual bool staged_has_span(size_t f, size_t i) const = 0;      // reference span > 0 of the i-th staged read of file f
    // longest reference span among the staged reads of file f (the staging arrays need not hold the CIGARs: with device-side pools
    // of BAM records they do not -- host_stage.h raw_mode)
    virtual int64_t staged_max_span(size_t f) const = 0;
    virtual void drop(size_t f, const std::vector<char> &dropped) = 0;
    // TPL_MPLP, after fill_staged(): the window's new reads visit the overlap hash (PumpConfig::pushed decides who reaches it) and
    // staged[f].mate is filled in
    virtual void pair_staged(std::vector<StagedFile> &staged) = 0;
    // ... the same with the device's verdict on every staged read of file f (info[i]: bit 0 = reached bam_plp_push, bit 1 = in the
    // pileup; a pushed read with a reference span that is not in the pileup was turned away by the -d cap): mate_out[i] = staged index
    // of the record whose entry read i found, or -1
    virtual void pair_from_info(size_t f, const uint32_t *info, int64_t n, int32_t *mate_out) = 0;
    virtual void retire(int64_t ce) = 0;
    virtual void drop_tid_carry() = 0;
    virtual int error() const = 0;
    virtual const char *error_text() const = 0;
    // seconds the producer spent waiting for decoded input / copying into the staging arrays (0, 0 when a lane does not keep them)
    virtual void producer_split(double *decode_wait, double *stage_copy) const { *decode_wait = 0; *stage_copy = 0; }
};

class Pump : public WindowSource {
public:
    Pump(std::vector<std::unique_ptr<AlnReader>> &readers, const PumpConfig &cfg);
    // smallest tid that still has unread records (or carried reads); -1 when everything is consumed
    int next_tid() override;
    // position of the first unread record on `tid` over all files (INT64_MAX if none)
    int64_t next_pos(int tid) override;
    bool has_carry() const override;
    // first column >= cursor that a carried read can touch (INT64_MAX if none)
    int64_t carry_next_covered(int64_t cursor) const override;
    int64_t carry_max_end() const override;
    // Consume records of `tid` starting before `ce_target` (possibly fewer: the read cap may cut the
    // window short) and return the actual window end.  reads[f] = carried + new records of file f.
    int64_t fill(int tid, int64_t cb, int64_t ce_target, std::vector<std::vector<const Rec *>> &reads);
    // After the window [cb, ce) was processed: keep only reads that extend beyond ce.
    int64_t fill_staged(int tid, int64_t cb, int64_t ce_target, std::vector<StagedFile> &staged) override;
    int64_t fill_unstaged(int tid, int64_t cb, int64_t ce_target) override;
    bool staged_has_span(size_t f, size_t i) const override { return f < last_.size() && i < last_[f].size() && last_[f][i]->rlen > 0; }
    int64_t staged_max_span(size_t f) const override { int64_t m = 0; if (f < last_.size()) for (const Rec *r : last_[f]) if ((int64_t)r->rlen > m) m = (int64_t)r->rlen; return m; }
    void pair_staged(std::vector<StagedFile> &staged) override;
    void pair_from_info(size_t f, const uint32_t *info, int64_t n, int32_t *mate_out) override;
    void retire(int64_t ce) override;
    // before retire(): reads of file f (indexed as fill() returned them) that the -d cap dropped in this window leave the
    // iterator for good, exactly as bam_plp_push never stored them
    void drop(size_t f, const std::vector<char> &dropped) override;
    void drop_tid_carry() override;    // forget carried reads when leaving a contig
    int error() const override { return err_; }   // <0 after a decode error or unsorted input
    const char *error_text() const override { return errtxt_.c_str(); }
private:
    std::vector<std::unique_ptr<AlnReader>> &rd_;
    PumpConfig cfg_;
    std::vector<Rec> pend_; std::vector<char> has_pend_, eof_;
    std::vector<std::deque<Rec>> carry_;
    std::vector<int64_t> last_pos_; std::vector<int> last_tid_;
    int err_ = 0; std::string errtxt_;
    std::vector<std::vector<const Rec *>> last_;      // reads of the window fill_staged() staged last
    void advance(size_t f);
    // template state in file order (host_names.h): running record numbers, the carried reads at the front of the staged order, the new
    // reads that have visited the overlap hash, the two hashes
    std::vector<int64_t> next_id_; std::vector<size_t> n_carry_staged_; std::vector<int64_t> n_fresh_paired_;
    std::vector<DepthMateClip> dclip_; std::vector<OverlapNames> onames_;
    void take(size_t f);
    void pair_fresh(size_t f, const uint32_t *info, int64_t n_info);
    void fill_mates(size_t f, int32_t *mate, int64_t n) const;
    int64_t span_end(const Rec &r) const { return cfg_.use_endpos ? r.endpos() : r.end(); }
};

}  // namespace sta
