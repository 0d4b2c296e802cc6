// driver_mpileup.cpp -- `samtools mpileup` command driver on top of the device engine.
//
// Mirrors bam_mpileup()/mpileup() of the reference (bam_plcmd.c:1075-1272, :470-934): same options,
// same stdout text, same mandatory stderr line and exit status.  What differs is the execution
// model: instead of asking the iterator for one column at a time, the driver cuts each contig into
// windows, stages every read that can touch a window (host_pump + host_stage), and the engine
// produces the window's text on the GPU (include/samtools_amd.h).  The -a/-aa bookkeeping of
// mpileup() (:610-660, :880-910) is restated at window granularity below.
#include "host_io.h"
#include "host_stage.h"
#include "host_pump.h"
#include "host_chunk.h"
#include "host_bgzf.h"
#include "driver_pipeline.h"
#include "driver_shard.h"
#include "driver_globalopts.h"
#include <atomic>
#include <getopt.h>
#include <ctime>
#include <cstdio>
#include <cstring>
#include <cerrno>
#include <climits>
#include <set>

using namespace sta;

namespace {

inline bool dev_captured_run(const DevCapture *c) { return c != nullptr; }

struct Conf {
    sta_mplp_params p{};
    std::string reg, fai_fname, output_fname;
    std::unique_ptr<Fasta> fai;
    std::unique_ptr<Bed> bed;
    std::set<std::string> rg_excl; bool has_rg_excl = false;
    std::vector<std::string> tags; char sep = ',', empty = '*';      // --output-extra aux tags, --output-sep, --output-empty
    int64_t window_cols = 1 << 20, max_reads = 4 << 20;
};

// sample.c:79-122 (bam_smpl_add): number of distinct samples for "[mpileup] N samples in M input files"
struct Samples {
    std::set<std::string> rg, sm;
    void add_pair(const std::string &k, const std::string &v) { if (rg.count(k)) return; rg.insert(k); sm.insert(v); }
    void add(const std::string &fn, const std::string *txt)
    {
        if (!txt) { add_pair(fn, fn); return; }
        size_t p = 0; int n = 0; std::string first_sm; bool have_first = false;
        for (;;) {
            size_t q = txt->find("@RG", p);
            if (q == std::string::npos) break;
            p = q + 3;
            size_t qi = txt->find("\tID:", p), ri = txt->find("\tSM:", p);
            if (qi == std::string::npos || ri == std::string::npos) break;
            qi += 4; ri += 4;
            size_t qe = txt->find_first_of("\t\n", qi), re = txt->find_first_of("\t\n", ri);
            std::string id = txt->substr(qi, qe == std::string::npos ? std::string::npos : qe - qi);
            std::string smv = txt->substr(ri, re == std::string::npos ? std::string::npos : re - ri);
            add_pair(fn + "/" + id, smv);
            if (!have_first) { first_sm = smv; have_first = true; }
            p = std::max(qi, ri);
            ++n;
        }
        if (n == 0) add_pair(fn, fn);
        else if (n == 1 && have_first) add_pair(fn, first_sm);
    }
};

struct Runner {
    Conf &conf;
    DevEngines &devs;                                 // one engine per device thread (started by sta_main_mpileup before the options are read)
    std::atomic<bool> no_device{false};
    std::vector<std::unique_ptr<AlnReader>> readers;
    const Header *h = nullptr;
    FILE *out = driver_default_out();
    DevCapture *dev_cap = driver_dev_capture();       // (sta_main_capture_device: window text stays on the device)
    bool has_reg = false; int tid0 = 0; int64_t beg0 = 0, end0 = INT64_MAX;
    std::atomic<bool> ring_asked{false};
    std::unique_ptr<WinPipe> pipe;                    // producer (this thread) -> device thread -> writer thread
    std::vector<std::vector<StagedFile>> no_reads_d;  // per engine: read-less windows (zero-depth rows); its device thread only
    std::vector<std::vector<char>> cap_dropped;      // per file: reads of the last window that the -d cap dropped
    // host side of mplp_get_ref (the pump's lookahead asks which contig's FASTA is "loaded" and how long it is); producer thread
    int loaded_ref_tid = -2;
    int64_t loaded_ref_len = INT64_MAX;               // length of the FASTA contig (INT64_MAX: none, no length filter)
    std::vector<int> dev_ref_tid;                     // per engine: contig whose sequence is in its HBM; its device thread only
    bool shard_done = false;            // the block's last column has been passed: the rest of the input is not read
    int64_t win_cols = 0; bool adaptive_windows = true;  // columns of the next data window (widened for sparse input unless STA_WINDOW_COLS fixes it)
    Shard shard;                                      // STA_SHARD=rank/world: this rank's block of the columns (driver_shard.h)
    WindowSource *src = nullptr;                      // the input lane (device_stage's resolver asks it; set by run())
    std::vector<int64_t> lin0;                        // linear coordinate of every contig's first column (no region)

    Runner(Conf &c, DevEngines &d) : conf(c), devs(d) {}

    void host_ref(int tid)
    {
        if (!conf.fai || tid == loaded_ref_tid) return;
        loaded_ref_tid = tid;
        const std::string *s = conf.fai->fetch(h->names[(size_t)tid]);
        loaded_ref_len = s ? (int64_t)s->size() : INT64_MAX;
    }

    // device stage of one window (device thread): reference, H2D, plan, emit, D2H of the text
    int device_stage(WinJob &j, int d)
    {
        if (devs.ready() != STA_OK) { if (!no_device.exchange(true)) fprintf(stderr, "samtools mpileup: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return -1; }
        sta_engine *eng = devs.eng[(size_t)d];
        if (!ring_asked.exchange(true)) { if (!dev_cap) pipe->use_ring(&devs.ring); }      // (made by the engines' thread: complete once ready() has returned)
        std::vector<StagedFile> &no_reads = no_reads_d[(size_t)d];
        const size_t nf = readers.size();
        if (conf.fai && j.tid != dev_ref_tid[(size_t)d]) {
            sta_clear_references(eng);
            dev_ref_tid[(size_t)d] = j.tid;
            const std::string *s = conf.fai->fetch(h->names[(size_t)j.tid]);
            if (s && sta_set_reference(eng, j.tid, s->data(), (int64_t)s->size(), STA_MEM_HOST) != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; }
        }
        std::vector<sta_reads> views(nf);
        if (!j.have_reads && no_reads.size() != nf) { no_reads.assign(nf, StagedFile()); for (auto &e : no_reads) e.finish(); }
        for (size_t f = 0; f < nf; ++f) views[f] = j.have_reads ? j.staged[f].view() : no_reads[f].view();
        sta_window w; memset(&w, 0, sizeof w);
        w.tid = j.tid; w.origin = j.cb; w.col_beg = 0; w.col_end = (int32_t)(j.ce - j.cb);
        w.tname = h->names[(size_t)j.tid].c_str(); w.tlen = h->lens[(size_t)j.tid];
        w.n_files = (int32_t)nf; w.files = views.data(); w.mem = STA_MEM_HOST;
        const Bed::Ivals *iv = conf.bed ? conf.bed->get(h->names[(size_t)j.tid]) : nullptr;
        static const int64_t none = 0;
        if (conf.bed) { w.has_bed = 1; w.n_bed = iv ? (int64_t)iv->beg.size() : 0; w.bed_beg = iv ? iv->beg.data() : &none; w.bed_end = iv ? iv->end.data() : &none; }
        if (has_reg) { w.has_reg = 1; w.reg_beg = beg0; w.reg_end = end0; }
        if (sta_stage_window(eng, &w) != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; }
        sta_mplp_params p = conf.p;
        p.all = j.all_mode;
        // a window whose overlap pairs wait for the device's verdict on its reads (-C, a -d cap that can trigger): the plan calls back into
        // the input lane -- whose thread is waiting for this window (lockstep) -- in front of its overlap pass
        if (j.resolve_mates && j.have_reads && src)
            sta_set_mate_resolver(eng, [](void *user, int32_t file, const uint32_t *state, int64_t n, int32_t *mate_out) { ((WindowSource *)user)->pair_from_info((size_t)file, state, n, mate_out); return 0; }, src);
        const int plan_rc = sta_mpileup_plan(eng, &p, &j.info);
        sta_set_mate_resolver(eng, nullptr, nullptr);
        if (plan_rc != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; }
        j.out_bytes = 0;
        j.read_info.clear();
        if (j.info.n_maxcnt_dropped && j.have_reads && !j.lockstep) {
            // safety net behind cap_may_trigger(): the producer has already moved on with reads the iterator never stored
            fprintf(stderr, "samtools mpileup: internal error: the -d cap removed reads near %s:%lld in a window the producer did not wait for\n",
                    h->names[(size_t)j.tid].c_str(), (long long)j.cb + 1);
            return -1;
        }
        if (j.info.n_maxcnt_dropped && j.have_reads) {
            // the cap removed reads from the iterator: the producer takes them out of the pump (info bit 0 = reached bam_plp_push,
            // bit 1 = in the pileup)
            j.read_info.resize(nf);
            for (size_t f = 0; f < nf; ++f) {
                j.read_info[f].resize((size_t)j.staged[f].n());
                if (j.read_info[f].empty()) continue;
                if (sta_fetch_read_state(eng, (int32_t)f, j.read_info[f].data(), nullptr) != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; }
            }
        }
        if (!j.write || j.info.out_bytes == 0) return 0;
        if (dev_cap) {
            // device capture: the rows are written right behind the text captured so far and stay on the device
            char *dst = dev_cap->reserve((size_t)j.info.out_bytes);
            if (!dst || sta_mpileup_emit(eng, dst, j.info.out_bytes) != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", dst ? sta_last_error(eng) : "no device memory for the captured text"); return -1; }
            dev_cap->len += (size_t)j.info.out_bytes;
            return 0;
        }
        if (sta_mpileup_emit(eng, nullptr, 0) != STA_OK) { fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; }
        { const int frc = fetch_text(*pipe, j, eng, j.info.out_bytes); if (frc) { if (frc == -2) fprintf(stderr, "samtools mpileup: %s\n", sta_last_error(eng)); return -1; } }
        return 0;
    }

    // -C with overlap detection: sam_cap_mapq reads BAQ-adjusted qualities, so only the device knows who reaches bam_plp_push
    bool mates_on_device() const { return (conf.p.flag & STA_MPLP_SMART_OVERLAPS) && conf.fai && conf.p.capQ_thres > 10; }

    // the part of [lo, hi) of contig tid this rank prints (everything without STA_SHARD)
    void owned(int tid, int64_t lo, int64_t hi, int64_t *pb, int64_t *pe) const
    {
        shard.clip(!shard.on ? 0 : (has_reg ? -beg0 : lin0[(size_t)tid]), lo, hi, pb, pe);
    }

    // zero-depth rows for [a,b) of a contig (print_empty_pileup), produced by read-less windows
    int run_empty(int tid, int64_t a, int64_t b)
    {
        owned(tid, a, b, &a, &b);
        while (a < b) {
            int64_t e = std::min(b, a + conf.window_cols);
            WinJob *j = pipe->acquire();
            j->tid = tid; j->cb = a; j->ce = e; j->have_reads = false; j->all_mode = 1; j->write = true; j->hold = false; j->lockstep = false;
            pipe->submit(j);
            if (pipe->error()) return -1;
            a = e;
        }
        return 0;
    }

    // One contig.  mode 0: only covered columns (no -a); 1: -a, zero-depth rows once the contig has shown a
    // data column; 2: -aa, always.  Equivalent to mpileup()'s last_tid/last_pos bookkeeping (:610-660,
    // :880-910): every contig that prints anything prints its whole [lo, hi_all) range, in contig order.
    int process_tid(WindowSource &pump, int tid, int mode)
    {
        int64_t tlen = h->lens[(size_t)tid];
        { const double th0 = WinPipe::now(); host_ref(tid); pipe->add_part_time(3, WinPipe::now() - th0); }      // tells the pump's lookahead the FASTA length of this contig
        int64_t lo = has_reg ? beg0 : 0;
        int64_t hi_all = has_reg ? std::min(end0, tlen) : tlen;
        int64_t stop = has_reg ? end0 : INT64_MAX;       // no window reaches beyond this column
        int64_t discard_until = INT64_MIN;               // sharded run: windows below this column are planned for their read states only
        if (shard.on) {
            // this rank's part of the contig: walk up to it with the pump's own bookkeeping (no staging, no device), so that the
            // first window inherits exactly the reads the unsharded run carries there
            int64_t pb, pe;
            owned(tid, lo, hi_all, &pb, &pe);
            if (pe <= pb) { pump.skip_to(tid, lo, INT64_MAX, conf.window_cols); pump.drop_tid_carry(); return pump.error() ? -1 : 0; }
            // (-C with overlap detection: who reaches the overlap hash is the device's to say, so the windows in front of the block go
            // through the plan like any other -- in lock-step, their text never made -- instead of being passed over on the host)
            if (pb > lo && mates_on_device()) discard_until = pb;
            else { if (pb > lo) pump.skip_to(tid, lo, pb, conf.window_cols); lo = pb; }
            if (pump.error()) return -1;
            hi_all = pe; stop = std::min(stop, pe);
        }
        bool started = mode == 2;
        if (!win_cols) win_cols = conf.window_cols;
        // (a block that starts inside the contig starts at its first column: carried reads may cover it)
        int64_t cursor = (started || shard.on) ? lo : std::max(lo, pump.next_pos(tid));
        for (;;) {
            bool more = pump.next_pos(tid) != INT64_MAX;
            if (!more && !pump.has_carry()) break;
            if (!started) cursor = std::max(cursor, std::min(pump.carry_next_covered(cursor), pump.next_pos(tid)));   // skip uncovered gap
            else if (cursor < discard_until) cursor = std::max(cursor, std::min(discard_until, std::min(pump.carry_next_covered(cursor), pump.next_pos(tid))));    // (nothing is printed there)
            int64_t ce_target = std::min(cursor + win_cols, stop);
            const bool discard = cursor < discard_until;
            if (discard) ce_target = std::min(ce_target, discard_until);
            if (ce_target <= cursor) {              // past the region / block end: pass over the rest of this contig
                if (shard.on && !has_reg) { shard_done = true; pump.drop_tid_carry(); break; }      // ... or stop reading altogether: nothing behind a block is this rank's
                pump.skip_to(tid, cursor, INT64_MAX, conf.window_cols);
                pump.drop_tid_carry();
                break;
            }
            WinJob *j = pipe->acquire();
            int64_t ce;
            { const double t0 = WinPipe::now(); ce = pump.fill_staged(tid, cursor, ce_target, j->staged); pipe->add_fill_time(WinPipe::now() - t0); double dw, sc; pump.producer_split(&dw, &sc); pipe->set_producer_split(dw, sc); }
            if (pump.error()) { pipe->release(j); return -1; }
            if (adaptive_windows) {
                // sparse input (a genome at 1x: ~7 000 reads per 2^20 columns): the per-window fixed cost (uploads, launches, host
                // round trips, ~1.3 ms) would dominate, so windows widen until they hold about 10^5 reads; dense input narrows
                // them again.  Where windows are cut never changes the text (the sharded and 37-column-window tests rely on it).
                // (by the reads a FULL window of this density would hold: the short last window of a contig says nothing)
                int64_t nr = 0;
                for (const StagedFile &sf : j->staged) nr += sf.n();
                const int64_t got_cols = std::max<int64_t>(1, std::min(ce, ce_target) - cursor);
                const double full = (double)nr / (double)got_cols * (double)win_cols;
                static const bool trace = getenv("STA_WINDOW_TRACE") != nullptr;
                if (trace) fprintf(stderr, "[window] tid %d [%lld, %lld) target %lld reads %lld win_cols %lld full %.0f\n", tid, (long long)cursor, (long long)ce, (long long)ce_target, (long long)nr, (long long)win_cols, full);
                if (got_cols >= win_cols / 2) {
                    if (full < 20000 && win_cols < ((int64_t)8 << 20)) win_cols *= 2;          // below ~3x depth
                    else if (full > 1500000 && win_cols > ((int64_t)1 << 18)) win_cols /= 2;
                }
            }
            if (pump.next_pos(tid) == INT64_MAX) {  // last reads of the contig: stop where they stop
                int64_t me = pump.carry_max_end();
                if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
            }
            cap_dropped.clear();
            if (ce > cursor) {
                j->tid = tid; j->cb = cursor; j->ce = ce; j->have_reads = true; j->hold = false;
                // the next window depends on this one's result only if the -d cap can drop reads here (they leave the pump)
                const double tc0 = WinPipe::now();
                bool lockstep = cap_may_trigger(j->staged, conf.p.max_depth, pump);
                const bool cap_lockstep = lockstep;
                pipe->add_part_time(0, WinPipe::now() - tc0);
                // the overlap hash: who reaches bam_plp_push is the host's to say -- unless the -d cap can turn reads away here, or -C drops /
                // re-scores reads on the device; then the device thread asks the lane with the window's read states (sta_set_mate_resolver),
                // and the producer waits for the window
                j->resolve_mates = false;
                if (conf.p.flag & STA_MPLP_SMART_OVERLAPS) {
                    if (lockstep || mates_on_device()) { j->resolve_mates = true; lockstep = true; }
                    else { const double tp0 = WinPipe::now(); pump.pair_staged(j->staged); pipe->add_part_time(1, WinPipe::now() - tp0); }
                }
                if (cap_lockstep && shard.on) {
                    pipe->release(j);
                    fprintf(stderr, "samtools mpileup: the -d depth cap can trigger near %s:%lld, which couples this block to its predecessors; run unsharded or raise -d\n",
                            h->names[(size_t)tid].c_str(), (long long)cursor + 1);
                    return -1;
                }
                bool waited = false;
                j->lockstep = lockstep || (mode == 1 && !started);
                if (mode == 1 && !started) {
                    // -a: nothing of this contig is printed before its first data column is known
                    j->all_mode = 0; j->write = false; j->hold = true;
                    pipe->submit(j);
                    if (pipe->wait(j) < 0) { pipe->release(j); return -1; }
                    waited = true;
                    if (j->info.n_data_cols) {
                        started = true;
                        if (run_empty(tid, lo, cursor) < 0) { pipe->release(j); return -1; }
                        j->all_mode = 1; j->write = true; j->hold = false;
                        pipe->submit(j);
                        if (pipe->wait(j) < 0) return -1;
                    } else pipe->release(j);                        // no column to print (the slot's read_info stays readable below)
                } else {
                    j->all_mode = started ? 1 : 0; j->write = !discard;
                    pipe->submit(j);
                    if (lockstep) { if (pipe->wait(j) < 0) return -1; waited = true; }
                }
                if (waited && !j->read_info.empty()) {
                    // reads the cap removed must not be carried into the next window (bam_plp_push never stored them)
                    cap_dropped.assign(readers.size(), {});
                    for (size_t f = 0; f < j->read_info.size(); ++f) {
                        const std::vector<uint32_t> &inf = j->read_info[f];
                        cap_dropped[f].assign(inf.size(), 0);
                        for (size_t i = 0; i < inf.size(); ++i) cap_dropped[f][i] = (inf[i] & 1u) && !(inf[i] & 2u) && pump.staged_has_span(f, i);
                    }
                }
            } else pipe->release(j);
            if (pipe->error()) return -1;
            for (size_t f = 0; f < cap_dropped.size(); ++f) if (!cap_dropped[f].empty()) pump.drop(f, cap_dropped[f]);
            cap_dropped.clear();
            { const double tr0 = WinPipe::now(); pump.retire(ce); pipe->add_part_time(2, WinPipe::now() - tr0); }
            cursor = std::max(cursor, ce);
        }
        pump.drop_tid_carry();
        if (started && run_empty(tid, cursor, hi_all) < 0) return -1;
        return 0;
    }

    int run()
    {
        PumpConfig pc; pc.window_cols = conf.window_cols; pc.max_reads = conf.max_reads; pc.use_endpos = false; pc.nref_limit = h->nref(); pc.device_pools = true; pc.inflate_device = getenv("STA_DEVICE") ? atoi(getenv("STA_DEVICE")) : 0;
        if (conf.p.flag & STA_MPLP_SMART_OVERLAPS) {
            // HTSlib's overlap hash is sequential over the file: the input lane keeps it itself (host_names.h) and stages, for every
            // record, the staged index of the record whose entry it found (sta_reads.olap_mate).  `pushed` = mplp_func's verdict
            // (bam_plcmd.c:400-461), bit for bit what k_prep_reads computes on the device; what only the device can know -- -C (sam_cap_mapq
            // reads the BAQ-adjusted qualities), a -d cap that triggers -- goes through the engine's resolver instead (device_stage).
            pc.tpl = PumpConfig::TPL_MPLP;
            pc.pushed_on_device = conf.fai && conf.p.capQ_thres > 10;
            const sta_mplp_params pp = conf.p;
            pc.pushed = [this, pp](const Rec &r) {
                if (r.flag & 4) return false;
                if (pp.rflag_require && !(pp.rflag_require & r.flag)) return false;
                if (pp.rflag_filter && (pp.rflag_filter & r.flag)) return false;
                if (conf.bed && pp.all == 0 && !conf.bed->overlap(h->names[(size_t)r.tid], r.pos, r.pos + (r.rlen > 0 ? r.rlen : 1))) return false;
                if (conf.has_rg_excl && !r.rg.empty() && conf.rg_excl.count(r.rg)) return false;
                host_ref(r.tid);
                if (r.pos >= loaded_ref_len) return false;             // "Skipping because ... is outside of ..." (INT64_MAX: the contig is not in the FASTA)
                if ((int)r.mapq < pp.min_mq) return false;
                if ((pp.flag & STA_MPLP_NO_ORPHAN) && (r.flag & 1) && !(r.flag & 2)) return false;
                return true;
            };
        }
        // input lane: chunk slices decoded on several threads, unless per-record host formatting is needed
        // (--output-extra tags / RNEXT, -G) or STA_IO_LANE=rec asks for the record-at-a-time lane
        pc.rg_excl = conf.has_rg_excl ? &conf.rg_excl : nullptr;
        pc.xs_rnext = (conf.p.flag & STA_MPLP_PRINT_RNEXT) != 0; pc.xs_n_tags = (int)conf.tags.size(); pc.xs_empty = conf.empty;
        pc.xs_mods = (conf.p.flag & STA_MPLP_OUTPUT_MODS) != 0;          // MM / ML are evaluated per record while staging (host_mods.cpp)
        const char *lane = getenv("STA_IO_LANE");
        const bool chunked = !pc.rg_excl && !pc.xs_rnext && !pc.xs_mods && !(lane && !strcmp(lane, "rec"));
        std::unique_ptr<WindowSource> src;
        if (chunked) src.reset(new ChunkPump(readers, pc, io_threads_per_input((int)readers.size())));
        else src.reset(new Pump(readers, pc));
        WindowSource &pump = *src;
        this->src = src.get();
        const int all = conf.p.all;
        const int mode = all >= 2 ? 2 : all;
        shard = Shard::from_env();
        if (shard.on) {
            if (mode == 1) { fprintf(stderr, "samtools mpileup: a sharded run (STA_SHARD) supports no single -a: whether a contig is printed depends on every block; use -aa or no -a\n"); return 1; }
            lin0.assign((size_t)h->nref() + 1, 0);
            for (int t = 0; t < h->nref(); ++t) lin0[(size_t)t + 1] = lin0[(size_t)t] + h->lens[(size_t)t];
            shard.set_total(has_reg ? std::max<int64_t>(0, std::min(end0, h->lens[(size_t)tid0]) - beg0) : lin0[(size_t)h->nref()]);
        }
        int next_full = 0;               // -aa without region: contigs below this index are done
        bool did_tid0 = false;
        for (;;) {
            int tid = pump.next_tid();
            if (pump.error()) break;
            if (all >= 2 && !has_reg) {
                int upto = tid < 0 ? h->nref() : tid;
                for (int t = next_full; t < upto; ++t) if (run_empty(t, 0, h->lens[(size_t)t]) < 0) { pipe->drain(); return 1; }
                next_full = tid < 0 ? h->nref() : tid + 1;
            }
            if (tid < 0) break;
            if (shard.on && !has_reg && lin0[(size_t)tid] >= shard.E) break;       // every later contig lies behind this rank's block
            if (has_reg && tid == tid0) did_tid0 = true;
            if (process_tid(pump, tid, mode) < 0) { if (pump.error()) break; pipe->drain(); return 1; }
            if (shard_done) break;
        }
        if (pump.error()) {
            pipe->drain();
            fflush(out);
            fprintf(stderr, "samtools mpileup: %s\n", pump.error_text());
            fprintf(stderr, "samtools mpileup: error reading from input file\n");
            return 1;
        }
        if (all >= 2 && has_reg && !did_tid0)
            if (run_empty(tid0, beg0, std::min(end0, h->lens[(size_t)tid0])) < 0) { pipe->drain(); return 1; }
        return pipe->drain() < 0 ? 1 : 0;
    }
};

void usage(FILE *fp)
{
    fprintf(fp, "\nUsage: samtools mpileup [options] in1.bam [in2.bam [...]]\n"
                "(MI355X engine; options as samtools 1.23.1 mpileup except CRAM input; -X takes BAI indexes)\n");
}

}  // namespace

extern "C" int sta_main_mpileup(int argc, char **argv)
{
    // the HIP runtime and the engine come up on their own thread while the options are read, the FASTA is loaded, the inputs are opened
    // and their decode threads fill the first windows (DevEngines, driver_pipeline.h); declared first = destroyed last
    timeline_mark("main entered");
    DevEngines devs;
    devs.start(getenv("STA_DEVICE") ? atoi(getenv("STA_DEVICE")) : 0);
    if (getenv("STA_DRIVER_TIMING")) sta::report_thread_budget();
    Conf conf;
    sta_mplp_params &mp = conf.p;
    mp.min_baseQ = 13; mp.capQ_thres = 0; mp.max_depth = 8000;
    mp.flag = STA_MPLP_NO_ORPHAN | STA_MPLP_REALN | STA_MPLP_SMART_OVERLAPS;
    mp.rflag_filter = 4 | 256 | 512 | 1024;
    int use_orphan = 0;
    std::string file_list;
    bool ignore_rg = false, has_index_file = false;
    if (const char *e = getenv("STA_WINDOW_COLS")) conf.window_cols = std::max<long long>(1, atoll(e));
    if (const char *e = getenv("STA_WINDOW_READS")) conf.max_reads = std::max<long long>(1, atoll(e));

    GlobalArgs ga;
    static const struct option lopts[] = {
        STA_GLOBAL_OPTIONS('-', 0, '-', '-', 0, '-'),          // bam_plcmd.c:1098: --input-fmt-option, --reference, --write-index, --verbosity
        { "rf", required_argument, NULL, 1 }, { "ff", required_argument, NULL, 2 },
        { "incl-flags", required_argument, NULL, 1 }, { "excl-flags", required_argument, NULL, 2 },
        { "output", required_argument, NULL, 3 },
        { "output-QNAME", no_argument, NULL, 5 }, { "output-qname", no_argument, NULL, 5 },
        { "illumina1.3+", no_argument, NULL, '6' }, { "count-orphans", no_argument, NULL, 'A' },
        { "bam-list", required_argument, NULL, 'b' },
        { "no-BAQ", no_argument, NULL, 'B' }, { "no-baq", no_argument, NULL, 'B' },
        { "adjust-MQ", required_argument, NULL, 'C' }, { "adjust-mq", required_argument, NULL, 'C' },
        { "max-depth", required_argument, NULL, 'd' },
        { "redo-BAQ", no_argument, NULL, 'E' }, { "redo-baq", no_argument, NULL, 'E' },
        { "fasta-ref", required_argument, NULL, 'f' },
        { "exclude-RG", required_argument, NULL, 'G' }, { "exclude-rg", required_argument, NULL, 'G' },
        { "positions", required_argument, NULL, 'l' }, { "region", required_argument, NULL, 'r' },
        { "ignore-RG", no_argument, NULL, 'R' }, { "ignore-rg", no_argument, NULL, 'R' },
        { "min-MQ", required_argument, NULL, 'q' }, { "min-mq", required_argument, NULL, 'q' },
        { "min-BQ", required_argument, NULL, 'Q' }, { "min-bq", required_argument, NULL, 'Q' },
        { "ignore-overlaps-removal", no_argument, NULL, 'x' }, { "disable-overlap-removal", no_argument, NULL, 'x' },
        { "output-mods", no_argument, NULL, 'M' },
        { "output-BP", no_argument, NULL, 'O' }, { "output-bp", no_argument, NULL, 'O' },
        { "output-BP-5", no_argument, NULL, 14 }, { "output-bp-5", no_argument, NULL, 14 },
        { "output-MQ", no_argument, NULL, 's' }, { "output-mq", no_argument, NULL, 's' },
        { "customized-index", no_argument, NULL, 'X' },
        { "reverse-del", no_argument, NULL, 6 }, { "output-extra", required_argument, NULL, 7 },
        { "output-sep", required_argument, NULL, 8 }, { "output-empty", required_argument, NULL, 9 },
        { "no-output-ins", no_argument, NULL, 10 }, { "no-output-ins-mods", no_argument, NULL, 11 },
        { "no-output-del", no_argument, NULL, 12 }, { "no-output-ends", no_argument, NULL, 13 },
        { NULL, 0, NULL, 0 } };

    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    int c;
    while ((c = getopt_long(argc, argv, "Af:r:l:q:Q:RC:Bd:b:o:EG:6OsxXaM", lopts, NULL)) >= 0) {
        switch (c) {
        case 'x': mp.flag &= ~STA_MPLP_SMART_OVERLAPS; break;
        case 1: mp.rflag_require = str2flag(optarg); if (mp.rflag_require < 0) { fprintf(stderr, "Could not parse --rf %s\n", optarg); return 1; } break;
        case 2: mp.rflag_filter = str2flag(optarg); if (mp.rflag_filter < 0) { fprintf(stderr, "Could not parse --ff %s\n", optarg); return 1; } break;
        case 3: case 'o': conf.output_fname = optarg; break;
        case 5: mp.flag |= STA_MPLP_PRINT_QNAME; break;
        case 6: mp.rev_del = 1; break;
        case 7: {
            // build_auxlist (bam_plcmd.c:240-287): fixed column names set flag bits, any other two-character name is an aux tag
            static const struct { const char *n; int f; } cols[] = {
                { "QNAME", STA_MPLP_PRINT_QNAME }, { "FLAG", STA_MPLP_PRINT_FLAG }, { "RNAME", STA_MPLP_PRINT_RNAME },
                { "POS", STA_MPLP_PRINT_POS }, { "MAPQ", STA_MPLP_PRINT_MAPQ }, { "RNEXT", STA_MPLP_PRINT_RNEXT },
                { "PNEXT", STA_MPLP_PRINT_PNEXT }, { "RLEN", STA_MPLP_PRINT_RLEN } };
            std::string s = optarg; size_t p = 0;
            while (p <= s.size()) {
                size_t e = s.find(',', p); if (e == std::string::npos) e = s.size();
                std::string tag = s.substr(p, e - p);
                bool hit = false;
                for (auto &cn : cols) if (tag == cn.n) { mp.flag |= cn.f; hit = true; }
                // build_auxlist (bam_plcmd.c:270-282): anything else with two characters is an aux tag
                if (!hit && !tag.empty()) {
                    if (tag.size() != 2) fprintf(stderr, "[build_auxlist] tag '%s' has more than two characters or not supported\n", tag.c_str());
                    else conf.tags.push_back(tag);
                }
                p = e + 1;
            }
            break;
        }
        case 8: conf.sep = optarg[0]; break;
        case 9: conf.empty = optarg[0]; break;
        case 10: mp.no_ins++; break;
        case 11: mp.no_ins_mods = 1; break;
        case 12: mp.no_del++; break;
        case 13: mp.no_ends = 1; break;
        case 'f':
            conf.fai = Fasta::load(optarg);
            if (!conf.fai) { fprintf(stderr, "[E::fai_load] failed to load %s\n", optarg); return 1; }
            conf.fai_fname = optarg; mp.has_fai = 1;
            break;
        case 'd': mp.max_depth = atoi(optarg); break;
        case 'r': conf.reg = optarg; break;
        case 'l':
            conf.bed = Bed::load(optarg);
            if (!conf.bed) { fprintf(stderr, "samtools mpileup: Could not read file \"%s\"\n", optarg); return 1; }
            break;
        case 'B': mp.flag &= ~STA_MPLP_REALN; break;
        case 'X': has_index_file = true; break;          // --customized-index: the second half of the file arguments names the indexes (bam_plcmd.c:1243-1262)
        case 'E': mp.flag |= STA_MPLP_REDO_BAQ; break;
        case '6': mp.flag |= STA_MPLP_ILLUMINA13; break;
        case 'R': ignore_rg = true; break;
        case 's': mp.flag |= STA_MPLP_PRINT_MAPQ_CHAR; break;
        case 'O': mp.flag |= STA_MPLP_PRINT_QPOS; break;
        case 14: mp.flag |= STA_MPLP_PRINT_QPOS5; break;
        case 'M': mp.flag |= STA_MPLP_OUTPUT_MODS; break;
        case 'C': mp.capQ_thres = atoi(optarg); break;
        case 'q': mp.min_mq = atoi(optarg); break;
        case 'Q': mp.min_baseQ = atoi(optarg); break;
        case 'b': file_list = optarg; break;
        case 'A': use_orphan = 1; break;
        case 'G': {
            conf.has_rg_excl = true;
            FILE *fp = fopen(optarg, "r");
            if (!fp) { fprintf(stderr, "[%s] Fail to open file %s. Continue anyway.\n", "bam_mpileup", optarg); break; }
            char buf[1024];
            while (fscanf(fp, "%1023s", buf) > 0) conf.rg_excl.insert(buf);
            fclose(fp);
            break;
        }
        case 'a': mp.all++; break;
        default:
            if (c != '?' && parse_global_opt(c, optarg, lopts, &ga) == 0) break;
            usage(stderr);
            return 1;
        }
    }
    if (!conf.fai && !ga.reference.empty()) {
        // bam_plcmd.c:1223-1227: --reference names the FASTA when -f did not
        conf.fai = Fasta::load(ga.reference);
        if (!conf.fai) { fprintf(stderr, "[E::fai_load] failed to load %s\n", ga.reference.c_str()); return 1; }
        conf.fai_fname = ga.reference; mp.has_fai = 1;
    }
    if (!(mp.flag & STA_MPLP_REALN) && (mp.flag & STA_MPLP_REDO_BAQ)) { fprintf(stderr, "Error: The -B option cannot be combined with -E\n"); return 1; }
    if (use_orphan) mp.flag &= ~STA_MPLP_NO_ORPHAN;
    if (argc == 1) { usage(stderr); return 1; }
    std::vector<std::string> fns;
    std::vector<std::string> idx_fns;
    if (!file_list.empty()) {
        if (has_index_file) { fprintf(stderr, "Error: The -b option cannot be combined with -X\n"); return 1; }
        if (!read_file_list(file_list, &fns)) { fprintf(stderr, "No files read from %s\n", file_list.c_str()); return 1; }
    } else if (has_index_file) {
        if ((argc - optind) % 2 != 0) { fprintf(stderr, "Odd number of filenames detected! Each BAM file should have an index file\n"); return 1; }
        const int nf = (argc - optind) / 2;
        for (int i = 0; i < nf; ++i) { fns.push_back(argv[optind + i]); idx_fns.push_back(argv[optind + nf + i]); }
    } else for (int i = optind; i < argc; ++i) fns.push_back(argv[i]);
    if (fns.empty()) { fprintf(stderr, "[mpileup] no input file/data given\n"); return 1; }

    mp.n_tags = (int32_t)conf.tags.size(); mp.tag_sep = conf.sep;
    Runner run(conf, devs);
    run.adaptive_windows = getenv("STA_WINDOW_COLS") == nullptr;
    Samples sm;
    driver_pin_policy(fns);
    for (auto &fn : fns) {
        std::string err;
        auto r = AlnReader::open(fn, &err, io_threads_per_input((int)fns.size()));
        if (!r) { fprintf(stderr, "[mpileup] failed to open %s: %s\n", fn.c_str(), strerror(errno ? errno : ENOENT)); return 1; }
        sm.add(fn, ignore_rg ? nullptr : &r->header().text);
        if (!conf.tags.empty()) r->set_wanted_tags(conf.tags);
        run.readers.push_back(std::move(r));
    }
    run.h = &run.readers[0]->header();
    if (!conf.reg.empty()) {
        for (size_t i = 0; i < run.readers.size(); ++i) {
            int t; int64_t b, e;
            if (!parse_region(run.readers[i]->header(), conf.reg, &t, &b, &e)) {
                fprintf(stderr, "[E::mpileup] fail to parse region '%s' with %s\n", conf.reg.c_str(), fns[i].c_str());
                return 1;
            }
            run.readers[i]->set_region(t, b, e);
            if (i == 0) { run.has_reg = true; run.tid0 = t; run.beg0 = b; run.end0 = e; }
        }
    }
    seek_readers_by_index(run.readers, fns, *run.h, run.has_reg, run.tid0, run.beg0, run.end0, (int64_t)1 << 20, has_index_file ? &idx_fns : nullptr);      // region / sharded runs start at their first column
    fprintf(stderr, "[mpileup] %d samples in %d input files\n", (int)sm.sm.size(), (int)fns.size());
    if (!conf.output_fname.empty() && run.dev_cap) {
        // device capture keeps the window text on the device for the caller: a command that names its own output file would be left with
        // an empty file and exit status 0 -- refused; samtools_amd/shard.py strips -o before it captures
        fprintf(stderr, "[mpileup] -o cannot be combined with device capture (sta_main_capture_device): the text is handed to the caller\n");
        return 1;
    }
    if (!conf.output_fname.empty()) {
        run.out = fopen(conf.output_fname.c_str(), "w");
        if (!run.out) { fprintf(stderr, "[mpileup] failed to write to %s: %s\n", conf.output_fname.c_str(), strerror(errno)); return 1; }
    }
    if (!mp.max_depth) { mp.max_depth = INT_MAX; fprintf(stderr, "[mpileup] Max depth set to maximum value (%d)\n", INT_MAX); }
    else if ((long long)mp.max_depth * (long long)fns.size() > 1 << 20) fprintf(stderr, "[mpileup] Combined max depth is above 1M. Potential memory hog!\n");

    run.dev_ref_tid.assign((size_t)run.devs.n(), -2); run.no_reads_d.resize((size_t)run.devs.n());
    timeline_mark("options read, FASTA loaded, inputs open");
    int ret;
    {
        run.pipe.reset(new WinPipe(pipe_slots_from_env(run.devs.n()), [&run](WinJob &j, int d) { return run.device_stage(j, d); }, run.out, "Failed to write pileup data.\n", run.devs.n()));
        ret = run.run();
        timeline_mark("last window submitted and drained");
        if (!dev_captured_run(run.dev_cap)) {
            // a command-line run ends here (main.cpp): drain() has seen every window written
            if (run.devs.ready() != STA_OK) { if (!run.no_device.exchange(true)) fprintf(stderr, "samtools mpileup: no usable HIP device (the MI355X engine has no CPU fallback)\n"); ret = 1; }
            fflush(run.out);
            driver_exit_now_if_asked(ret, driver_out_is_borrowed(run.out) ? nullptr : run.out);
        }
        run.pipe.reset();                 // joins the device and writer threads (everything is written)
        timeline_mark("pipeline threads joined");
    }
    fflush(run.out);
    if (!driver_out_is_borrowed(run.out)) fclose(run.out);
    if (run.devs.ready() != STA_OK) { if (!run.no_device.exchange(true)) fprintf(stderr, "samtools mpileup: no usable HIP device (the MI355X engine has no CPU fallback)\n"); ret = 1; }
    run.devs.destroy();
    timeline_mark("engines destroyed");
    return ret;
}
