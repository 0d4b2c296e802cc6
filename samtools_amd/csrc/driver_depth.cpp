// driver_depth.cpp -- `samtools depth` command driver on top of the device engine.
//
// Mirrors main_depth()/fastdepth_core() of the reference (bam2depth.c:732-1006, :486-699): same
// options, same rows.  The N-file merge by (tid,pos) (:578-596) becomes "every window receives the
// reads of every file", the ring histogram + row flushing (:209-477) becomes the difference /
// scan / format kernels of the engine, and the -a/-aa zero_region() calls (:88-118, :246-287) are
// restated per contig below.
#include "host_io.h"
#include "host_stage.h"
#include "host_pump.h"
#include "host_chunk.h"
#include "host_bgzf.h"
#include "driver_pipeline.h"
#include "driver_shard.h"
#include "driver_globalopts.h"
#include <cstdlib>
#include <atomic>
#include <getopt.h>
#include <cstdio>
#include <cstring>
#include <cerrno>
#include <climits>

using namespace sta;

namespace {

struct DRunner {
    sta_depth_params p{};
    DevEngines devs;                    // one engine per device thread
    std::atomic<bool> no_device{false};
    std::vector<std::unique_ptr<AlnReader>> readers;
    const Header *h = nullptr;
    FILE *out = driver_default_out();
    DevCapture *dev_cap = driver_dev_capture();       // (sta_main_capture_device: window text stays on the device)
    std::unique_ptr<Bed> bed;
    bool has_reg = false; int tid0 = 0; int64_t beg0 = 0, end0 = INT64_MAX;
    int64_t window_cols = 1 << 20, max_reads = 4 << 20;
    std::atomic<bool> ring_asked{false};
    std::unique_ptr<WinPipe> pipe;      // producer (this thread) -> device thread -> writer thread (driver_pipeline.h)
    std::vector<std::vector<StagedFile>> no_reads_d;   // per engine: read-less windows; its device thread only
    bool shard_done = false;            // the block's last column has been passed: the rest of the input is not read
    int64_t win_cols = 0; bool adaptive_windows = true;   // columns of the next data window (widened for sparse input unless STA_WINDOW_COLS fixes it)
    Shard shard;                        // STA_SHARD=rank/world: this rank's block of the columns (driver_shard.h)
    std::vector<int64_t> lin0;          // linear coordinate of every contig's first column (no region)

    void owned(int tid, int64_t lo, int64_t hi, int64_t *pb, int64_t *pe) const
    {
        shard.clip(!shard.on ? 0 : (has_reg ? -beg0 : lin0[(size_t)tid]), lo, hi, pb, pe);
    }

    // device stage of one window (device thread): H2D, the depth kernels, D2H of the rows
    int device_stage(WinJob &j, int d)
    {
        if (devs.ready() != STA_OK) { if (!no_device.exchange(true)) fprintf(stderr, "samtools depth: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return -1; }
        sta_engine *eng = devs.eng[(size_t)d];
        if (!ring_asked.exchange(true)) { if (!dev_cap) pipe->use_ring(&devs.ring); }      // (made by the engines' thread: complete once ready() has returned)
        std::vector<StagedFile> &no_reads = no_reads_d[(size_t)d];
        size_t nf = readers.size();
        std::vector<sta_reads> views(nf);
        if (!j.have_reads && no_reads.size() != nf) { no_reads.assign(nf, StagedFile()); for (auto &e : no_reads) e.finish(); }
        for (size_t f = 0; f < nf; ++f) views[f] = j.have_reads ? j.staged[f].view() : no_reads[f].view();
        sta_window w; memset(&w, 0, sizeof w);
        w.tid = j.tid; w.origin = j.cb; w.col_beg = 0; w.col_end = (int32_t)(j.ce - j.cb);
        w.tname = h->names[(size_t)j.tid].c_str(); w.tlen = h->lens[(size_t)j.tid];
        w.n_files = (int32_t)nf; w.files = views.data(); w.mem = STA_MEM_HOST;
        const Bed::Ivals *iv = bed ? bed->get(h->names[(size_t)j.tid]) : nullptr;
        static const int64_t none = 0;
        if (bed) { w.has_bed = 1; w.n_bed = iv ? (int64_t)iv->beg.size() : 0; w.bed_beg = iv ? iv->beg.data() : &none; w.bed_end = iv ? iv->end.data() : &none; }
        if (sta_stage_window(eng, &w) != STA_OK) { fprintf(stderr, "samtools depth: %s\n", sta_last_error(eng)); return -1; }
        sta_depth_params pp = p;
        pp.all_pos = j.all_mode;
        j.out_bytes = 0;
        if (sta_depth_plan(eng, &pp, &j.info) != STA_OK) { fprintf(stderr, "samtools depth: %s\n", sta_last_error(eng)); return -1; }
        if (!j.write || j.info.out_bytes == 0) return 0;
        if (dev_cap) {
            // device capture: the window's rows are copied (device to device) behind the text captured so far
            char *dst = dev_cap->reserve((size_t)j.info.out_bytes);
            if (!dst || sta_depth_emit(eng, dst, j.info.out_bytes) != STA_OK || sta_sync(eng) != STA_OK) { fprintf(stderr, "samtools depth: %s\n", dst ? sta_last_error(eng) : "no device memory for the captured text"); return -1; }
            dev_cap->len += (size_t)j.info.out_bytes;
            return 0;
        }
        { const int frc = fetch_text(*pipe, j, eng, j.info.out_bytes); if (frc) { if (frc == -2) fprintf(stderr, "samtools depth: %s\n", sta_last_error(eng)); return -1; } }
        return 0;
    }

    int run_empty(int tid, int64_t a, int64_t b)      // zero_region
    {
        owned(tid, a, b, &a, &b);
        while (a < b) {
            int64_t e = std::min(b, a + window_cols);
            WinJob *j = pipe->acquire();
            j->tid = tid; j->cb = a; j->ce = e; j->have_reads = false; j->all_mode = 1; j->write = true; j->hold = false;
            pipe->submit(j);
            if (pipe->error()) return -1;
            a = e;
        }
        return 0;
    }

    // mode 0: covered rows only; 1: -a (zero rows once a read of this contig passed the filters); 2: always
    int process_tid(WindowSource &pump, int tid, int mode)
    {
        int64_t tlen = h->lens[(size_t)tid];
        int64_t lo = has_reg ? beg0 : 0;
        int64_t hi_all = has_reg ? std::min(end0, tlen) : tlen;
        int64_t stop = has_reg ? end0 : INT64_MAX;       // no window reaches beyond this column
        if (shard.on) {
            // this rank's part of the contig: walk up to it with the pump's own bookkeeping (no staging, no device)
            int64_t pb, pe;
            owned(tid, lo, hi_all, &pb, &pe);
            if (pe <= pb) { pump.skip_to(tid, lo, INT64_MAX, window_cols); pump.drop_tid_carry(); return pump.error() ? -1 : 0; }
            if (pb > lo) pump.skip_to(tid, lo, pb, window_cols);
            if (pump.error()) return -1;
            lo = pb; hi_all = pe; stop = std::min(stop, pe);
        }
        bool started = mode == 2;
        // (a block that starts inside the contig starts at its first column: carried reads may cover it)
        int64_t cursor = (started || shard.on) ? lo : std::max(lo, pump.next_pos(tid));
        for (;;) {
            bool more = pump.next_pos(tid) != INT64_MAX;
            if (!more && !pump.has_carry()) break;
            if (!started && !pump.has_carry()) cursor = std::max(cursor, pump.next_pos(tid));
            if (!win_cols) win_cols = window_cols;
            int64_t ce_target = std::min(cursor + win_cols, stop);
            if (ce_target <= cursor) {
                if (shard.on && !has_reg) { shard_done = true; pump.drop_tid_carry(); break; }      // nothing behind a block is this rank's: stop reading
                pump.skip_to(tid, cursor, INT64_MAX, window_cols); pump.drop_tid_carry(); break;
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
            if (pump.next_pos(tid) == INT64_MAX) {
                int64_t me = pump.carry_max_end();
                if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
            }
            if (ce > cursor) {
                j->tid = tid; j->cb = cursor; j->ce = ce; j->have_reads = true; j->hold = false;
                if (mode == 1 && !started) {
                    // -a: nothing of this contig is printed before a read of it is known to pass the filters
                    j->all_mode = 0; j->write = false; j->hold = true;
                    pipe->submit(j);
                    if (pipe->wait(j) < 0) { pipe->release(j); return -1; }
                    if (j->info.n_kept_reads) {
                        started = true;
                        if (run_empty(tid, lo, cursor) < 0) { pipe->release(j); return -1; }
                        j->all_mode = 1; j->write = true; j->hold = false;
                        pipe->submit(j);
                    } else pipe->release(j);
                } else {
                    j->all_mode = started ? 1 : 0; j->write = true;
                    pipe->submit(j);
                }
            } else pipe->release(j);
            if (pipe->error()) return -1;
            pump.retire(ce);
            cursor = std::max(cursor, ce);
        }
        pump.drop_tid_carry();
        if (started && run_empty(tid, cursor, hi_all) < 0) return -1;
        return 0;
    }

    int run()
    {
        PumpConfig pc; pc.window_cols = window_cols; pc.max_reads = max_reads; pc.use_endpos = true; pc.nref_limit = h->nref(); pc.device_pools = true; pc.inflate_device = getenv("STA_DEVICE") ? atoi(getenv("STA_DEVICE")) : 0;
        // -s: the name hash of bam2depth.c:598-623 is sequential over the whole file; the input lane keeps it itself, in file order, and
        // stages every record's clip column (host_names.h; sta_reads.olap_clip) -- no window depends on records it does not stage
        if (p.remove_overlaps) {
            pc.tpl = PumpConfig::TPL_DEPTH;
            pc.depth_filter.flag = p.flag; pc.depth_filter.incl_flag = p.incl_flag; pc.depth_filter.require_flag = p.require_flag;
            pc.depth_filter.min_mqual = p.min_mqual; pc.depth_filter.min_len = p.min_len;
        }
        const char *lane = getenv("STA_IO_LANE");
        std::unique_ptr<WindowSource> src;
        if (lane && !strcmp(lane, "rec")) src.reset(new Pump(readers, pc));          // record-at-a-time lane
        else src.reset(new ChunkPump(readers, pc, io_threads_per_input((int)readers.size())));           // chunk slices decoded on several threads
        WindowSource &pump = *src;
        const int all = p.all_pos;
        // with a region every -a/-aa run prints the whole region of tid0 (bam2depth.c:267-270)
        const int mode = has_reg ? (all ? 2 : 0) : (all >= 2 ? 2 : all);
        shard = Shard::from_env();
        if (shard.on) {
            if (mode == 1) { fprintf(stderr, "samtools depth: a sharded run (STA_SHARD) supports no single -a: whether a contig is printed depends on every block; use -aa or no -a\n"); return 1; }
            lin0.assign((size_t)h->nref() + 1, 0);
            for (int t = 0; t < h->nref(); ++t) lin0[(size_t)t + 1] = lin0[(size_t)t] + h->lens[(size_t)t];
            shard.set_total(has_reg ? std::max<int64_t>(0, std::min(end0, h->lens[(size_t)tid0]) - beg0) : lin0[(size_t)h->nref()]);
        }
        int next_full = 0; bool did_tid0 = false;
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
            if (pump.error() == -2) fprintf(stderr, "samtools depth: Data is not position sorted\n");
            else fprintf(stderr, "samtools depth: %s\n", pump.error_text());
            return 1;
        }
        if (all && has_reg && !did_tid0)
            if (run_empty(tid0, beg0, std::min(end0, h->lens[(size_t)tid0])) < 0) { pipe->drain(); return 1; }
        return pipe->drain() < 0 ? 1 : 0;
    }
};

void usage_exit(FILE *fp)
{
    fprintf(fp, "Usage: samtools depth [options] in.bam [in.bam ...]\n"
                "(MI355X engine; options as samtools 1.23.1 depth except CRAM input; -X takes BAI indexes)\n");
}

}  // namespace

extern "C" int sta_main_depth(int argc, char **argv)
{
    timeline_mark("main entered");
    DRunner run;
    run.devs.start(getenv("STA_DEVICE") ? atoi(getenv("STA_DEVICE")) : 0);
    if (getenv("STA_DRIVER_TIMING")) sta::report_thread_budget();      // the runtime comes up while the inputs are opened and decoded
    sta_depth_params &opt = run.p;
    opt.flag = 4 | 256 | 1024 | 512;
    opt.skip_del = 1;
    std::string file_list, out_file, reg;
    bool header = false, has_index_file = false;
    if (const char *e = getenv("STA_WINDOW_COLS")) { run.window_cols = std::max<long long>(1, atoll(e)); run.adaptive_windows = false; }
    if (const char *e = getenv("STA_WINDOW_READS")) run.max_reads = std::max<long long>(1, atoll(e));

    GlobalArgs ga;
    static const struct option lopts[] = {
        STA_GLOBAL_OPTIONS('-', 0, '-', '-', 0, '@'),          // bam2depth.c:765: --input-fmt-option, --reference, --threads / -@, --write-index, --verbosity
        { "min-MQ", required_argument, NULL, 'Q' }, { "min-mq", required_argument, NULL, 'Q' },
        { "min-BQ", required_argument, NULL, 'q' }, { "min-bq", required_argument, NULL, 'q' },
        { "excl-flags", required_argument, NULL, 'G' }, { "incl-flags", required_argument, NULL, 1 },
        { "require-flags", required_argument, NULL, 2 },
        { NULL, 0, NULL, 0 } };
    optind = 0;          // (glibc: 0 = full re-initialisation; with 1 a second in-process call resumes at a stale pointer into the PREVIOUS argv)
    int c, tmp;
    while ((c = getopt_long(argc, argv, "@:q:Q:JHd:m:l:g:G:o:ar:Xf:b:s", lopts, NULL)) >= 0) {
        switch (c) {
        case 'a': opt.all_pos++; break;
        case 'b':
            run.bed = Bed::load(optarg);
            if (!run.bed) { fprintf(stderr, "samtools depth: Could not read file \"%s\"\n", optarg); return 1; }
            break;
        case 'f': file_list = optarg; break;
        case 'd': case 'm': break;
        case 'g': tmp = str2flag(optarg); if (tmp < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.flag &= ~tmp; break;
        case 'G': tmp = str2flag(optarg); if (tmp < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.flag |= tmp; break;
        case 1: tmp = str2flag(optarg); if (tmp < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.incl_flag |= tmp; break;
        case 2: tmp = str2flag(optarg); if (tmp < 0) { fprintf(stderr, "samtools depth: Unknown flag '%s'\n", optarg); return 1; } opt.require_flag |= tmp; break;
        case 'l': opt.min_len = atoi(optarg); break;
        case 'H': header = true; break;
        case 'q': opt.min_qual = atoi(optarg); break;
        case 'Q': opt.min_mqual = atoi(optarg); break;
        case 'J': opt.skip_del = 0; break;
        case 'o': if (out_file.empty()) out_file = optarg; break;
        case 'r': reg = optarg; break;
        case 's': opt.remove_overlaps = 1; break;
        case 'X': has_index_file = true; break;          // the second half of the file arguments names the indexes (bam2depth.c:873-911)
        default:
            // bam2depth.c:877: the global options (-@ / --threads: decompression threads of HTSlib's reader; this engine sizes its own
            // decode threads, STA_IO_THREADS; --reference: CRAM only)
            if (c != '?' && parse_global_opt(c, optarg, lopts, &ga) == 0) break;
            usage_exit(stderr); return 1;
        }
    }
    if (argc < optind + 1 && file_list.empty()) { usage_exit(argc == optind ? stdout : stderr); return argc == optind ? 0 : 1; }
    std::vector<std::string> fns;
    std::vector<std::string> idx_fns;
    if (!file_list.empty()) {
        if (has_index_file) { fprintf(stderr, "samtools depth: The -f option cannot be combined with -X\n"); return 1; }
        if (!read_file_list(file_list, &fns)) return 1;
    } else if (has_index_file) {
        // (the reference tests `nfiles % 1`, which never fires, and then halves: an odd count drops the last name.  An odd count is refused
        // here with the message the reference meant to print.)
        if ((argc - optind) % 2 != 0) { fprintf(stderr, "samtools depth: -X needs one index specified per bam file\n"); return 1; }
        const int nf = (argc - optind) / 2;
        for (int i = 0; i < nf; ++i) { fns.push_back(argv[optind + i]); idx_fns.push_back(argv[optind + nf + i]); }
    } else for (int i = optind; i < argc; ++i) fns.push_back(argv[i]);

    driver_pin_policy(fns);
    for (auto &fn : fns) {
        std::string err;
        auto r = AlnReader::open(fn, &err, io_threads_per_input((int)fns.size()));
        if (!r) { fprintf(stderr, "samtools depth: Cannot open input file \"%s\": %s\n", fn.c_str(), strerror(errno ? errno : ENOENT)); return 1; }
        run.readers.push_back(std::move(r));
    }
    run.h = &run.readers[0]->header();
    if (!reg.empty()) {
        for (size_t i = 0; i < run.readers.size(); ++i) {
            int t; int64_t b, e;
            if (!parse_region(run.readers[i]->header(), reg, &t, &b, &e)) { fprintf(stderr, "samtools depth: cannot parse region \"%s\"\n", reg.c_str()); return 1; }
            run.readers[i]->set_region(t, b, e);
            if (i == 0) { run.has_reg = true; run.tid0 = t; run.beg0 = b; run.end0 = e; }
        }
    }
    seek_readers_by_index(run.readers, fns, *run.h, run.has_reg, run.tid0, run.beg0, run.end0, (int64_t)1 << 20, has_index_file ? &idx_fns : nullptr,
                          !opt.remove_overlaps);      // region / sharded runs start at their first column (-s: a sharded run reads from where the unsharded one starts)
    if (!out_file.empty() && run.dev_cap) {
        // device capture keeps the window text on the device for the caller: a command that names its own output file would be left with
        // an empty file and exit status 0 -- refused; samtools_amd/shard.py strips -o before it captures
        fprintf(stderr, "samtools depth: -o cannot be combined with device capture (sta_main_capture_device): the text is handed to the caller\n");
        return 1;
    }
    if (!out_file.empty()) {
        run.out = fopen(out_file.c_str(), "w");
        if (!run.out) { fprintf(stderr, "samtools depth: Cannot open \"%s\" for writing.\n", out_file.c_str()); return 1; }
    }
    if (header && Shard::from_env().rank == 0) {          // (a sharded run: the header line belongs to the first block)
        fprintf(run.out, "#CHROM\tPOS");
        for (auto &fn : fns) fprintf(run.out, "\t%s", fn.c_str());
        fputc('\n', run.out);
    }
    timeline_mark("options read, inputs open");
    run.no_reads_d.resize((size_t)run.devs.n());
    fflush(run.out);                  // the header line: the writer thread owns the stream from here on
    int ret;
    {
        run.pipe.reset(new WinPipe(pipe_slots_from_env(run.devs.n()), [&run](WinJob &j, int d) { return run.device_stage(j, d); }, run.out, "samtools depth: failed to write the output\n", run.devs.n()));
        ret = run.run();
        timeline_mark("last window submitted and drained");
        if (!run.dev_cap) {
            // a command-line run ends here (main.cpp): drain() has seen every window written
            if (run.devs.ready() != STA_OK) { if (!run.no_device.exchange(true)) fprintf(stderr, "samtools depth: no usable HIP device (the MI355X engine has no CPU fallback)\n"); ret = 1; }
            fflush(run.out);
            driver_exit_now_if_asked(ret, driver_out_is_borrowed(run.out) ? nullptr : run.out);
        }
        run.pipe.reset();
        timeline_mark("pipeline threads joined");
    }
    fflush(run.out);
    if (!driver_out_is_borrowed(run.out)) fclose(run.out);
    // (an input without a single window never asked for the engine: a machine without a device is an error all the same)
    if (run.devs.ready() != STA_OK) { if (!run.no_device.exchange(true)) fprintf(stderr, "samtools depth: no usable HIP device (the MI355X engine has no CPU fallback)\n"); ret = 1; }
    run.devs.destroy();
    timeline_mark("engines destroyed");
    return ret;
}
