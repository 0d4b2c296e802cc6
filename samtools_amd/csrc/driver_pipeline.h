// driver_pipeline.h -- the drivers' three-stage window pipeline.
//
// The reference's column loop is strictly serial: pull records, pile one column, print it (bam_plcmd.c:607-868,
// bam2depth.c:578-699).  Here a window is a job that moves through three stages on three threads,
//     producer (the driver's own thread): pump.fill_staged() -> the job's staging arrays (page-locked)
//     device thread                     : reference upload, sta_stage_window (H2D), plan + emit kernels, D2H of the text
//     writer thread                     : fwrite of the text, in submission order
// so that staging window k+1, computing window k and writing window k-1 overlap.  Jobs live in a small ring of slots (each
// owns its staging arrays and its text buffer); the engine is only ever touched by the device thread, the pump only by the
// producer.  The producer can wait for a job's device stage when the NEXT window depends on its result (the -d cap dropped
// reads; `-a` still waiting for the contig's first data column) -- everything else runs ahead.
#pragma once
#include <unistd.h>
#include <ctime>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include "host_stage.h"
#include "host_pump.h"
#include "../../include/samtools_amd.h"
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <ctime>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace sta {

// STA_DRIVER_TIMING=2: where a whole process's wall time goes outside the window pipeline -- every mark prints the seconds since the
// process was started (its start time from /proc/self/stat, so that loading the executable and the HIP libraries is inside)
inline void timeline_mark(const char *what)
{
    static const int on = [] { const char *e = getenv("STA_DRIVER_TIMING"); return e && atoi(e) >= 2 ? 1 : 0; }();
    if (!on) return;
    static const double t_start = [] {
        double st = -1;
        if (FILE *f = fopen("/proc/self/stat", "r")) {
            char buf[2048]; size_t n = fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; fclose(f);
            if (const char *p = strrchr(buf, ')')) {
                unsigned long long ticks = 0; int field = 2;
                for (const char *q = p + 1; *q && field < 22; ++q) if (*q == ' ') { ++field; if (field == 22) ticks = strtoull(q + 1, nullptr, 10); }
                if (ticks) st = (double)ticks / (double)sysconf(_SC_CLK_TCK);
            }
        }
        return st;
    }();
    timespec ts; clock_gettime(CLOCK_BOOTTIME, &ts);
    const double nowb = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    fprintf(stderr, "[timeline] +%.3f s %s\n", t_start >= 0 ? nowb - t_start : 0.0, what);
}

struct WinJob {
    // set by the producer
    int tid = -1; int64_t cb = 0, ce = 0;
    bool have_reads = false;             // false: a read-less window (zero-depth rows): `staged` is ignored
    int all_mode = 0; bool write = true;
    bool lockstep = false;               // the producer waits for this job's result before it stages the next window (-d cap, single -a)
    bool hold = false;                   // the producer will submit this job again (same staged reads): the slot is not recycled
    bool resolve_mates = false;          // mpileup: the overlap pairs of this window wait for the device's read states (sta_set_mate_resolver; implies lockstep)
    std::vector<StagedFile> staged;
    // set by the device stage
    pvector<char> text; uint64_t out_bytes = 0;
    std::deque<std::pair<int, size_t>> pieces;      // the text ring's pieces that hold this job's text, in order (WinPipe::ring_push; guarded by the pipe's mutex)
    sta_plan_info info{};
    std::vector<std::vector<uint32_t>> read_info;      // per file: engine info words, fetched only when the -d cap dropped reads
    int rc = 0;
    bool ringed = false;                 // this run of the device stage sent the text through the ring (nothing in `text`)
    // pipeline state
    int state = 0;                       // 0 free / with the producer, 1 queued for the device, 2 device done, 3 written
    size_t trace_ix = (size_t)-1;        // STA_DRIVER_TIMING=3
};

// The text ring: a few page-locked pieces through which the device thread fetches a window's text and from which the writer writes it,
// instead of one page-locked buffer of a whole window's text per pipeline slot (80 MB each for `mpileup` at 30x: 20-25 ms of page-locking
// per slot on the device thread, inside the first windows' time -- profiles/r06_sessionG_e2e_window_trace.log).  Allocated once by
// DevEngines when the runtime is up; used when one device thread feeds the writer (pieces then arrive in submission order).
struct TextRing { std::vector<char *> buf; size_t piece = 0; };

class WinPipe {
public:
    // n_dev device threads (each drives its own engine on its own stream: device_fn's second argument says which), so that the
    // H2D / D2H copies of one window overlap the kernels of another; the writer keeps submission order
    WinPipe(size_t n_slots, std::function<int(WinJob &, int)> device_fn, FILE *out, const char *write_error_text, int n_dev = 1)
        : fn_(std::move(device_fn)), out_(out), werr_(write_error_text), slots_(n_slots < (size_t)(n_dev < 1 ? 1 : n_dev) + 2 ? (size_t)(n_dev < 1 ? 1 : n_dev) + 2 : n_slots)
    {
        // (at least three slots: the single `-a` path holds one job while it acquires the next; with one slot it would wait for itself)
        timing_ = getenv("STA_DRIVER_TIMING") != nullptr;
        trace_ = timing_ && atoi(getenv("STA_DRIVER_TIMING")) >= 3;
        t0_ = now();
        if (n_dev < 1) n_dev = 1;
        t_devn_.assign((size_t)n_dev, 0.0);
        for (int d = 0; d < n_dev; ++d) dev_.emplace_back([this, d] { device_loop(d); });
        wr_ = std::thread([this] { writer_loop(); });
    }
    ~WinPipe()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : dev_) if (t.joinable()) t.join();
        if (wr_.joinable()) wr_.join();
        for (double x : t_devn_) t_dev_ = x > t_dev_ ? x : t_dev_;
        if (trace_)
            for (const Trace &t : trace_log_)
                fprintf(stderr, "[window %llu] tid %d cols %lld reads %lld | submitted +%.3f | device +%.3f .. +%.3f (%.1f ms) | written +%.3f | text %.1f MB\n", t.seq, t.tid, (long long)t.cols, (long long)t.reads,
                        t.t_submit - t0_, t.t_dev0 - t0_, t.t_dev1 - t0_, (t.t_dev1 - t.t_dev0) * 1e3, t.t_written - t0_, (double)t.bytes / 1e6);
        if (timing_)
            fprintf(stderr, "[driver timing] wall %.3f s | producer: fill+stage %.3f s (of it: waiting for the decode threads %.3f s, copying slices %.3f s), "
                            "waiting for a slot %.3f s, waiting for results %.3f s, depth-cap bound %.3f s, overlap pairs %.3f s, retiring reads %.3f s, reference %.3f s | device thread busy %.3f s (the busiest of %d) | writer busy %.3f s | %llu windows\n",
                    now() - t0_, t_fill_, t_decode_wait_, t_stage_copy_, t_slot_, t_wait_, t_part_[0], t_part_[1], t_part_[2], t_part_[3], t_dev_, (int)t_devn_.size(), t_wr_, (unsigned long long)n_jobs_);
    }
    // a slot the producer may fill (blocks while all are in flight)
    WinJob *acquire()
    {
        const double a = now();
        std::unique_lock<std::mutex> lk(m_);
        WinJob *j = nullptr;
        cv_.wait(lk, [&] { for (auto &s : slots_) if (s.state == 0 && !s.hold) { j = &s; return true; } return false; });
        j->state = -1;                    // with the producer
        t_slot_ += now() - a;
        return j;
    }
    void release(WinJob *j) { std::lock_guard<std::mutex> g(m_); j->state = 0; j->hold = false; cv_.notify_all(); }   // not submitted after all
    void submit(WinJob *j)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            j->state = 1; j->rc = 0; j->ringed = false; order_.push_back(j); devq_.push_back(j); ++n_jobs_;
            if (trace_) {
                Trace t; t.seq = n_jobs_ - 1; t.tid = j->tid; t.cols = j->ce - j->cb; t.reads = 0; t.t_submit = now();
                if (j->have_reads) for (const StagedFile &sf : j->staged) t.reads += sf.n();
                j->trace_ix = trace_log_.size(); trace_log_.push_back(t);
            }
        }
        cv_.notify_all();
    }
    // device stage of j finished (plan info, text in j->text): returns its rc.  A held job is waited for until the writer has
    // passed it too, so that the producer may submit it again.
    int wait(WinJob *j)
    {
        const double a = now();
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return j->hold ? j->state == 3 : (j->state >= 2 || j->state == 0); });
        t_wait_ += now() - a;
        return j->rc;
    }
    // every submitted job written; returns the first error (<0) or 0
    int drain()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return order_.empty(); });
        return err_;
    }
    int error() { std::lock_guard<std::mutex> g(m_); return err_; }
    // ---- the text ring (one device thread only) ----
    void use_ring(TextRing *r)
    {
        std::lock_guard<std::mutex> g(m_);
        if (!r || r->buf.empty() || !r->piece || t_devn_.size() != 1) return;
        ring_ = r; ring_free_.clear();
        for (size_t i = 0; i < r->buf.size(); ++i) ring_free_.push_back((int)i);
    }
    bool ring_on() { std::lock_guard<std::mutex> g(m_); return ring_ != nullptr; }
    size_t ring_piece() const { return ring_ ? ring_->piece : 0; }
    // a free piece (blocks while the writer holds them all); nullptr after an error
    char *ring_acquire(int *idx)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !ring_free_.empty() || err_.load() || stop_; });
        if (ring_free_.empty()) return nullptr;
        *idx = ring_free_.back(); ring_free_.pop_back();
        return ring_->buf[(size_t)*idx];
    }
    void ring_release(int idx) { { std::lock_guard<std::mutex> g(m_); ring_free_.push_back(idx); } cv_.notify_all(); }
    void ring_push(WinJob *j, int idx, size_t len) { { std::lock_guard<std::mutex> g(m_); j->pieces.emplace_back(idx, len); } cv_.notify_all(); }
    void add_fill_time(double s) { t_fill_ += s; }
    void add_part_time(int what, double s) { if (what >= 0 && what < 4) t_part_[what] += s; }      // producer: 0 depth-cap bound, 1 overlap pairs, 2 retire, 3 reference
    void set_producer_split(double decode_wait, double stage_copy) { t_decode_wait_ = decode_wait; t_stage_copy_ = stage_copy; }
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

private:
    void device_loop(int d)
    {
        for (;;) {
            WinJob *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !devq_.empty(); });
                if (devq_.empty()) return;
                j = devq_.front(); devq_.pop_front();
            }
            const double a = now();
            const int prior = err_.load();
            int rc = prior ? prior : fn_(*j, d);     // after an error the remaining jobs only drain
            const double b = now();
            t_devn_[(size_t)d] += b - a;
            { std::lock_guard<std::mutex> g(m_); j->rc = rc; if (rc < 0 && !err_.load()) err_ = rc; j->state = 2; if (trace_ && j->trace_ix < trace_log_.size()) { Trace &t = trace_log_[j->trace_ix]; t.t_dev0 = a; t.t_dev1 = b; t.bytes = j->out_bytes; } }
            cv_.notify_all();
        }
    }
    void writer_loop()
    {
        for (;;) {
            WinJob *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return (stop_ && order_.empty()) || (!order_.empty() && (order_.front()->state == 2 || !order_.front()->pieces.empty())); });
                if (order_.empty()) return;
                j = order_.front();
                if (!j->pieces.empty()) {
                    // a piece of the front job's text (the job may still be on the device: its text leaves as it arrives)
                    const std::pair<int, size_t> pc = j->pieces.front(); j->pieces.pop_front();
                    const bool ok = !err_.load();
                    lk.unlock();
                    const double a = now();
                    int rc = 0;
                    if (ok && fwrite(ring_->buf[(size_t)pc.first], 1, pc.second, out_) != pc.second) { fprintf(stderr, "%s", werr_); rc = -1; }
                    t_wr_ += now() - a;
                    lk.lock();
                    if (rc < 0 && !err_.load()) err_ = rc;
                    ring_free_.push_back(pc.first);
                    lk.unlock();
                    cv_.notify_all();
                    continue;
                }
            }
            const double a = now();
            int rc = 0;
            if (j->rc >= 0 && j->write && j->out_bytes && !j->ringed && !err_.load()) {
                if (fwrite(j->text.data(), 1, (size_t)j->out_bytes, out_) != (size_t)j->out_bytes) { fprintf(stderr, "%s", werr_); rc = -1; }
            }
            t_wr_ += now() - a;
            {
                std::lock_guard<std::mutex> g(m_);
                if (trace_ && j->trace_ix < trace_log_.size()) trace_log_[j->trace_ix].t_written = now();
                if (rc < 0 && !err_.load()) err_ = rc;
                order_.pop_front();
                j->state = j->hold ? 3 : 0;       // a held job stays with the producer (it waits for state >= 2 and submits again)
            }
            cv_.notify_all();
        }
    }

    std::function<int(WinJob &, int)> fn_;
    FILE *out_; const char *werr_;
    std::deque<WinJob> slots_;
    std::deque<WinJob *> devq_, order_;
    std::mutex m_; std::condition_variable cv_;
    std::vector<std::thread> dev_; std::thread wr_;
    std::vector<double> t_devn_;
    bool stop_ = false; std::atomic<int> err_{0};      // written under m_, read by the stage threads outside it
    struct Trace { unsigned long long seq = 0; int tid = 0; int64_t cols = 0, reads = 0; double t_submit = 0, t_dev0 = 0, t_dev1 = 0, t_written = 0; uint64_t bytes = 0; };
    std::vector<Trace> trace_log_; bool trace_ = false;      // STA_DRIVER_TIMING=3: one line per window at the end
    TextRing *ring_ = nullptr; std::vector<int> ring_free_;
    double t_part_[4] = { 0, 0, 0, 0 };
    bool timing_ = false; double t0_ = 0, t_decode_wait_ = 0, t_stage_copy_ = 0, t_fill_ = 0, t_slot_ = 0, t_wait_ = 0, t_dev_ = 0, t_wr_ = 0; unsigned long long n_jobs_ = 0;
};

// The device side of a driver: n engines (default 1; STA_DEV_THREADS=2..4), each on its own non-blocking stream, one per device thread
// of the WinPipe.  While one engine runs the kernels of a window the other copies its window in or its text out.
// Bringing up the HIP runtime and an engine takes ~0.25 s -- a third of a whole 1 Gbase run -- and needs nothing from the input: start()
// does it on a thread of its own while the driver parses options, loads the FASTA, opens the inputs and its decode threads fill the first
// windows; ready() is what the first user of an engine (a device thread of the WinPipe) waits on.
struct DevEngines {
    std::vector<sta_engine *> eng;
    std::vector<void *> streams;
    TextRing ring;                       // allocated behind the engines by start()'s thread (STA_TEXT_RING=pieces x MiB, default 6x8; 0: none)
    DevEngines() : n_(dev_threads()) {}
    ~DevEngines() { destroy(); }
    void start(int device);              // begins creating n() engines in the background
    int ready();                         // waits for start(); 0, or the sta_engine_create error (STA_ERR_HIP if fewer than n() engines came up)
    void destroy();                      // waits, then destroys what was created
    int n() const { return n_; }         // engines asked for (STA_DEV_THREADS, default 1): known before they exist
private:
    static int dev_threads();
    int create(int device);
    int n_, rc_ = STA_ERR_NO_DEVICE;
    bool started_ = false, joined_ = false;
    std::thread th_; std::mutex m_;
};
int dev_threads_from_env();
size_t pipe_slots_from_env(int n_dev);
void driver_pin_policy(const std::vector<std::string> &paths);

// the device stage's last step in both drivers: the window's text from the engine to the writer -- piece by piece through the ring, or
// (several device threads, no ring) into the job's own buffer
inline int fetch_text(WinPipe &pipe, WinJob &j, sta_engine *eng, uint64_t total)
{
    if (pipe.ring_on()) {
        const size_t piece = pipe.ring_piece();
        for (uint64_t off = 0; off < total; off += piece) {
            const size_t n = (size_t)(total - off < piece ? total - off : piece);
            int idx = -1;
            char *b = pipe.ring_acquire(&idx);
            if (!b) return -1;
            if (sta_fetch_output_at(eng, b, off, n) != STA_OK) { pipe.ring_release(idx); return -2; }
            pipe.ring_push(&j, idx, n);
        }
        j.ringed = true;
    } else {
        if (j.text.size() < (size_t)total) j.text.resize((size_t)total + (size_t)(total >> 3));
        if (sta_fetch_output(eng, j.text.data(), total) != STA_OK) return -2;
    }
    j.out_bytes = total;
    return 0;
}

// Can the -d cap (bam_plp_push: a read is dropped when more than max_depth reads are live at its start) possibly trigger for these
// staged reads?  Conservative host-side bound: at a read's start at most the reads starting within the longest reference span
// before it are live.  False for ordinary depths, so that the producer need not wait for the device's verdict.
inline bool cap_may_trigger(const std::vector<StagedFile> &staged, int64_t max_depth, const WindowSource &src)
{
    if (max_depth <= 0 || max_depth >= INT32_MAX) return false;
    for (size_t fi = 0; fi < staged.size(); ++fi) {
        const StagedFile &f = staged[fi];
        const int64_t n = f.n();
        if (n <= max_depth) continue;
        // (the spans come from the source: the staging arrays of a BAM lane with device-side pools hold no CIGARs -- reading them
        // here made this bound meaningless and the safety net of the device thread ended such runs, found by scripts/hunt4.py)
        int64_t span_max = src.staged_max_span(fi);
        if (span_max < 1) span_max = 1;
        int64_t lo = 0;
        for (int64_t i = 0; i < n; ++i) {
            // the device counts read j as live at read i's start when pos[j] + span >= pos[i] (k_maxcnt_detect): keep every read
            // with pos[j] >= pos[i] - span_max, boundary included, so that this stays an upper bound of the device's count
            while (f.pos[(size_t)lo] < f.pos[(size_t)i] - span_max) ++lo;
            if (i - lo + 1 > max_depth) return true;
        }
    }
    return false;
}

}  // namespace sta
