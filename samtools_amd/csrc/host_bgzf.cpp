// host_bgzf.cpp -- see host_bgzf.h
#include "host_bgzf.h"
#include "host_inflate.h"
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace sta {

// CPUs this process may actually use: the smaller of the hardware threads, the scheduler affinity mask and the cgroup CPU quota (cgroup v2
// cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  The GPU boxes of the measurements report 256 hardware threads and run the
// container under `cpu.max = 1600000 100000`: 16 CPUs (profiles/r04_box_cpu_probe.log) -- hardware_concurrency() alone oversubscribes 16x.
int host_cpus_available()
{
    long n = (long)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    auto quota = [&](long q, long period) { if (q > 0 && period > 0) { const long c = (q + period - 1) / period; if (c >= 1 && c < n) n = c; } };
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[32] = { 0 }; long period = 0;
        if (fscanf(f, "%31s %ld", a, &period) == 2 && strcmp(a, "max") != 0) quota(atol(a), period);
        fclose(f);
    } else {
        long q = -1, period = 0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &q) != 1) q = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        quota(q, period);
    }
    return (int)n;
}

// processes of this job that share the node's CPUs: STA_NODE_RANKS if set, else the world of STA_SHARD=rank/world (one process per GPU of one
// node: driver_shard.h, samtools_amd/shard.py), else torchrun's LOCAL_WORLD_SIZE
int host_node_ranks()
{
    if (const char *e = getenv("STA_NODE_RANKS")) { const int v = atoi(e); if (v >= 1) return v; }
    if (const char *e = getenv("STA_SHARD")) { int r = 0, w = 0; if (sscanf(e, "%d/%d", &r, &w) == 2 && w >= 1) return w; }
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) { const int v = atoi(e); if (v >= 1) return v; }
    return 1;
}

int io_default_threads()
{
    if (const char *e = getenv("STA_IO_THREADS")) { int v = atoi(e); if (v > 0) return v > 64 ? 64 : v; }
    // this rank's share of the CPUs it may use, at most 16 decode threads: on the 16-CPU box 16 are the best setting and 32 or more make the
    // whole pipeline slower (profiles/r04_e2e_threads_16cpu_quota.log; r03_e2e_30x_1gbase_threads.log on the unrestricted 256-thread host);
    // eight ranks on such a box get two each instead of 8 x 16
    const int share = host_cpus_available() / host_node_ranks();
    return share > 16 ? 16 : share < 2 ? 2 : share;
}

// STA_DRIVER_TIMING=1: what this rank decided (one line per process, on stderr)
void report_thread_budget()
{
    const char *sh = getenv("STA_SHARD");
    fprintf(stderr, "[driver threads] shard %s: %d decode threads per input lane (CPUs this process may use: %d, ranks sharing the node: %d)\n",
            sh && *sh ? sh : "0/1", io_default_threads(), host_cpus_available(), host_node_ranks());
}

int io_threads_per_input(int n_inputs)
{
    const int t = io_default_threads();
    if (n_inputs <= 2) return t;
    const int share = (2 * t + n_inputs - 1) / n_inputs;
    return share < 1 ? 1 : share;
}

namespace {

// ---- zlib gzread on the caller's thread (plain text, ordinary gzip, stdin) ----
class GzSource : public ByteSource {
public:
    explicit GzSource(gzFile f) : fp_(f) { gzbuffer(fp_, 1 << 18); }
    ~GzSource() override { if (fp_) gzclose(fp_); }
    size_t read(void *dst, size_t n) override
    {
        size_t got = 0;
        while (got < n && !eof_) {
            unsigned want = (unsigned)std::min<size_t>(n - got, 1u << 30);
            int k = gzread(fp_, (char *)dst + got, want);
            if (k < 0) { bad_ = true; eof_ = true; break; }
            if (k == 0) { eof_ = true; break; }
            got += (size_t)k;
        }
        return got;
    }
    bool failed() const override { return bad_; }
private:
    gzFile fp_; bool eof_ = false, bad_ = false;
};

// ---- BGZF: one I/O thread cuts blocks, workers inflate, the consumer takes them in file order ----
class BgzfSource : public ByteSource {
    enum { EMPTY = 0, QUEUED = 1, DONE = 2 };
    struct Slot {
        std::vector<uint8_t> comp, out;
        uint32_t clen = 0, olen = 0, crc = 0, isize = 0;
        int state = EMPTY;
        bool bad = false;
    };
public:
    BgzfSource(FILE *fp, int threads) : fp_(fp)
    {
        if (threads < 1) threads = 1;
        slots_.resize((size_t)threads * 8 + 8);
        // (+16: fast_inflate reads up to 8 bytes behind the deflate data and its wide copies may write a few bytes behind the output)
        for (auto &s : slots_) { s.comp.resize((1 << 16) + 16); s.out.resize((1 << 16) + 16); }
        io_ = std::thread([this] { io_loop(); });
        for (int i = 0; i < threads; ++i) workers_.emplace_back([this] { work_loop(); });
    }
    ~BgzfSource() override
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_free_.notify_all(); cv_work_.notify_all(); cv_done_.notify_all();
        if (io_.joinable()) io_.join();
        for (auto &t : workers_) if (t.joinable()) t.join();
        if (fp_) fclose(fp_);
    }
    size_t read(void *dst, size_t n) override
    {
        size_t got = 0;
        while (got < n) {
            if (cur_ && cur_off_ < cur_->olen) {
                size_t k = std::min<size_t>(cur_->olen - cur_off_, n - got);
                memcpy((char *)dst + got, cur_->out.data() + cur_off_, k);
                cur_off_ += (uint32_t)k; got += k;
                continue;
            }
            if (!advance()) break;
        }
        return got;
    }
    bool failed() const override { return bad_; }
private:
    FILE *fp_;
    std::vector<Slot> slots_;
    std::mutex m_;
    std::condition_variable cv_free_, cv_work_, cv_done_;
    std::deque<uint64_t> work_;
    uint64_t n_cut_ = 0, n_taken_ = 0;     // blocks handed to the pool / returned by the consumer (guarded by m_)
    bool stop_ = false, io_eof_ = false, io_bad_ = false;
    std::thread io_; std::vector<std::thread> workers_;
    Slot *cur_ = nullptr; uint32_t cur_off_ = 0; bool bad_ = false, end_ = false;
    const bool use_fast_ = !(getenv("STA_INFLATE") && !strcmp(getenv("STA_INFLATE"), "zlib"));

    // release the current block and wait for the next one in file order
    bool advance()
    {
        if (end_) return false;
        std::unique_lock<std::mutex> lk(m_);
        if (cur_) { cur_->state = EMPTY; cur_ = nullptr; ++n_taken_; cv_free_.notify_one(); }
        for (;;) {
            if (n_taken_ < n_cut_) {
                Slot &s = slots_[(size_t)(n_taken_ % slots_.size())];
                if (s.state == DONE) {
                    if (s.bad) { bad_ = true; end_ = true; return false; }
                    cur_ = &s; cur_off_ = 0;
                    return true;
                }
            } else if (io_eof_) {
                if (io_bad_) bad_ = true;
                end_ = true;
                return false;
            }
            cv_done_.wait(lk);
        }
    }

    void io_loop()
    {
        for (;;) {
            Slot *s;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_free_.wait(lk, [this] { return stop_ || n_cut_ - n_taken_ < slots_.size(); });
                if (stop_) return;
                s = &slots_[(size_t)(n_cut_ % slots_.size())];
            }
            // SAM spec 4.1: 12 fixed bytes, XLEN, extra subfields (BC carries BSIZE = total block size - 1), deflate data, CRC32, ISIZE
            uint8_t h[12];
            size_t k = fread(h, 1, 12, fp_);
            bool eof = k == 0, bad = false;
            uint32_t bsize = 0, xlen = 0;
            if (!eof) {
                if (k != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) bad = true;
                else {
                    xlen = h[10] | (uint32_t)h[11] << 8;
                    uint8_t x[65536];
                    if (fread(x, 1, xlen, fp_) != xlen) bad = true;
                    else {
                        bool found = false;
                        for (uint32_t o = 0; o + 4 <= xlen;) {
                            uint32_t sl = x[o + 2] | (uint32_t)x[o + 3] << 8;
                            if (x[o] == 'B' && x[o + 1] == 'C' && sl == 2 && o + 6 <= xlen) { bsize = (x[o + 4] | (uint32_t)x[o + 5] << 8) + 1u; found = true; }
                            o += 4 + sl;
                        }
                        if (!found || bsize < 12 + xlen + 8) bad = true;
                    }
                }
                if (!bad) {
                    uint32_t rest = bsize - 12 - xlen;            // deflate data + CRC32 + ISIZE (the 12 fixed bytes include XLEN)
                    if (rest < 8 || rest > (1u << 16) + 8) bad = true;
                    else {
                        s->clen = rest - 8;
                        uint8_t tail[8];
                        if (fread(s->comp.data(), 1, s->clen, fp_) != s->clen || fread(tail, 1, 8, fp_) != 8) bad = true;
                        else { memcpy(&s->crc, tail, 4); memcpy(&s->isize, tail + 4, 4); if (s->isize > (1u << 16)) bad = true; }
                    }
                }
            }
            std::lock_guard<std::mutex> g(m_);
            if (eof || bad) { io_eof_ = true; io_bad_ = bad; cv_done_.notify_all(); return; }
            s->state = QUEUED; s->bad = false;
            work_.push_back(n_cut_++);
            cv_work_.notify_one();
        }
    }

    void work_loop()
    {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return;
        for (;;) {
            Slot *s;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [this] { return stop_ || !work_.empty(); });
                if (stop_) break;
                s = &slots_[(size_t)(work_.front() % slots_.size())];
                work_.pop_front();
            }
            bool bad = false;
            // the block decoder of host_inflate.h first; whatever it does not deliver with the right size and CRC goes through zlib, whose
            // verdict counts (STA_INFLATE=zlib: zlib only)
            size_t fl = 0;
            const bool fast_ok = use_fast_ && fast_inflate(s->comp.data(), s->clen, s->out.data(), 1u << 16, &fl) == 0 && fl == s->isize
                                 && fast_crc32(s->out.data(), fl) == s->crc;
            if (fast_ok) s->olen = (uint32_t)fl;
            else {
                inflateReset(&zs);
                zs.next_in = s->comp.data(); zs.avail_in = s->clen;
                zs.next_out = s->out.data(); zs.avail_out = 1u << 16;
                int rc = inflate(&zs, Z_FINISH);
                s->olen = (uint32_t)((1u << 16) - zs.avail_out);
                if (rc != Z_STREAM_END || s->olen != s->isize) bad = true;
                else if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), s->out.data(), s->olen) != s->crc) bad = true;
            }
            std::lock_guard<std::mutex> g(m_);
            s->bad = bad; s->state = DONE;
            cv_done_.notify_all();
        }
        inflateEnd(&zs);
    }
};

bool looks_like_bgzf(const uint8_t *h, size_t n)
{
    if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
    uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
    for (uint32_t o = 0; o + 4 <= xlen && 12 + o + 4 <= n;) {
        uint32_t sl = h[12 + o + 2] | (uint32_t)h[12 + o + 3] << 8;
        if (h[12 + o] == 'B' && h[12 + o + 1] == 'C' && sl == 2) return true;
        o += 4 + sl;
    }
    return false;
}

}  // namespace

// ---- the mapped file ----
std::unique_ptr<BgzfMap> BgzfMap::open(const std::string &path, std::string *err)
{
    if (path == "-") return nullptr;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { if (err) *err = "failed to open " + path; return nullptr; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) { ::close(fd); return nullptr; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) return nullptr;
    if (!looks_like_bgzf((const uint8_t *)m, (size_t)st.st_size < 64 ? (size_t)st.st_size : 64)) { munmap(m, (size_t)st.st_size); return nullptr; }
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    std::unique_ptr<BgzfMap> r(new BgzfMap());
    r->base_ = (const uint8_t *)m; r->size_ = (size_t)st.st_size;
    return r;
}

BgzfMap::~BgzfMap() { if (base_) munmap(const_cast<uint8_t *>(base_), size_); }

int BgzfMap::block_at(uint64_t *coffset, Block *b) const
{
    const uint64_t o = *coffset;
    if (o >= size_) return o == size_ ? 0 : -1;
    // SAM spec 4.1 (as BgzfSource::io_loop): 12 fixed bytes, XLEN, extra subfields with BC = BSIZE - 1, deflate data, CRC32, ISIZE
    if (size_ - o < 18) return -1;
    const uint8_t *h = base_ + o;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return -1;
    const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
    if (size_ - o < 12 + (uint64_t)xlen + 8) return -1;
    uint32_t bsize = 0; bool found = false;
    const uint8_t *x = h + 12;
    for (uint32_t k = 0; k + 4 <= xlen;) {
        const uint32_t sl = x[k + 2] | (uint32_t)x[k + 3] << 8;
        if (x[k] == 'B' && x[k + 1] == 'C' && sl == 2 && k + 6 <= xlen) { bsize = (x[k + 4] | (uint32_t)x[k + 5] << 8) + 1u; found = true; }
        k += 4 + sl;
    }
    if (!found || bsize < 12 + xlen + 8 || size_ - o < bsize) return -1;
    const uint32_t rest = bsize - 12 - xlen;
    if (rest < 8 || rest > (1u << 16) + 8) return -1;
    b->comp = h + 12 + xlen; b->clen = rest - 8;
    memcpy(&b->crc, b->comp + b->clen, 4); memcpy(&b->isize, b->comp + b->clen + 4, 4);
    if (b->isize > (1u << 16)) return -1;
    *coffset = o + bsize;
    return 1;
}

bool bgzf_inflate_block(const BgzfMap::Block &b, uint8_t *dst)
{
    static const bool use_fast = !(getenv("STA_INFLATE") && !strcmp(getenv("STA_INFLATE"), "zlib"));
    if (b.isize == 0) return true;            // (whatever the deflate data says: an empty block carries nothing)
    size_t fl = 0;
    if (use_fast && fast_inflate(b.comp, b.clen, dst, b.isize, &fl) == 0 && fl == b.isize && fast_crc32(dst, fl) == b.crc) return true;
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(b.comp); zs.avail_in = b.clen;
    zs.next_out = dst; zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 && (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.isize) == b.crc;
    inflateEnd(&zs);
    return ok;
}

std::unique_ptr<ByteSource> ByteSource::open(const std::string &path, int threads, std::string *err)
{
    if (threads <= 0) threads = io_default_threads();
    if (path != "-") {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) { if (err) *err = "failed to open " + path; return nullptr; }
        setvbuf(fp, nullptr, _IOFBF, 1 << 20);      // before any other operation on the stream (ISO C 7.21.5.6)
        uint8_t h[64];
        size_t n = fread(h, 1, sizeof h, fp);
        if (looks_like_bgzf(h, n) && fseek(fp, 0, SEEK_SET) == 0)
            return std::unique_ptr<ByteSource>(new BgzfSource(fp, threads));
        fclose(fp);
    }
    gzFile g = path == "-" ? gzdopen(fileno(stdin), "rb") : gzopen(path.c_str(), "rb");
    if (!g) { if (err) *err = "failed to open " + path; return nullptr; }
    return std::unique_ptr<ByteSource>(new GzSource(g));
}

std::unique_ptr<ByteSource> ByteSource::open_bgzf_at(const std::string &path, int threads, uint64_t coffset, std::string *err)
{
    if (threads <= 0) threads = io_default_threads();
    FILE *fp = path == "-" ? nullptr : fopen(path.c_str(), "rb");
    if (!fp) { if (err) *err = "failed to open " + path; return nullptr; }
    setvbuf(fp, nullptr, _IOFBF, 1 << 20);
    uint8_t h[64];
    const size_t n = fread(h, 1, sizeof h, fp);
    if (!looks_like_bgzf(h, n) || fseeko(fp, (off_t)coffset, SEEK_SET) != 0) { fclose(fp); if (err) *err = "cannot seek in " + path; return nullptr; }
    return std::unique_ptr<ByteSource>(new BgzfSource(fp, threads));
}

}  // namespace sta
