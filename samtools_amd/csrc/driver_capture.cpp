// driver_capture.cpp -- the command drivers with their text handed back in memory instead of written to a file.
//
// A sharded run (samtools_amd/shard.py) gathers every rank's block of text on rank 0; the drivers' writer thread
// appends to a FILE*, so the capture form gives it a memory stream and returns the buffer: no temporary file between the
// driver and the gather.  The reference has no counterpart (its column loop prints as it goes, bam_plcmd.c:663-868).
#include <unistd.h>
#include <sys/stat.h>
#include <string>
#include <vector>
#include "driver_pipeline.h"
#include <atomic>
#include <unistd.h>
#include "driver_shard.h"
#include <hip/hip_runtime.h>
#include "../../include/samtools_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sta {
static thread_local FILE *t_capture = nullptr;
static thread_local DevCapture *t_dev_capture = nullptr;
DevCapture *driver_dev_capture() { return t_dev_capture; }
char *DevCapture::reserve(size_t more)
{
    if (failed) return nullptr;
    // (64 bytes of slack behind the text, like the engine's own output buffer: the emit kernels' 16-byte flush may touch the tail's line)
    if (len + more + 64 > cap) {
        size_t want = cap ? cap * 2 : (size_t)64 << 20;
        while (want < len + more + 64) want *= 2;
        char *nb = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&nb, want) != hipSuccess) { (void)hipGetLastError(); failed = true; return nullptr; }
        // (the emits so far ran on the engine's stream: everything is waited for before the text moves)
        if (hipDeviceSynchronize() != hipSuccess || (len && hipMemcpy(nb, buf, len, hipMemcpyDeviceToDevice) != hipSuccess)) { (void)hipGetLastError(); hipFree(nb); failed = true; return nullptr; }
        hipFree(buf);
        buf = nb; cap = want;
    }
    return buf + len;
}
FILE *driver_default_out() { return t_capture ? t_capture : stdout; }

// The command-line program is about to end with the driver's status: what is left after the last window has been written -- joining the
// pipeline's threads, freeing the page-locked staging pools (0.11-0.18 s on the GPU box), destroying the engine and unloading the HIP
// runtime (another ~0.13 s: profiles/r06_sessionA_e2e_timeline.log) -- buys a process nothing it does not get from the kernel at exit.
// Set by main.cpp only (never by the in-process entries: sta_main_capture, the ctypes mirror); STA_NO_FAST_EXIT=1 keeps the orderly
// teardown (profilers that flush their traces from exit handlers).
static std::atomic<int> g_exit_after_main{0};
void driver_exit_now_if_asked(int status, FILE *out)
{
    if (!g_exit_after_main.load()) return;
    if (out && out != stdout && out != stderr) { if (fclose(out) != 0) status = status ? status : 1; }
    if (fflush(stdout) != 0) status = status ? status : 1;
    fflush(stderr);
    timeline_mark("fast exit");
    _exit(status);
}
bool driver_out_is_borrowed(FILE *f) { return f == stdout || (t_capture && f == t_capture); }

int dev_threads_from_env()
{
    const char *e = getenv("STA_DEV_THREADS");
    const int n = e ? atoi(e) : 1;      // measured (profiles/r03_e2e_*.log): the drivers are producer-bound, a second engine buys nothing and doubles the BAQ slab start-up
    return n < 1 ? 1 : (n > 4 ? 4 : n);
}
// Pipeline slots: a slot owns a window's staging arrays.  With page-locked pools (large inputs) few of them, each costs page-locking time;
// with plain memory (small inputs, driver_pin_policy) eight, so that the producer runs ahead while the HIP runtime is still coming up.
size_t pipe_slots_from_env(int n_dev)
{
    const char *ns = getenv("STA_PIPE_SLOTS");
    return ns && atoi(ns) > 0 ? (size_t)atoi(ns) : pinned_policy() ? (size_t)n_dev + 2 : (size_t)n_dev + 7;
}
// Page-locked staging pools only for inputs large enough to pay for them (host_pinned.h): >= 1 GiB of input files, or input that is
// not a regular file (a pipe: size unknown).  STA_PIN=0 / 1 overrides.  Call before the inputs are opened (their decode threads allocate).
void driver_pin_policy(const std::vector<std::string> &paths)
{
    uint64_t total = 0; bool unknown = false;
    for (const std::string &fn : paths) {
        struct stat st;
        if (fn == "-" || stat(fn.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) unknown = true; else total += (uint64_t)st.st_size;
    }
    pinned_set_policy(unknown || total >= (1ull << 30));
}
int DevEngines::dev_threads() { return dev_threads_from_env(); }
int DevEngines::create(int device)
{
    for (int d = 0; d < n_; ++d) {
        hipStream_t st = nullptr;
        if (n_ > 1) {
            if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return STA_ERR_HIP; }
        }
        sta_engine *e = nullptr;
        const int rc = sta_engine_create(&e, device, st);
        if (rc != STA_OK) { if (st) hipStreamDestroy(st); return rc; }
        eng.push_back(e); streams.push_back(st);
    }
    // the runtime exists: page-locked allocations are possible from here on (host_pinned.h), and the text ring is made now, on this
    // thread, while the producer stages its first windows
    pinned_runtime_is_up();
    if (n_ == 1) {
        int pieces = 6, mib = 8;
        if (const char *e = getenv("STA_TEXT_RING")) { if (sscanf(e, "%dx%d", &pieces, &mib) != 2) { pieces = atoi(e) > 0 ? 6 : 0; mib = 8; } }
        if (pieces > 0 && mib > 0) {
            ring.piece = (size_t)mib << 20;
            for (int i = 0; i < pieces; ++i) {
                void *b = nullptr;
                if (hipHostMalloc(&b, ring.piece, hipHostMallocDefault) != hipSuccess || !b) { (void)hipGetLastError(); break; }
                ring.buf.push_back((char *)b);
            }
            if (ring.buf.size() < 2) { for (char *b : ring.buf) (void)hipHostFree(b); ring.buf.clear(); }
        }
    }
    return STA_OK;
}
void DevEngines::start(int device)
{
    std::lock_guard<std::mutex> lk(m_);
    if (started_) return;
    started_ = true;
    th_ = std::thread([this, device] { rc_ = create(device); timeline_mark("HIP runtime and engine(s) up"); });
}
int DevEngines::ready()
{
    std::lock_guard<std::mutex> lk(m_);
    if (!started_) return STA_ERR_ARG;
    if (!joined_) { th_.join(); joined_ = true; }
    return rc_;
}
void DevEngines::destroy()
{
    if (started_) (void)ready();
    for (size_t d = 0; d < eng.size(); ++d) {
        sta_engine_destroy(eng[d]);
        if (streams[d]) hipStreamDestroy((hipStream_t)streams[d]);
    }
    eng.clear(); streams.clear();
    for (char *b : ring.buf) (void)hipHostFree(b);
    ring.buf.clear();
}
}  // namespace sta

// (not when something reports from the teardown: the drivers' timing lines and the staging report are printed by destructors, and a
// profiler flushes its traces from exit handlers that _exit() skips)
static bool teardown_has_something_to_say()
{
    if (const char *f = getenv("STA_FAST_EXIT")) if (atoi(f) != 0) return false;      // (measurements: the timeline's marks up to "fast exit" with the timing variables set)
    for (const char *v : { "STA_NO_FAST_EXIT", "STA_DRIVER_TIMING", "STA_STAGE_REPORT", "STA_DEBUG", "STA_PROFILE", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB" })
        if (getenv(v)) return true;
    const char *pre = getenv("LD_PRELOAD");
    return pre && (strstr(pre, "rocprof") || strstr(pre, "asan") || strstr(pre, "tsan"));
}
extern "C" void sta_exit_after_main(int on) { sta::g_exit_after_main.store(on && !teardown_has_something_to_say() ? 1 : 0); }

extern "C" int sta_main_capture(int argc, char **argv, char **text, uint64_t *n_bytes)
{
    if (!text || !n_bytes || argc < 1 || !argv || !argv[0]) return STA_ERR_ARG;
    *text = nullptr; *n_bytes = 0;
    const bool mp = !strcmp(argv[0], "mpileup"), dp = !strcmp(argv[0], "depth");
    if (!mp && !dp) return STA_ERR_ARG;
    char *buf = nullptr; size_t len = 0;
    FILE *ms = open_memstream(&buf, &len);
    if (!ms) return STA_ERR_IO;
    sta::t_capture = ms;
    const int rc = mp ? sta_main_mpileup(argc, argv) : sta_main_depth(argc, argv);
    sta::t_capture = nullptr;
    if (fclose(ms) != 0) { free(buf); return STA_ERR_IO; }
    *text = buf; *n_bytes = (uint64_t)len;
    return rc;
}

extern "C" void sta_capture_free(char *text) { free(text); }

// ---- the same with the windows' text left on the device ----
namespace { sta::DevCapture g_kept; bool g_have_kept = false; }

extern "C" int sta_main_capture_device(int argc, char **argv, uint64_t *n_dev_bytes, char **host_text, uint64_t *n_host_bytes)
{
    if (!n_dev_bytes || !host_text || !n_host_bytes || argc < 1 || !argv || !argv[0]) return STA_ERR_ARG;
    *n_dev_bytes = 0; *host_text = nullptr; *n_host_bytes = 0;
    const bool mp = !strcmp(argv[0], "mpileup"), dp = !strcmp(argv[0], "depth");
    if (!mp && !dp) return STA_ERR_ARG;
    if (sta::dev_threads_from_env() != 1) return STA_ERR_ARG;        // windows must reach the device in output order
    if (g_have_kept) { hipSetDevice(g_kept.device); hipFree(g_kept.buf); g_kept = sta::DevCapture(); g_have_kept = false; }
    char *buf = nullptr; size_t len = 0;
    FILE *ms = open_memstream(&buf, &len);
    if (!ms) return STA_ERR_IO;
    sta::DevCapture cap;
    cap.device = getenv("STA_DEVICE") ? atoi(getenv("STA_DEVICE")) : 0;
    sta::t_capture = ms; sta::t_dev_capture = &cap;
    int rc = mp ? sta_main_mpileup(argc, argv) : sta_main_depth(argc, argv);
    sta::t_capture = nullptr; sta::t_dev_capture = nullptr;
    if (fclose(ms) != 0) { free(buf); hipFree(cap.buf); return STA_ERR_IO; }
    if (cap.failed && rc == 0) rc = STA_ERR_HIP;
    hipSetDevice(cap.device);
    (void)hipDeviceSynchronize();
    // what the driver wrote itself (depth -H's header line) comes first in the output: the windows follow it
    *host_text = buf; *n_host_bytes = (uint64_t)len; *n_dev_bytes = (uint64_t)cap.len;
    g_kept = cap; g_have_kept = true;
    return rc;
}

extern "C" int sta_capture_device_take(void *dev_dst, uint64_t capacity)
{
    if (!g_have_kept) return STA_ERR_ARG;
    int rc = STA_OK;
    hipSetDevice(g_kept.device);
    if (g_kept.len) {
        if (!dev_dst || capacity < g_kept.len) rc = STA_ERR_ARG;
        else if (hipMemcpy(dev_dst, g_kept.buf, g_kept.len, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipGetLastError(); rc = STA_ERR_HIP; }
    }
    hipFree(g_kept.buf);
    g_kept = sta::DevCapture(); g_have_kept = false;
    return rc;
}
