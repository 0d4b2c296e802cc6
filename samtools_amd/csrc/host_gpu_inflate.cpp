// host_gpu_inflate.cpp -- see host_gpu_inflate.h
#include "host_gpu_inflate.h"
#include "sta_dev.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

namespace sta {

namespace {

class Inflater : public GpuInflater {
    struct Slot {
        uint8_t *comp_h = nullptr, *comp_d = nullptr, *out_d = nullptr;
        StaBgzfBlock *blk_h = nullptr, *blk_d = nullptr;
        uint32_t *st_h = nullptr, *st_d = nullptr;
        size_t comp_cap = 0, out_cap = 0, blk_cap = 0, st_cap = 0;
        hipEvent_t done = nullptr;
        size_t n = 0;
        bool busy = false;
    };
    int dev_;
    hipStream_t s_ = nullptr;
    Slot slot_[2];
    bool ok_ = false;

    template <class T> static bool grow_pair(T *&h, T *&d, size_t &cap, size_t want)
    {
        if (want <= cap) return true;
        if (h) hipHostFree(h);
        if (d) hipFree(d);
        h = nullptr; d = nullptr; cap = 0;
        const size_t c = want + (want >> 2) + 4096;
        if (hipHostMalloc((void **)&h, c * sizeof(T), hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&d, c * sizeof(T)) != hipSuccess) return false;
        cap = c;
        return true;
    }

public:
    explicit Inflater(int dev) : dev_(dev)
    {
        if (hipSetDevice(dev_) != hipSuccess) return;
        if (hipStreamCreateWithFlags(&s_, hipStreamNonBlocking) != hipSuccess) return;
        for (Slot &sl : slot_) if (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) return;
        ok_ = true;
    }
    ~Inflater() override
    {
        hipSetDevice(dev_);
        if (s_) hipStreamSynchronize(s_);
        for (Slot &sl : slot_) {
            if (sl.comp_h) hipHostFree(sl.comp_h);
            if (sl.blk_h) hipHostFree(sl.blk_h);
            if (sl.st_h) hipHostFree(sl.st_h);
            hipFree(sl.comp_d); hipFree(sl.out_d); hipFree(sl.blk_d); hipFree(sl.st_d);
            if (sl.done) hipEventDestroy(sl.done);
        }
        if (s_) hipStreamDestroy(s_);
    }
    bool ok() const { return ok_; }

    int submit(const GpuInflateJob *jobs, size_t n) override
    {
        if (!ok_ || n == 0) return -1;
        int t = !slot_[0].busy ? 0 : !slot_[1].busy ? 1 : -1;
        if (t < 0) return -1;
        if (hipSetDevice(dev_) != hipSuccess) return -1;
        Slot &sl = slot_[t];
        size_t cbytes = 0, obytes = 0;
        for (size_t i = 0; i < n; ++i) { cbytes += ((size_t)jobs[i].clen + 15) & ~(size_t)15; obytes += jobs[i].isize; }
        // (1 KiB behind the last block's data: the decoder's input window runs ahead of its bit reader)
        if (!grow_pair(sl.comp_h, sl.comp_d, sl.comp_cap, cbytes + 1024) || !grow_pair(sl.blk_h, sl.blk_d, sl.blk_cap, n) || !grow_pair(sl.st_h, sl.st_d, sl.st_cap, n)) { ok_ = false; return -1; }
        if (obytes + 64 > sl.out_cap) {
            hipFree(sl.out_d); sl.out_d = nullptr; sl.out_cap = 0;
            const size_t c = obytes + (obytes >> 2) + 4096;
            if (hipMalloc((void **)&sl.out_d, c) != hipSuccess) { ok_ = false; return -1; }
            sl.out_cap = c;
        }
        size_t co = 0, oo = 0;
        for (size_t i = 0; i < n; ++i) {
            memcpy(sl.comp_h + co, jobs[i].comp, jobs[i].clen);
            sl.blk_h[i] = StaBgzfBlock{ (uint64_t)co, jobs[i].clen, jobs[i].isize, (uint64_t)oo };
            co += ((size_t)jobs[i].clen + 15) & ~(size_t)15; oo += jobs[i].isize;
        }
        memset(sl.comp_h + co, 0, 1024);
        bool good = hipMemcpyAsync(sl.comp_d, sl.comp_h, co + 1024, hipMemcpyHostToDevice, s_) == hipSuccess
                 && hipMemcpyAsync(sl.blk_d, sl.blk_h, n * sizeof(StaBgzfBlock), hipMemcpyHostToDevice, s_) == hipSuccess;
        if (good) {
            sta_launch_bgzf_inflate(s_, sl.comp_d, sl.blk_d, (int)n, sl.out_d, sl.st_d);
            good = hipGetLastError() == hipSuccess && hipMemcpyAsync(sl.st_h, sl.st_d, n * 4, hipMemcpyDeviceToHost, s_) == hipSuccess;
        }
        // download: runs of jobs whose destinations follow each other (a reader's group) go as one copy
        for (size_t i = 0; good && i < n;) {
            size_t j = i + 1; size_t bytes = jobs[i].isize;
            while (j < n && jobs[j].dst == jobs[j - 1].dst + jobs[j - 1].isize) { bytes += jobs[j].isize; ++j; }
            if (bytes) good = hipMemcpyAsync(jobs[i].dst, sl.out_d + sl.blk_h[i].out_off, bytes, hipMemcpyDeviceToHost, s_) == hipSuccess;
            i = j;
        }
        good = good && hipEventRecord(sl.done, s_) == hipSuccess;
        if (!good) { (void)hipGetLastError(); hipStreamSynchronize(s_); ok_ = false; return -1; }
        sl.n = n; sl.busy = true;
        return t;
    }

    bool wait(int ticket, std::vector<uint32_t> &status) override
    {
        if (ticket < 0 || ticket > 1 || !slot_[ticket].busy) return false;
        Slot &sl = slot_[ticket];
        hipSetDevice(dev_);
        const bool good = hipEventSynchronize(sl.done) == hipSuccess;
        sl.busy = false;
        if (!good) { (void)hipGetLastError(); ok_ = false; return false; }
        status.assign(sl.st_h, sl.st_h + sl.n);
        return true;
    }
};

}  // namespace

std::unique_ptr<GpuInflater> make_gpu_inflater(int device)
{
    // Opt-in (STA_GPU_INFLATE=1).  Measured on the 1-Gbase file (profiles/r04_bgzf_inflate_device.md): the kernel inflates 19 GB/s and
    // takes a third off the decode wait of a run on the 16-CPU container, but the run as a whole is still slower (page-locked buffers of
    // two batches to allocate, the inflated bytes crossing PCIe twice more, the pileup kernels sharing the device).
    const char *e = getenv("STA_GPU_INFLATE");
    if (!e || atoi(e) == 0) return nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return nullptr; }
    std::unique_ptr<Inflater> p(new Inflater(device));
    if (!p->ok()) return nullptr;
    return std::unique_ptr<GpuInflater>(p.release());
}

}  // namespace sta
