// tests/cpu/gpu_inflate_stub.cpp -- TEST INFRASTRUCTURE: host-only harnesses have no device decoder; the chunked reader then inflates
// on its own threads (samtools_amd/csrc/host_gpu_inflate.h).  STA_FAKE_GPU_INFLATE=1 gives the reader a stand-in that inflates every
// batch with the host decoder at wait() time -- the reader's feeder / parser threading is then exercised (and sanitised) without a GPU;
// =2 also reports every 7th job as "given up by the device" so that the redo path runs.
#include "../../samtools_amd/csrc/host_gpu_inflate.h"
#include "../../samtools_amd/csrc/host_bgzf.h"
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
namespace sta {
namespace {
class Fake : public GpuInflater {
    std::vector<GpuInflateJob> jobs_[2]; bool busy_[2] = { false, false }; int mode_;
public:
    explicit Fake(int m) : mode_(m) {}
    int submit(const GpuInflateJob *jobs, size_t n) override
    {
        const int t = !busy_[0] ? 0 : !busy_[1] ? 1 : -1;
        if (t < 0) return -1;
        jobs_[t].assign(jobs, jobs + n); busy_[t] = true;
        return t;
    }
    bool wait(int t, std::vector<uint32_t> &status) override
    {
        if (t < 0 || t > 1 || !busy_[t]) return false;
        status.assign(jobs_[t].size(), 0);
        for (size_t i = 0; i < jobs_[t].size(); ++i) {
            const GpuInflateJob &j = jobs_[t][i];
            BgzfMap::Block b; b.comp = j.comp; b.clen = j.clen; b.isize = j.isize; memcpy(&b.crc, j.comp + j.clen, 4);
            if (mode_ == 2 && i % 7 == 3) { memset(j.dst, 0xAB, j.isize); status[i] = 2; }
            else if (!bgzf_inflate_block(b, j.dst)) status[i] = 1;
        }
        busy_[t] = false;
        return true;
    }
};
}  // namespace
std::unique_ptr<GpuInflater> make_gpu_inflater(int)
{
    const char *e = getenv("STA_FAKE_GPU_INFLATE");
    if (!e || atoi(e) == 0) return nullptr;
    // (the real decoder takes a while to come up -- the HIP runtime -- and the reader starts without it: STA_FAKE_GPU_INFLATE_DELAY_MS)
    if (const char *d = getenv("STA_FAKE_GPU_INFLATE_DELAY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(d)));
    if (atoi(e) == 3) return nullptr;            // "no usable device": the reader stays with its own inflate
    return std::unique_ptr<GpuInflater>(new Fake(atoi(e)));
}
}  // namespace sta
