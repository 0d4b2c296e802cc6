// host_gpu_inflate.h -- batches of BGZF blocks sent through the device's decoder (kernels_inflate.hip) on behalf of the chunked BAM
// reader (host_chunk.cpp): compressed bytes up, inflated bytes back into the reader's page-locked group buffers.  Two batches may be
// in flight, so the upload of one overlaps the kernel and the download of the other.  Host-only builds (tests/cpu harnesses) link
// tests/cpu/gpu_inflate_stub.cpp, whose factory returns nullptr: the reader then inflates on its own threads.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace sta {

struct GpuInflateJob { const uint8_t *comp; uint32_t clen, isize; uint8_t *dst; };     // dst: host memory (page-locked for full speed)

class GpuInflater {
public:
    virtual ~GpuInflater() {}
    // queues upload + kernel + download of the jobs; returns a ticket (>= 0), or -1 on a device error / when two batches are already
    // in flight.  The jobs' compressed bytes are copied before submit() returns; dst must stay valid until wait().
    virtual int submit(const GpuInflateJob *jobs, size_t n) = 0;
    // blocks until the ticket's bytes are in place; status[i] = 0 or why the device gave up on job i (the caller inflates those itself).
    // false: device error, nothing of the batch can be trusted
    virtual bool wait(int ticket, std::vector<uint32_t> &status) = 0;
};

// nullptr: no usable device (or STA_GPU_INFLATE=0)
std::unique_ptr<GpuInflater> make_gpu_inflater(int device);

}  // namespace sta
