// host_bgzf.h -- byte sources for the drivers' SAM/BAM reader.
// Stands where HTSlib's bgzf.c (absent from the reference tree; SAM spec section 4.1 is the format) stands for
// sam_open/sam_read1: a BGZF file is a series of <=64 KiB gzip members whose extra field "BC" carries the
// compressed block size, so blocks can be cut on the host by one I/O thread and inflated by a pool of workers
// (host inflate is the end-to-end limiter either side of the pileup engine).
// Anything that is not BGZF (plain text, ordinary gzip, stdin) goes through zlib's gzread on the caller's thread.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

namespace sta {

class ByteSource {
public:
    virtual ~ByteSource() {}
    // up to n bytes; fewer only at end of data or on error
    virtual size_t read(void *dst, size_t n) = 0;
    virtual bool failed() const = 0;
    // threads <= 0: $STA_IO_THREADS, else 4..8 depending on the machine
    static std::unique_ptr<ByteSource> open(const std::string &path, int threads, std::string *err);
    // a BGZF file from the block that starts at compressed offset `coffset` on (the upper 48 bits of a BAI virtual offset);
    // nullptr when the file is not BGZF or the offset cannot be reached
    static std::unique_ptr<ByteSource> open_bgzf_at(const std::string &path, int threads, uint64_t coffset, std::string *err);
};

// A BGZF file mapped read-only, for readers that cut it into blocks themselves and inflate them on their own threads straight into
// their own buffers (host_chunk.h: the stream interface above hands every inflated byte over through one thread -- a copy of the
// whole uncompressed file on the critical path; here no inflated byte is copied at all).
class BgzfMap {
public:
    struct Block { const uint8_t *comp; uint32_t clen, crc, isize; };     // deflate data (readable 8 bytes beyond: CRC32 + ISIZE follow)
    // nullptr when the path cannot be mapped (stdin, a pipe) or does not start with a BGZF block
    static std::unique_ptr<BgzfMap> open(const std::string &path, std::string *err);
    ~BgzfMap();
    // the block at compressed offset *coffset (advanced behind it): 1 = a block (isize may be 0: the end-of-file marker), 0 = clean end
    // of the file, -1 = not a BGZF block / truncated
    int block_at(uint64_t *coffset, Block *b) const;
private:
    const uint8_t *base_ = nullptr; size_t size_ = 0;
};
// inflates one block into dst[0, isize) -- nothing beyond is written -- and checks size and CRC (host_inflate.h first, zlib for
// whatever that does not deliver: zlib's verdict counts)
bool bgzf_inflate_block(const BgzfMap::Block &b, uint8_t *dst);

int io_default_threads();
int host_cpus_available();      // min(hardware threads, affinity mask, cgroup CPU quota)
void report_thread_budget();     // STA_DRIVER_TIMING: one line on stderr
int host_node_ranks();          // processes of this job sharing the node (STA_NODE_RANKS, STA_SHARD world, LOCAL_WORLD_SIZE)
// per-input worker count when a command reads n_inputs files at once: the default is shared out (at least 1 each), so that
// a hundred-file mpileup does not start a thousand threads
int io_threads_per_input(int n_inputs);

}  // namespace sta
