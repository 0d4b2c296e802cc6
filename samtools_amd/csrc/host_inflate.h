// host_inflate.h -- raw DEFLATE (RFC 1951) decoding of one BGZF block and its CRC-32 (RFC 1952), for the drivers' decode threads.
// Stands where zlib's inflate() + crc32() stood in host_bgzf.cpp (HTSlib: bgzf.c inflate_block / bgzf_read_block): a BGZF block is a
// self-contained stream of at most 64 KiB either side, so the decoder is written for exactly that -- whole input and output in memory,
// no streaming state, two-level look-up tables built per block, a 64-bit bit buffer refilled once per length / distance pair, word-wise
// match copies -- and runs 1.5-1.6x as fast as zlib 1.2.11 on BAM data (symbol-dense streams: 300 -> 500 MB/s per thread on the build host);
// the CRC is slicing-by-8 (1.0 -> 1.8 GB/s); together a block takes 60 % of the time.  Little-endian hosts (table entries are read as words).  It is an accelerator, not an authority:
// any error it reports, a size or CRC mismatch sends the block through zlib again (host_bgzf.cpp), whose verdict is the one the
// reader acts on.
#pragma once
#include <cstddef>
#include <cstdint>

namespace sta {

// in[0, in_len): the deflate data; the buffer must be readable up to in + in_len + 8 (bytes beyond in_len are never used for output).
// out[0, out_cap): nothing beyond out + out_cap is written (the wide copies of the fast loop stop 274 bytes before it, the last bytes
// are copied exactly), so a block may be inflated in front of live data when out_cap is its known size.
// Returns 0 and *out_len on success (the final block was seen and everything fitted), non-zero on any malformed or truncated input.
int fast_inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len);

uint32_t fast_crc32(const uint8_t *p, size_t n);          // CRC-32 of RFC 1952 (reflected 0xEDB88320, initial value and final xor ~0)

}  // namespace sta
