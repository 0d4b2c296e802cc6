// deep_strip.h -- the per-block alignment arithmetic of k_mplp_emit_deep (kernels_plp.hip): a read that is plain inside a strip
// of 16 columns has its 16 quality bytes and its packed bases (loaded as 16 + 12 bytes starting at the query index `qb` of the
// first column it covers) shifted ONCE so that column k of the strip finds its quality in byte k and its base code in nibble
// k -- the unrolled column loop then extracts with constant offsets.  Plain functions shared by the kernel and by a CPU unit
// test (tests/cpu/deep_strip_test.cpp) that checks them against the straightforward indexing.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define STA_HD __host__ __device__ __forceinline__
#else
#define STA_HD inline
#endif

// q4: bytes [qb, qb + 16) of the read's qualities; d: strip column of the first covered column (0..15).
// out: byte k = quality shown in strip column k (k >= d)
STA_HD void deep_shift_quals(const uint32_t q4[4], int d, uint32_t out[4])
{
    const uint64_t lo = q4[0] | (uint64_t)q4[1] << 32, hi = q4[2] | (uint64_t)q4[3] << 32;
    const int sh = d << 3;
    uint64_t nlo, nhi;
    if (sh == 0) { nlo = lo; nhi = hi; }
    else if (sh < 64) { nlo = lo << sh; nhi = (hi << sh) | (lo >> (64 - sh)); }
    else { nlo = 0; nhi = lo << (sh - 64); }
    out[0] = (uint32_t)nlo; out[1] = (uint32_t)(nlo >> 32); out[2] = (uint32_t)nhi; out[3] = (uint32_t)(nhi >> 32);
}

// s4: the 12 bytes of packed bases from byte (qb >> 1) of the read on (two bases per byte, the earlier one in the high nibble);
// returns 16 nibbles, nibble k = 4-bit code shown in strip column k (k >= d) -- 0 where it equals the column's reference code
// (rbpack: reference code of column k in nibble k; has_ref false: no comparison)
STA_HD uint64_t deep_shift_bases(const uint32_t s4[3], int qb, int d, uint64_t rbpack, bool has_ref)
{
    // nibble-swap every byte: the stream becomes little-endian in nibbles (nibble j = base (qb & ~1) + j)
    const uint32_t w0 = ((s4[0] & 0x0f0f0f0fu) << 4) | ((s4[0] >> 4) & 0x0f0f0f0fu);
    const uint32_t w1 = ((s4[1] & 0x0f0f0f0fu) << 4) | ((s4[1] >> 4) & 0x0f0f0f0fu);
    const uint32_t w2 = ((s4[2] & 0x0f0f0f0fu) << 4) | ((s4[2] >> 4) & 0x0f0f0f0fu);
    const uint64_t n64 = w0 | (uint64_t)w1 << 32;
    const int j0 = qb & 1;                      // the base of query index qb sits in nibble j0; it belongs in nibble d
    uint64_t n;
    if (d == 0 && j0) n = (n64 >> 4) | ((uint64_t)(w2 & 15u) << 60);
    else n = n64 << ((d - j0) << 2);           // 0 .. 60 bits
    if (has_ref) {
        const uint64_t m = n ^ rbpack;
        uint64_t t = m | (m >> 1);
        t |= t >> 2;
        t &= 0x1111111111111111ull;              // 1 in every nibble that differs from the reference
        n &= t * 15ull;
    }
    return n;
}
