// dev_lookback.h -- decoupled look-back over column tiles, shared by the single-pass text kernels (k_mplp_fused, k_depth_fused).
//
// A launch hands tiles out by ticket (start order), so a tile only waits for tiles that are already running.  Every tile owns two
// 8-byte status words, {flag, bytes} and {flag, rows << 31 | data columns}; each is written with ONE device-scope store, so the
// value IS the flag and no fence pair is needed (MI355X_MICROARCH.md, "granule").  flag 1 = this tile's aggregate, 2 = inclusive
// prefix over all tiles up to and including this one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ST_FLAG(v) ((unsigned)((v) >> 62))
#define ST_VAL(v) ((v) & 0x3fffffffffffffffull)
#define ST_AGG 1ull
#define ST_PREFIX 2ull

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x)
{
    for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// Called by ONE whole wave of the tile's workgroup: publishes the tile's aggregates, sums its predecessors' (64 at a time) until
// one of them holds a prefix, publishes the inclusive prefix.  ex0 / ex1 = exclusive prefixes of the two words.
__device__ __forceinline__ void tile_lookback(unsigned long long *status, unsigned tile, unsigned long long agg0, unsigned long long agg1,
                                              unsigned long long &ex0, unsigned long long &ex1)
{
    const int lane = threadIdx.x & 63;
    ex0 = 0; ex1 = 0;
    if (tile > 0) {
        if (lane == 0) {
            __hip_atomic_store(&status[2 * (size_t)tile], (ST_AGG << 62) | agg0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&status[2 * (size_t)tile + 1], (ST_AGG << 62) | agg1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        long long look = (long long)tile - 1;
        for (;;) {
            const long long idx = look - lane;
            unsigned long long v0 = ST_PREFIX << 62, v1 = ST_PREFIX << 62;          // before tile 0: an empty prefix
            if (idx >= 0) {
                v0 = __hip_atomic_load(&status[2 * (size_t)idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v1 = __hip_atomic_load(&status[2 * (size_t)idx + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned f0 = ST_FLAG(v0), f1 = ST_FLAG(v1);
            const bool ready = f0 != 0 && f0 == f1;              // both words from the same publication
            const bool is_prefix = ready && f0 == (unsigned)ST_PREFIX;
            const unsigned long long m_prefix = __ballot(is_prefix), m_wait = __ballot(!ready);
            const int first = m_prefix ? __ffsll((long long)m_prefix) - 1 : 64;      // nearest predecessor holding a prefix
            const unsigned long long need = first < 63 ? ((2ull << first) - 1ull) : ~0ull;
            if (m_wait & need) { __builtin_amdgcn_s_sleep(2); continue; }
            const bool take = lane <= first;
            ex0 += wave_sum_u64(take ? ST_VAL(v0) : 0ull);
            ex1 += wave_sum_u64(take ? ST_VAL(v1) : 0ull);
            if (first < 64) break;
            look -= 64;
        }
    }
    if (lane == 0) {
        __hip_atomic_store(&status[2 * (size_t)tile], (ST_PREFIX << 62) | (ex0 + agg0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&status[2 * (size_t)tile + 1], (ST_PREFIX << 62) | (ex1 + agg1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (Round 5 measured two other forms on k_depth_fused and kept neither -- profiles/r05_depth_phases.md: a workgroup-wide version asking for
// 256 .. 2 048 predecessors per round trip instead of 64, 0.244 .. 0.31 ms against 0.235; and ONE status word per tile with the window totals
// added to the counters by two fire-and-forget atomics per tile, 0.262 against 0.219 -- 2 048 atomics on one address are 23 us at the L2.)

// the wave's finished text: LDS [lds, lds + n) -> dst, where the LDS offset is congruent to the global address mod 16 (16-byte
// body stores, byte stores for the ragged ends)
__device__ __forceinline__ void wave_flush_text(const char *lds, char *dst, uint32_t n)
{
    const int lane = threadIdx.x & 63;
    const uint32_t mis = (uint32_t)((uintptr_t)dst & 15);
    uint32_t head = mis ? 16 - mis : 0; if (head > n) head = n;
    if ((uint32_t)lane < head) dst[lane] = lds[lane];
    const uint32_t body = (n - head) >> 4;
    const uint4 *src4 = reinterpret_cast<const uint4 *>(lds + head);
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t i = lane; i < body; i += 64) dst4[i] = src4[i];
    const uint32_t done = head + (body << 4);
    if (done + lane < n) dst[done + lane] = lds[done + lane];
}
