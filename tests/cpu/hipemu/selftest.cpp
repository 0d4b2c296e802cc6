// selftest.cpp -- hipemu's own semantics, checked on kernels small enough to work out by hand (tests/test_hipemu_selftest.py builds
// this file with hipemu.cpp, writes the loop table of the executable with mkloops.py and runs it).  TEST INFRASTRUCTURE ONLY.
//
// Every kernel here is a shape the engine's kernels have, and every expected value is what a wave64 machine with reconvergence at
// the immediate post-dominator computes:
//   k_basic      ballots, shuffles, readlane / readfirstlane, mbcnt on a full wave and on a partial last wave; __syncthreads + LDS
//   k_if_in_loop a persistent loop whose top reads a ticket with readfirstlane(lane 0's atomicAdd) and whose body has wave operations
//                only some lanes execute (k_baq7s): the lanes that skip the body wait at the loop top, which has the LOWER address
//   k_trip       a loop with a lane-dependent trip count and a ballot inside, a full-wave shuffle behind it
//   k_nested     an inner loop only some lanes enter, a join behind it, inside an outer loop
//   k_lookback   workgroup b waits for a flag workgroup b - 1 publishes (decoupled look-back)
#include <hip/hip_runtime.h>
#include <vector>

static int g_bad = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_bad; } } while (0)

__global__ void k_basic(unsigned long long *out, int n)
{
    __shared__ int sh[256];
    const int t = threadIdx.x, lane = t & 63, g = blockIdx.x * blockDim.x + t;
    const bool in = g < n;
    const unsigned long long b = __ballot(in && (lane & 1));
    const int up = __shfl_up(lane, 1), down = __shfl_down(lane, 2), x = __shfl_xor(lane, 5), bc = __shfl(lane * 3, 7), seg = __shfl(lane, 1, 16);
    const int rl = __builtin_amdgcn_readlane(lane + 100, 9), rf = __builtin_amdgcn_readfirstlane(lane + 5);
    const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    sh[t] = t;
    __syncthreads();
    const int other = sh[(t + 64) % (int)blockDim.x];
    if (in) {
        unsigned long long *o = out + (size_t)g * 10;
        o[0] = b; o[1] = (unsigned)up; o[2] = (unsigned)down; o[3] = (unsigned)x; o[4] = (unsigned)bc; o[5] = (unsigned)seg; o[6] = (unsigned)rl; o[7] = (unsigned)rf; o[8] = below; o[9] = (unsigned)other;
    }
}

// groups of work handed out by a ticket; in every group only the lanes whose bit is set in `active[g]` do the body
__global__ void k_if_in_loop(const unsigned long long *active, int ngroups, unsigned *next, unsigned *sum_out, unsigned *trace)
{
    const int lane = threadIdx.x;
    for (;;) {
        unsigned gt = 0;
        if (lane == 0) gt = atomicAdd(next, 1u);
        const int g = __builtin_amdgcn_readfirstlane((int)gt);
        if (g >= ngroups) break;
        const bool act = (active[g] >> lane) & 1;
        if (act) {
            // wave operations among the active lanes only
            const unsigned long long m = __ballot(true);
            const int first = __builtin_amdgcn_readfirstlane(lane);
            int v = lane;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o) * 0 + 0;        // (a few more meetings inside the divergent body)
            if (lane == first) { sum_out[g] = (unsigned)__popcll(m); trace[g] = (unsigned)first; }
        }
    }
}

__global__ void k_trip(const int *trips, unsigned long long *masks /*[64][8]*/, int *total)
{
    const int lane = threadIdx.x;
    const int n = trips[lane];
    int acc = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long m = __ballot(true);          // the lanes still in the loop
        if (i < 8) masks[lane * 8 + i] = m;
        acc += __popcll(m);
    }
    int s = acc;
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);       // behind the loop: every lane again
    if (lane == 0) *total = s;
}

__global__ void k_nested(const int *inner, int outer, unsigned long long *joined, int *inner_seen)
{
    const int lane = threadIdx.x;
    for (int it = 0; it < outer; ++it) {
        const int n = inner[it * 64 + lane];
        int seen = 0;
        if (n > 0) for (int j = 0; j < n; ++j) seen += __popcll(__ballot(true));       // only the lanes inside, fewer every round
        const unsigned long long all = __ballot(true);                                  // the join: everybody
        if (lane == 0) joined[it] = all;
        inner_seen[it * 64 + lane] = seen;
    }
}

__global__ void k_lookback(unsigned long long *flags, unsigned *prefix)
{
    const int b = blockIdx.x;
    unsigned mine = (unsigned)b + 1, ex = 0;
    if (threadIdx.x == 0) {
        if (b > 0) {
            unsigned long long v;
            while (((v = __hip_atomic_load(&flags[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != 1) __builtin_amdgcn_s_sleep(2);
            ex = (unsigned)v;
        }
        __hip_atomic_store(&flags[b], (1ull << 32) | (ex + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ex = (unsigned)__shfl((int)ex, 0);
    if (threadIdx.x == 63) prefix[b] = ex;
}

template <class T> static T *dev(size_t n) { T *p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) abort(); (void)hipMemset(p, 0, n * sizeof(T)); return p; }

int main()
{
    {   // k_basic: 2 workgroups of 128, the last 24 threads beyond n
        const int n = 232;
        unsigned long long *out = dev<unsigned long long>((size_t)n * 10);
        hipLaunchKernelGGL(k_basic, dim3(2), dim3(128), 0, 0, out, n);
        for (int g = 0; g < n; ++g) {
            const int t = g % 128, lane = t & 63, wave_first = g - lane;
            unsigned long long b = 0;
            for (int l = 0; l < 64; ++l) if (wave_first + l < n && (l & 1)) b |= 1ull << l;
            const unsigned long long *o = out + (size_t)g * 10;
            CHECK(o[0] == b);
            CHECK(o[1] == (unsigned)(lane == 0 ? 0 : lane - 1));
            CHECK(o[2] == (unsigned)(lane + 2 < 64 ? lane + 2 : lane));
            CHECK(o[3] == (unsigned)(lane ^ 5));
            CHECK(o[4] == 21u);
            CHECK(o[5] == (unsigned)((lane & ~15) + 1));
            CHECK(o[6] == 109u);
            CHECK(o[7] == 5u);
            CHECK(o[8] == (unsigned)__builtin_popcountll(b & ((1ull << lane) - 1)));
            CHECK(o[9] == (unsigned)((t + 64) % 128));
        }
        (void)hipFree(out);
    }
    {   // k_if_in_loop: 40 groups, one wave; a group's body runs with exactly its active lanes, the ticket is always lane 0's
        const int ng = 40;
        std::vector<unsigned long long> act((size_t)ng);
        unsigned long long x = 0x9E3779B97F4A7C15ull;
        for (int g = 0; g < ng; ++g) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; act[(size_t)g] = g % 7 == 0 ? 0ull : g % 5 == 0 ? ~0ull : g % 3 == 0 ? (x & 0xffffffff00000000ull) : x; }
        unsigned long long *d_act = dev<unsigned long long>(ng); unsigned *next = dev<unsigned>(1), *sum = dev<unsigned>(ng), *tr = dev<unsigned>(ng);
        (void)hipMemcpy(d_act, act.data(), sizeof(unsigned long long) * ng, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_if_in_loop, dim3(1), dim3(64), 0, 0, d_act, ng, next, sum, tr);
        CHECK(*next == (unsigned)ng + 1);
        for (int g = 0; g < ng; ++g) {
            CHECK(sum[g] == (unsigned)__builtin_popcountll(act[(size_t)g]));
            if (act[(size_t)g]) CHECK(tr[g] == (unsigned)__builtin_ctzll(act[(size_t)g]));
        }
    }
    {   // k_trip: lane l goes round (l * 7) % 11 times
        int *trips = dev<int>(64), *total = dev<int>(1); unsigned long long *masks = dev<unsigned long long>(64 * 8);
        for (int l = 0; l < 64; ++l) trips[l] = (l * 7) % 11;
        hipLaunchKernelGGL(k_trip, dim3(1), dim3(64), 0, 0, trips, masks, total);
        int want = 0;
        for (int i = 0; i < 11; ++i) {
            unsigned long long m = 0;
            for (int l = 0; l < 64; ++l) if (trips[l] > i) m |= 1ull << l;
            for (int l = 0; l < 64; ++l) if (trips[l] > i) { want += __builtin_popcountll(m); if (i < 8) CHECK(masks[l * 8 + i] == m); }
        }
        CHECK(*total == want);
    }
    {   // k_nested
        const int outer = 6;
        int *inner = dev<int>(outer * 64), *seen = dev<int>(outer * 64); unsigned long long *joined = dev<unsigned long long>(outer);
        for (int it = 0; it < outer; ++it) for (int l = 0; l < 64; ++l) inner[it * 64 + l] = ((l + it) % 4 == 0) ? 0 : (l * 5 + it) % 6;
        hipLaunchKernelGGL(k_nested, dim3(1), dim3(64), 0, 0, inner, outer, joined, seen);
        for (int it = 0; it < outer; ++it) {
            CHECK(joined[it] == ~0ull);
            for (int l = 0; l < 64; ++l) {
                int want = 0;
                for (int j = 0; j < inner[it * 64 + l]; ++j) { int c = 0; for (int k = 0; k < 64; ++k) if (inner[it * 64 + k] > j) ++c; want += c; }
                CHECK(seen[it * 64 + l] == want);
            }
        }
    }
    {   // k_lookback: exclusive prefix of b + 1 over 37 workgroups
        const int nb = 37;
        unsigned long long *flags = dev<unsigned long long>(nb); unsigned *prefix = dev<unsigned>(nb);
        hipLaunchKernelGGL(k_lookback, dim3(nb), dim3(64), 0, 0, flags, prefix);
        unsigned run = 0;
        for (int b = 0; b < nb; ++b) { CHECK(prefix[b] == run); run += (unsigned)b + 1; }
    }
    if (g_bad) { fprintf(stderr, "hipemu selftest: %d checks failed\n", g_bad); return 1; }
    printf("hipemu selftest: ok\n");
    return 0;
}
