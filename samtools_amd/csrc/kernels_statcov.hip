// kernels_statcov.hip -- the coverage distribution of `samtools stats` (the COV section; SURVEY.md 8(f) row 3, second half).
//
// Replaces the pileup round buffer of stats.c:311-391 (round_buffer_insert_read adds 1 to every slot of an aligned block,
// round_buffer_flush visits every slot behind the next read and bins its depth with coverage_idx).  Here an aligned block is two
// marks (+1 at its first position, -1 behind its last), the host hands over the marks of an epoch (one contig) sorted by position --
// with the ring's aliasing already applied to them (driver_stats.cpp) -- and the device turns them into the histogram without ever
// materialising a per-position array:
//   depth after mark i      = carry_in + inclusive prefix sum of the deltas up to i
//   positions at that depth = pos[i + 1] - pos[i]                (a run; zero between marks at one position)
//   cov[coverage_idx(depth)] += run                              for every non-zero depth
// Work is proportional to the number of aligned blocks, not to the number of reference positions.  Three launches per batch: block
// sums, a one-block scan of them, then the rescan that bins (LDS histogram per block when the bins fit, 64-bit counters).
#include "dev_util.h"

#define SC_THREADS 256
#define SC_ITEMS 8
#define SC_TILE (SC_THREADS * SC_ITEMS)
#define SC_LDS_BINS 2048

__global__ void __launch_bounds__(SC_THREADS) k_statcov_sums(const int32_t *delta, int64_t n, long long *sums)
{
    const int64_t base = (int64_t)blockIdx.x * SC_TILE;
    long long x = 0;
    for (int k = 0; k < SC_ITEMS; ++k) { const int64_t i = base + (int64_t)k * SC_THREADS + threadIdx.x; if (i < n) x += delta[i]; }
    unsigned long long v[1] = { (unsigned long long)x };
    __shared__ unsigned long long red[SC_THREADS / 64];
    for (int o = 32; o; o >>= 1) v[0] += __shfl_down(v[0], o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v[0];
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < SC_THREADS / 64; ++w) t += red[w]; sums[blockIdx.x] = (long long)t; }
}

// exclusive scan of the block sums in place (one workgroup; a batch of 4 M marks has 2 048 of them)
__global__ void __launch_bounds__(SC_THREADS) k_statcov_scan(long long *sums, int64_t nb, long long carry_in)
{
    __shared__ long long part[SC_THREADS];
    const int64_t per = (nb + SC_THREADS - 1) / SC_THREADS;
    const int64_t a = (int64_t)threadIdx.x * per, b = a + per < nb ? a + per : nb;
    long long t = 0;
    for (int64_t i = a; i < b; ++i) t += sums[i];
    part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) { long long run = carry_in; for (int k = 0; k < SC_THREADS; ++k) { long long x = part[k]; part[k] = run; run += x; } }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int64_t i = a; i < b; ++i) { long long x = sums[i]; sums[i] = run; run += x; }
}

struct StatCovPar { int32_t cov_min, cov_max, cov_step, ncov; };

// stats.c:311-320
__device__ __forceinline__ int statcov_idx(const StatCovPar &P, long long depth)
{
    if (depth < P.cov_min) return 0;
    if (depth > P.cov_max) return P.ncov - 1;
    return 1 + (int)((depth - P.cov_min) / P.cov_step);
}

// n marks; mark n - 1 only closes the last run (its delta belongs to the next batch)
__global__ void __launch_bounds__(SC_THREADS) k_statcov_bins(const int64_t *pos, const int32_t *delta, int64_t n, const long long *sums, StatCovPar P,
                                                            unsigned long long *cov)
{
    __shared__ unsigned long long hist[SC_LDS_BINS];
    __shared__ long long wave_tot[SC_THREADS / 64];
    const bool lds_hist = P.ncov <= SC_LDS_BINS;
    if (lds_hist) for (int k = threadIdx.x; k < P.ncov; k += SC_THREADS) hist[k] = 0;
    // a thread owns SC_ITEMS consecutive marks, so that its prefix is a register loop; the block's marks are contiguous
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    long long mine = 0;
    int32_t d[SC_ITEMS];
    for (int k = 0; k < SC_ITEMS; ++k) { const int64_t i = base + k; d[k] = i < n - 1 ? delta[i] : 0; mine += d[k]; }
    // exclusive prefix of `mine` over the block: wave scan + wave totals
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long inc = mine;
    for (int o = 1; o < 64; o <<= 1) { long long y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    long long run = sums[blockIdx.x] + (inc - mine);
    for (int w = 0; w < wid; ++w) run += wave_tot[w];
    for (int k = 0; k < SC_ITEMS; ++k) {
        const int64_t i = base + k;
        if (i >= n - 1) break;
        run += d[k];
        const long long len = pos[i + 1] - pos[i];
        if (run != 0 && len > 0) {
            const int b = statcov_idx(P, run);
            if (lds_hist) atomicAdd(&hist[b], (unsigned long long)len); else atomicAdd(&cov[b], (unsigned long long)len);
        }
    }
    if (lds_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < P.ncov; k += SC_THREADS) if (hist[k]) atomicAdd(&cov[k], hist[k]);
    }
}

size_t sta_statcov_tmp_bytes(int64_t n) { return (size_t)((n + SC_TILE - 1) / SC_TILE + 1) * sizeof(long long); }

void sta_launch_statcov(hipStream_t s, const int64_t *pos, const int32_t *delta, int64_t n, long long carry_in,
                        int cov_min, int cov_max, int cov_step, int ncov, unsigned long long *cov, void *tmp)
{
    if (n < 2) return;
    const int64_t nb = (n - 1 + SC_TILE - 1) / SC_TILE;
    long long *sums = (long long *)tmp;
    hipLaunchKernelGGL(k_statcov_sums, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, delta, n - 1, sums);
    hipLaunchKernelGGL(k_statcov_scan, dim3(1), dim3(SC_THREADS), 0, s, sums, nb, carry_in);
    StatCovPar p{ cov_min, cov_max, cov_step, ncov };
    hipLaunchKernelGGL(k_statcov_bins, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, pos, delta, n, (const long long *)sums, p, cov);
}
