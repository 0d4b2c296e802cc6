// kernels_cov.hip -- per-column reductions for `coverage` and `bedcov`.
//
// Replaces the column loops of coverage.c:621-672 and bedcov.c:316-333, which ask the pileup iterator for every
// column and count entries on the host: here one wave owns 64 columns (one lane per column, reads walked uniformly
// as in the pileup kernels), every lane classifies its column's entries per input file
//   n_plp (all entries) / deletions + reference skips / bases under the quality threshold / sum + count of the qualities kept
// and the per-window totals are block-reduced into a handful of 64-bit counters.  No text, no per-column output.
#include "dev_util.h"

struct CovPar { int32_t mode, min_baseQ, min_depth, skip_dn, hist_bins, hist_depth; int64_t hist_beg, hist_bin_width; };

// coverage -m / -D (coverage.c:632-668): `add` goes to bin (pos - hist_beg) / hist_bin_width.  Columns of a wave are consecutive, so
// the bins are non-decreasing over the lanes: one lane per run of equal bins adds the run's sum (a 64-lane wave usually lies in one bin)
__device__ __forceinline__ void cov_hist_add(const CovPar &P, uint32_t *hist, int64_t pos, bool on, uint32_t add)
{
    const int lane = threadIdx.x & 63;
    int64_t bin = -1;
    if (on && pos >= P.hist_beg) { bin = (pos - P.hist_beg) / P.hist_bin_width; if (bin >= P.hist_bins) bin = -1; }
    if (bin < 0) add = 0;
    const int b32 = (int)bin;
    // inclusive prefix sum of `add` over the wave, then the sum of a run = prefix at its last lane - prefix before its first
    uint32_t pre = add;
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl_up((int)pre, d); if (lane >= d) pre += o; }
    const int prev_bin = __shfl_up(b32, 1), next_bin = __shfl_down(b32, 1);
    const bool head = lane == 0 || prev_bin != b32, tail = lane == 63 || next_bin != b32;
    const unsigned long long heads = __ballot(head);
    const int first = 63 - __clzll((long long)(heads & (~0ull >> (63 - lane))));          // first lane of this lane's run
    const uint32_t before_all = (uint32_t)__shfl((int)pre, first > 0 ? first - 1 : 0);   // (every lane takes part in the shuffle)
    if (tail && b32 >= 0) {
        const uint32_t sum = pre - (first > 0 ? before_all : 0u);
        if (sum) atomicAdd(&hist[b32], sum);
    }
}

// totals[0..4] = n_covered_bases, summed_coverage, summed_baseQ, quality_bases, missing_qual (coverage)
// per_file[f*2 + 0] = cnt, [f*2 + 1] = pcov (bedcov)
__global__ void __launch_bounds__(256) k_cov_cols(StaWinDev W, CovPar P, unsigned long long *totals, unsigned long long *per_file, uint32_t *hist)
{
    const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)wave * 64;
    const bool wave_on = c0 < ncols;
    const int p0 = W.col_beg + (int)(wave_on ? c0 : 0);
    const int p = p0 + lane;
    const bool active = wave_on && p < W.col_end && (!W.has_reg || (W.origin + p >= W.reg_beg && W.origin + p < W.reg_end));
    const int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;

    unsigned long long depth = 0, sumq_all = 0, nq_all = 0, noq = 0;
    uint32_t hsum = 0;                     // coverage -D: the per-file depths of this column, summed
    unsigned long long okmask = 0;         // bedcov -d: bit f = this column reaches the depth threshold in file f
    bool count_base = false, visited = false;
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        uint32_t n_plp = 0, n_dn = 0, n_low = 0, nq = 0; unsigned long long sumq = 0;
        if (wave_on && R.n) {
            int64_t rlo = wave_upper_bound(R.maxend, R.n, p0), rhi = wave_upper_bound(R.pos, R.n, plast);
            if (rlo > rhi) rlo = rhi;
            for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
                const int64_t ri = b0 + lane;
                const bool ok = ri < rhi;
                const uint32_t v_info = ok ? R.info[ri] : 0u;
                const int v_pos = ok ? R.pos[ri] : 0, v_end = ok ? R.end[ri] : 0;
                unsigned long long live = __ballot(ok && (v_info & RI_KEEP) && v_end > p0 && v_pos <= plast);
                while (live) {
                    const int j = __ffsll((long long)live) - 1; live &= live - 1;
                    const uint32_t info = (uint32_t)__builtin_amdgcn_readlane((int)v_info, j);
                    const int rpos = __builtin_amdgcn_readlane(v_pos, j), rend = __builtin_amdgcn_readlane(v_end, j);
                    if (!(active && p >= rpos && p < rend)) continue;
                    const int64_t r = b0 + j;
                    int qpos = p - rpos, indel = 0, k = 0; bool is_del = false, is_refskip = false;
                    if (!(info & RI_SIMPLE))
                        plp_resolve(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), rpos, p, qpos, indel, k, is_del, is_refskip);
                    n_plp++;
                    if (is_del || is_refskip) n_dn++;
                    else if (qpos < R.l_qseq[r]) {
                        int q = R.qual[((uint64_t)R.base_off8[r] << 3) + (uint64_t)qpos];
                        if (q < P.min_baseQ) n_low++; else { sumq += (unsigned long long)q; nq++; }
                    } else noq = 1;
                }
            }
        }
        if (P.mode == 0) {
            // coverage.c:639-662
            int dap = (int)n_plp - (int)n_dn - (int)n_low;
            if (dap > 0) { count_base = true; depth += (unsigned long long)dap; }
            hsum += (uint32_t)dap;
            sumq_all += sumq; nq_all += nq;
        } else {
            // bedcov.c:318-330 (deletions / ref skips are subtracted with -j, and also whenever -d is given)
            visited |= n_plp > 0;
            const int m = (P.skip_dn || P.min_depth >= 0) ? (int)n_dn : 0;
            const int pd = (int)n_plp - m;
            if (P.min_depth >= 0 && pd >= P.min_depth && f < 64) okmask |= 1ull << f;
            unsigned long long v[1] = { active ? (unsigned long long)pd : 0ull };
            unsigned long long *const dst[1] = { &per_file[f * 2] };
            block_reduce_atomic<1, 1>(v, dst);
        }
    }
    if (P.mode == 0) {
        const bool take = active && count_base && depth >= (unsigned long long)(P.min_depth > 0 ? P.min_depth : 1);
        unsigned long long v[5] = { take ? 1ull : 0ull, take ? depth : 0ull, take ? sumq_all : 0ull, take ? nq_all : 0ull, active ? noq : 0ull };
        unsigned long long *const dst[5] = { &totals[0], &totals[1], &totals[2], &totals[3], &totals[4] };
        block_reduce_atomic<5, 5>(v, dst);
        if (P.hist_bins > 0) cov_hist_add(P, hist, W.origin + p, active, P.hist_depth ? hsum : (take ? 1u : 0u));
    } else if (P.min_depth >= 0) {
        // the iterator only visits columns where some file has an entry: only those can count towards the -d column
        for (int f = 0; f < W.nfiles && f < 64; ++f) {
            unsigned long long v[1] = { (active && visited && ((okmask >> f) & 1ull)) ? 1ull : 0ull };
            unsigned long long *const dst[1] = { &per_file[f * 2 + 1] };
            block_reduce_atomic<1, 1>(v, dst);
        }
    }
}

void sta_launch_cov_cols(hipStream_t s, const StaWinDev &w, int mode, int min_baseQ, int min_depth, int skip_dn,
                         unsigned long long *totals, unsigned long long *per_file,
                         uint32_t *hist, int hist_bins, int hist_depth, int64_t hist_beg, int64_t hist_bin_width)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    CovPar p{ mode, min_baseQ, min_depth, skip_dn, hist && hist_bin_width > 0 ? hist_bins : 0, hist_depth, hist_beg, hist_bin_width };
    hipLaunchKernelGGL(k_cov_cols, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, w, p, totals, per_file, hist);
}
