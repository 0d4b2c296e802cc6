// kernels_maxcnt.hip -- the -d / bam_mplp_set_maxcnt depth cap (gfx950).
//
// HTSlib bam_plp_push drops an arriving read when it starts on the iterator's current column and
// the live-node count already exceeds maxcnt (pinned by test/mpileup/expected/47.out).
// In window terms: the first read of every start position is always kept; a later read of the same
// start is dropped iff (#kept reads with start <= p and end >= p) + 1 > maxcnt.  That is an
// order-dependent recurrence, but it can only trigger where more than maxcnt-1 reads are stacked,
// so: k_maxcnt_detect bounds the stack height of every read with one binary search on `maxend`
// (parallel, cheap) and raises a flag; only flagged windows run k_maxcnt_serial, a single-lane
// exact replay (still on the device -- there is no host fallback).  Across windows the host keeps the
// iterator's state consistent: reads the cap dropped are not carried into the next window, and carried
// reads arrive flagged STA_AUX_ACCEPTED (they count as live but are never re-tested).
#include "dev_util.h"

__global__ void __launch_bounds__(256) k_maxcnt_detect(StaReadsDev R, int maxcnt, StaCounters *ctr)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < R.n && (R.info[i] & RI_KEEP) && i + 1 >= maxcnt) {
        int32_t key = R.pos[i] - 1;       // reads with end >= pos[i]  <=>  maxend > pos[i]-1
        int64_t lo = 0, hi = i;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (R.maxend[mid] > key) hi = mid; else lo = mid + 1; }
        hit = (i - lo + 1) >= (int64_t)maxcnt;
    }
    if (__ballot(hit) && (threadIdx.x & 63) == 0) atomicAdd(&ctr->maxcnt_flag, 1ull);
}

// exact replay; hist[] counts kept reads by (end - col_lo), zero-initialised, length span+2
__global__ void k_maxcnt_serial(StaReadsDev R, int maxcnt, int32_t col_lo, int32_t span, int32_t *hist, StaCounters *ctr)
{
    if (threadIdx.x || blockIdx.x) return;
    long long live = 0;
    int32_t cur_p = INT32_MIN, retired = col_lo - 1;   // ends <= retired have left the buffer
    unsigned long long dropped = 0;
    for (int64_t i = 0; i < R.n; ++i) {
        uint32_t info = R.info[i];
        if (!(info & RI_PUSHED)) continue;
        int32_t p = R.pos[i], e = R.end[i];
        bool first = p != cur_p;
        if (first) {
            // reads with end <= p-1 were removed while the iterator advanced to p
            for (int32_t c = retired + 1; c <= p - 1; ++c) { if (c - col_lo >= 0 && c - col_lo <= span) live -= hist[c - col_lo]; }
            if (p - 1 > retired) retired = p - 1;
            cur_p = p;
        } else if (!(R.aux[i] & STA_AUX_ACCEPTED) && live + 1 > (long long)maxcnt) {
            if (info & RI_KEEP) { R.info[i] = info & ~(RI_KEEP | RI_OLAP_EL); dropped++; }
            else R.info[i] = info & ~RI_PUSHED;      // zero-span read that would have been dropped: no effect
            continue;
        }
        if (info & RI_KEEP) { live++; hist[e - col_lo]++; }
    }
    if (dropped) atomicAdd(&ctr->n_dropped, dropped);
}

void sta_launch_maxcnt_detect(hipStream_t s, const StaReadsDev &r, int maxcnt, StaCounters *ctr)
{
    if (r.n == 0 || r.n < maxcnt) return;
    hipLaunchKernelGGL(k_maxcnt_detect, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, maxcnt, ctr);
}

void sta_launch_maxcnt(hipStream_t s, const StaReadsDev &r, int maxcnt, int32_t col_lo, int32_t span,
                       int32_t *scratch, StaCounters *ctr)
{
    if (r.n == 0) return;
    hipMemsetAsync(scratch, 0, ((size_t)span + 2) * sizeof(int32_t), s);
    hipLaunchKernelGGL(k_maxcnt_serial, dim3(1), dim3(64), 0, s, r, maxcnt, col_lo, span, scratch, ctr);
}
