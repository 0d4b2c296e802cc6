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

// first / last read the cap can possibly drop (the detector's bound, per read): range[0] = n - first (so that zero means "none" and a
// maximum finds the smallest index), range[1] = last + 1
__global__ void __launch_bounds__(256) k_maxcnt_range(StaReadsDev R, int maxcnt, unsigned long long *range)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < R.n && (R.info[i] & RI_PUSHED) && i + 1 >= maxcnt) {      // (zero-span reads too: the replay un-marks those the cap would have dropped)
        int32_t key = R.pos[i] - 1;
        int64_t lo = 0, hi = i;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (R.maxend[mid] > key) hi = mid; else lo = mid + 1; }
        hit = (i - lo + 1) >= (int64_t)maxcnt;
    }
    const unsigned long long m = __ballot(hit);
    if (m && (threadIdx.x & 63) == 0) {
        const int64_t base = i;                                   // lane 0's read
        atomicMax(&range[0], (unsigned long long)(R.n - (base + (__ffsll((long long)m) - 1))));
        atomicMax(&range[1], (unsigned long long)(base + (63 - __clzll((long long)m)) + 1));
    }
}

// Exact replay of bam_plp_push's cap; hist[] counts kept reads by (end - col_lo), zero-initialised, length span+2.
// Only the reads between the first and the last one the cap can possibly drop are replayed one at a time (a read whose stack bound
// is below maxcnt is never dropped, and nothing behind the last candidate depends on the drops): the state in front of the first
// candidate's start position -- every kept read still alive there -- is built by the whole workgroup, then lane 0 walks.  A window with
// one 10 000x amplicon replays 20 000 reads instead of all 860 000 (0.71 s -> ~20 ms).
__global__ void __launch_bounds__(256) k_maxcnt_serial(StaReadsDev R, int maxcnt, int32_t col_lo, int32_t span, int32_t *hist, StaCounters *ctr,
                                                       const unsigned long long *range)
{
    __shared__ long long s_i[3];
    __shared__ unsigned long long s_live;
    if (range[1] == 0) return;
    const int64_t i0 = R.n - (int64_t)range[0], i1 = (int64_t)range[1] - 1;
    const int32_t p0 = R.pos[i0];
    if (threadIdx.x == 0) {
        int64_t lo = 0, hi = i0;                                   // first read starting at p0
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (R.pos[mid] >= p0) hi = mid; else lo = mid + 1; }
        s_i[0] = lo;
        int64_t a = 0, b = lo;                                     // first read whose prefix-maximum end reaches p0
        while (a < b) { int64_t mid = (a + b) >> 1; if (R.maxend[mid] > p0 - 1) b = mid; else a = mid + 1; }
        s_i[1] = a;
        s_live = 0;
    }
    __syncthreads();
    const int64_t s0 = s_i[0], lo0 = s_i[1];
    unsigned long long mine = 0;
    for (int64_t j = lo0 + threadIdx.x; j < s0; j += blockDim.x) {
        const uint32_t info = R.info[j];
        const int32_t e = R.end[j];
        if ((info & RI_PUSHED) && (info & RI_KEEP) && e >= p0) { atomicAdd(&hist[e - col_lo], 1); ++mine; }
    }
    if (mine) atomicAdd(&s_live, mine);
    __syncthreads();
    if (threadIdx.x) return;
    __threadfence();
    long long live = (long long)s_live;
    int32_t cur_p = INT32_MIN, retired = p0 - 1;       // ends <= retired have left the buffer
    unsigned long long dropped = 0;
    for (int64_t i = s0; i <= i1; ++i) {
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

// scratch: span + 2 histogram ints, then two 8-byte range words (8-byte aligned).  R.maxend must hold the prefix maxima of the reads
// as they are BEFORE the cap (the caller runs the scan first and again afterwards).
void sta_launch_maxcnt(hipStream_t s, const StaReadsDev &r, int maxcnt, int32_t col_lo, int32_t span,
                       int32_t *scratch, StaCounters *ctr)
{
    if (r.n == 0) return;
    const size_t hist_ints = ((size_t)span + 2 + 1) & ~(size_t)1;
    hipMemsetAsync(scratch, 0, hist_ints * sizeof(int32_t) + 16, s);
    unsigned long long *range = reinterpret_cast<unsigned long long *>(scratch + hist_ints);
    hipLaunchKernelGGL(k_maxcnt_range, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, maxcnt, range);
    hipLaunchKernelGGL(k_maxcnt_serial, dim3(1), dim3(256), 0, s, r, maxcnt, col_lo, span, scratch, ctr, (const unsigned long long *)range);
}
