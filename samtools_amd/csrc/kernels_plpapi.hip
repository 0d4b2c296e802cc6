// kernels_plpapi.hip -- binary per-column pileup entries for the bam_plp_* / bam_mplp_* callback
// surface (include/samtools_amd_plp.h).  Replaces HTSlib bam_plp64_next + resolve_cigar2
//  for callers that want bam_pileup1_t arrays instead of mpileup text
// (bam_plbuf.c:59-69, bedcov.c:316-333, coverage.c:589, cut_target.c:223-248 in the reference).
//
// Same layout as the text kernels: one wave per 64 columns, one lane per column, reads walked in
// file order so that a column's entries come out in the reference's (linked list) order.
//  k_plp_count : entries per column (every kept read covering the column; no base-quality filter)
//  (scan)      : entry offsets
//  k_plp_fill  : sta_plp_entry {read, qpos, indel, bits} per entry, 16 B, one dwordx4 store each
#include "dev_util.h"

template <bool FILL>
__global__ void __launch_bounds__(256) k_plp_walk(StaWinDev W, uint32_t *line_len, const uint64_t *offs, uint4 *out)
{
    const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)wave * 64;
    if (c0 >= ncols) return;
    const int p0 = W.col_beg + (int)c0;
    const int p = p0 + lane;
    const bool active = p < W.col_end;
    const int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
    const StaReadsDev &R = W.files[0];
    int64_t rlo = 0, rhi = 0;
    if (R.n) {
        rlo = wave_upper_bound(R.maxend, R.n, p0);
        rhi = wave_upper_bound(R.pos, R.n, plast);
        if (rlo > rhi) rlo = rhi;
    }
    uint32_t n = 0;
    uint64_t cur = (FILL && active) ? offs[c0 + lane] : 0;
    for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
        const int64_t ri = b0 + lane;
        const bool ok = ri < rhi;
        const uint32_t v_info = ok ? R.info[ri] : 0u;
        const int v_pos = ok ? R.pos[ri] : 0;
        const int v_end = ok ? R.end[ri] : 0;
        unsigned long long live = __ballot(ok && (v_info & RI_KEEP) && v_end > p0 && v_pos <= plast);
        while (live) {
            const int j = __ffsll((long long)live) - 1; live &= live - 1;
            const uint32_t info = (uint32_t)__builtin_amdgcn_readlane((int)v_info, j);
            const int rpos = __builtin_amdgcn_readlane(v_pos, j), rend = __builtin_amdgcn_readlane(v_end, j);
            if (!(active && p >= rpos && p < rend)) continue;
            if (!FILL) { n++; continue; }
            const int64_t r = b0 + j;
            int qpos = p - rpos, indel = 0, k = 0; bool is_del = false, is_refskip = false;
            if (!(info & RI_SIMPLE))
                plp_resolve(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), rpos, p, qpos, indel, k, is_del, is_refskip);
            uint32_t bits = (is_del ? 1u : 0u) | (p == rpos ? 2u : 0u) | (p == rend - 1 ? 4u : 0u) | (is_refskip ? 8u : 0u) | ((uint32_t)k << 4);
            out[cur++] = make_uint4((uint32_t)r, (uint32_t)qpos, (uint32_t)indel, bits);
        }
    }
    if (!FILL && active) line_len[c0 + lane] = n | (n ? 0x80000000u : 0u);
}

void sta_launch_plp_count(hipStream_t s, const StaWinDev &w, uint32_t *line_len)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    hipLaunchKernelGGL(k_plp_walk<false>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, w, line_len, (const uint64_t *)nullptr, (uint4 *)nullptr);
}

void sta_launch_plp_fill(hipStream_t s, const StaWinDev &w, const uint64_t *offs, void *entries)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    hipLaunchKernelGGL(k_plp_walk<true>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, w, (uint32_t *)nullptr, offs, (uint4 *)entries);
}
