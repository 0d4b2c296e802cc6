// kernels_depth.hip -- samtools depth on the device (gfx950).
//
// Replaces add_depth()/incr_hist[_qual]() (bam2depth.c:165-195, :209-477) and the row formatter
// (:219-245, :289-316, zero_region :88-118).  The reference keeps a ring histogram and bumps
// hist[i]++ for every aligned base; here every read emits +1/-1 *difference* marks (one pair per
// run of counted bases, so a plain 150M read costs two L2 atomics instead of 150 increments) into
// per-file difference rows plus one "covered" row, an inclusive scan turns them into per-column
// counts, and the rows are formatted to text by one wave per 64 columns (LDS staged, coalesced
// flush) exactly like the mpileup emitter.
#include "dev_util.h"
#include <cstdlib>
#include "dev_lookback.h"

extern __shared__ __attribute__((aligned(16))) char lds_dtext[];

struct DepthDevPar { int32_t min_qual, skip_del, all_pos; };

__device__ __forceinline__ void mark_range(int32_t *row, int32_t a, int32_t b, int32_t col_beg, int32_t col_end)
{
    if (a < col_beg) a = col_beg;
    if (b > col_end) b = col_end;
    if (b <= a) return;
    atomicAdd(&row[a - col_beg], 1);
    atomicAdd(&row[b - col_beg], -1);
}

__global__ void __launch_bounds__(256) k_depth_count(StaReadsDev R, int32_t col_beg, int32_t col_end, DepthDevPar P,
                                                    int32_t *file_row, int32_t *cover_row)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n) return;
    if (!(R.info[r] & RI_KEEP)) return;
    int32_t pos = R.pos[r], end = (R.info[r] & RI_UNMAP_SPAN) ? pos + 1 : R.end[r];      // bam_endpos
    mark_range(cover_row, pos, end, col_beg, col_end);
    int32_t clip = R.clip[r];
    bool has_clip = clip != 0;
    int lq = R.l_qseq[r];
    const uint8_t *qual = R.qual + ((uint64_t)R.base_off8[r] << 3);
    int32_t i = pos, spos = 0;
    for (uint32_t k = R.cig_off[r]; k < R.cig_off[r + 1]; ++k) {
        uint32_t c = R.cigar[k];
        int op = c & 0xf; int32_t oplen = (int32_t)(c >> 4);
        if (op == CG_D || op == CG_N) {
            if (op == CG_D && !P.skip_del) {
                int32_t a = i;
                if (has_clip && a < clip) a = clip;
                bool ok = spos < lq ? (int)qual[spos] >= P.min_qual : true;
                if (ok) mark_range(file_row, a, i + oplen, col_beg, col_end);
            }
            i += oplen;
        } else if (cg_is_mop(op)) {
            int32_t a = i, b = i + oplen;
            if (has_clip && a < clip) a = clip;
            if (a < b) {
                if (!P.min_qual) mark_range(file_row, a, b, col_beg, col_end);
                else {
                    // run-length encode the passing bases
                    int32_t lo = a < col_beg ? col_beg : a, hi = b > col_end ? col_end : b;
                    int32_t run = -1;
                    for (int32_t x = lo; x < hi; ++x) {
                        int q = spos + (x - i);
                        bool ok = q < lq ? (int)qual[q] >= P.min_qual : true;
                        if (ok) { if (run < 0) run = x; }
                        else if (run >= 0) { mark_range(file_row, run, x, col_beg, col_end); run = -1; }
                    }
                    if (run >= 0) mark_range(file_row, run, hi, col_beg, col_end);
                }
            }
            spos += oplen; i += oplen;
        } else if (op == CG_I || op == CG_S) spos += oplen;
    }
}

void sta_launch_depth_count(hipStream_t s, const StaWinDev &w, const StaReadsDev *files_host, int nfiles,
                            const sta_depth_params &p, int32_t *diff)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    DepthDevPar d{ p.min_qual, p.skip_del, p.all_pos };
    for (int f = 0; f < nfiles; ++f) {
        const StaReadsDev &R = files_host[f];
        if (R.n == 0) continue;
        unsigned nb = (unsigned)((R.n + 255) / 256);
        hipLaunchKernelGGL(k_depth_count, dim3(nb), dim3(256), 0, s, R, w.col_beg, w.col_end, d,
                           diff + (int64_t)f * (ncols + 1), diff + (int64_t)nfiles * (ncols + 1));
    }
}

// rows: counts[f][c] for f < nfiles, counts[nfiles][c] = number of covering reads
__device__ __forceinline__ bool depth_row_exists(const StaWinDev &W, const DepthDevPar &P, const int32_t *counts,
                                                 int64_t ncols, int64_t c, int64_t apos, bool &covered)
{
    covered = counts[(int64_t)W.nfiles * (ncols + 1) + c] > 0;
    bool ex = covered || (P.all_pos && apos < W.tlen);
    if (ex && W.has_bed) ex = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
    return ex;
}

__global__ void __launch_bounds__(256) k_depth_len(StaWinDev W, DepthDevPar P, const int32_t *counts, uint32_t *line_len, StaCounters *ctr)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t ncols = (int64_t)W.col_end - W.col_beg;
    bool active = c < ncols;
    uint32_t len = 0; bool covered = false, ex = false;
    if (active) {
        int64_t apos = W.origin + W.col_beg + c;
        ex = depth_row_exists(W, P, counts, ncols, c, apos, covered);
        if (ex) {
            len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1;
            for (int f = 0; f < W.nfiles; ++f)
                len += 1 + (uint32_t)dec_digits_u32((uint32_t)counts[(int64_t)f * (ncols + 1) + c]);
        }
        line_len[c] = len | (covered ? 0x80000000u : 0u);     // rows / covered columns are counted by k_col_stats
    }
    (void)ctr;
}

void sta_launch_depth_len(hipStream_t s, const StaWinDev &w, const sta_depth_params &p, const int32_t *counts,
                          uint32_t *line_len, StaCounters *ctr)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    DepthDevPar d{ p.min_qual, p.skip_del, p.all_pos };
    hipLaunchKernelGGL(k_depth_len, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, w, d, counts, line_len, ctr);
}

template <bool LDS> struct DSink {
    uint32_t cur; char *g;
    __device__ __forceinline__ void put(char c) { if (LDS) lds_dtext[cur++] = c; else *g++ = c; }
    __device__ __forceinline__ void put_dec(unsigned long long u)
    {
        int n = dec_digits(u);
        if (LDS) { uint32_t e = cur + n; for (uint32_t q = e; q > cur;) { lds_dtext[--q] = (char)('0' + u % 10); u /= 10; } cur = e; }
        else { char *e = g + n; for (char *q = e; q > g;) { *--q = (char)('0' + u % 10); u /= 10; } g = e; }
    }
};

template <bool LDS>
__device__ __forceinline__ void depth_row_write(const StaWinDev &W, const int32_t *counts, int64_t ncols, int64_t c, DSink<LDS> &s)
{
    int64_t apos = W.origin + W.col_beg + c;
    for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
    s.put('\t');
    s.put_dec((unsigned long long)(apos + 1));
    for (int f = 0; f < W.nfiles; ++f) { s.put('\t'); s.put_dec((uint32_t)counts[(int64_t)f * (ncols + 1) + c]); }
    s.put('\n');
}

__global__ void __launch_bounds__(256) k_depth_emit(StaWinDev W, const int32_t *counts, const uint64_t *__restrict__ offs, char *out, uint32_t lds_cap)
{
    int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int64_t ncols = (int64_t)W.col_end - W.col_beg;
    int64_t c0 = wave * 64;
    if (c0 >= ncols) return;
    int64_t c1 = c0 + 64 < ncols ? c0 + 64 : ncols;
    bool active = c0 + lane < ncols;
    uint64_t o0 = offs[c0], o1 = offs[c1];
    uint64_t my0 = active ? offs[c0 + lane] : o1, my1 = active ? offs[c0 + lane + 1] : o1;
    uint64_t wbytes = o1 - o0;
    if (wbytes == 0) return;
    if (wbytes <= lds_cap) {
        uint32_t slice = (lds_cap + 16 + 15) & ~15u;
        uint32_t base = (uint32_t)wid * slice;
        uint32_t mis = (uint32_t)((uintptr_t)(out + o0) & 15);
        DSink<true> s; s.g = nullptr; s.cur = base + mis + (uint32_t)(my0 - o0);
        if (my1 > my0) depth_row_write<true>(W, counts, ncols, c0 + lane, s);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        char *dst = out + o0;
        uint32_t n = (uint32_t)wbytes;
        uint32_t head = mis ? 16 - mis : 0; if (head > n) head = n;
        if ((uint32_t)lane < head) dst[lane] = lds_dtext[base + mis + lane];
        uint32_t body = (n - head) >> 4;
        const uint4 *src4 = reinterpret_cast<const uint4 *>(lds_dtext + base + mis + head);
        uint4 *dst4 = reinterpret_cast<uint4 *>(dst + head);
        for (uint32_t i = lane; i < body; i += 64) dst4[i] = src4[i];
        uint32_t done = head + (body << 4);
        if (done + lane < n) dst[done + lane] = lds_dtext[base + mis + done + lane];
    } else {
        DSink<false> s; s.cur = 0; s.g = out + my0;
        if (my1 > my0) depth_row_write<false>(W, counts, ncols, c0 + lane, s);
    }
}

void sta_launch_depth_emit(hipStream_t s, const StaWinDev &w, const sta_depth_params &p, const int32_t *counts,
                           const uint64_t *offs, char *out, uint32_t lds_cap)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    uint32_t slice = (lds_cap + 16 + 15) & ~15u;
    hipLaunchKernelGGL(k_depth_emit, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 4 * slice, s, w, counts, offs, out, lds_cap);
}


// ================================================================================================
// Single-pass depth: per-column counts, row lengths, offsets (decoupled look-back, dev_lookback.h) and text in ONE launch.
// Instead of difference marks + atomics + scans, a wave walks the reads that can touch its 64 columns (the same contiguous
// read range and uniform walk as the mpileup kernels; k_prep_reads_depth + the maxend scan provide info / end / maxend) and
// every lane counts its own column: a read that is one M run costs a handful of integer instructions per (read, wave).
// The counts are also left in `counts` ([nfiles + 1][ncols + 1] int32, last row = covering reads) for sta_depth_counts_dev.
struct DepthFusedArgs {
    unsigned long long *status; unsigned int *ticket;
    char *out; unsigned long long capacity;
    int32_t *counts;
    StaCounters *ctr;
    uint32_t lbuf, per_wave, n_tiles, tiles_per_batch;
    int32_t has_clip;
};

// aligned bases of file R covering column p (bam2depth.c:396-424 rules: M/=/X counted under -q, D only with -J and judged by
// the quality of the next query base, N never; -s clips below `clip`), and whether p lies in the read's covered span
__device__ __forceinline__ void depth_walk(const StaReadsDev &R, const DepthDevPar &P, int has_clip, int p0, int plast, int p, bool active,
                                           int64_t rlo, int64_t rhi, uint32_t &cnt, uint32_t &cover)
{
    const int lane = threadIdx.x & 63;
    const auto g_info = (const __attribute__((address_space(1))) uint32_t *)R.info;
    const auto g_pos = (const __attribute__((address_space(1))) int32_t *)R.pos;
    const auto g_end = (const __attribute__((address_space(1))) int32_t *)R.end;
    const auto g_b8 = (const __attribute__((address_space(1))) uint32_t *)R.base_off8;
    const auto g_clip = (const __attribute__((address_space(1))) int32_t *)R.clip;
    const auto g_qual = (const __attribute__((address_space(1))) uint8_t *)R.qual;
    for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
        const int64_t ri = b0 + lane;
        const bool ok = ri < rhi;
        const uint32_t v_info = ok ? g_info[ri] : 0u;
        const int v_pos = ok ? g_pos[ri] : 0;
        const int v_end = ok ? g_end[ri] : 0;
        const uint32_t v_b8 = (ok && P.min_qual) ? g_b8[ri] : 0u;
        const int v_clip = (ok && has_clip) ? g_clip[ri] : 0;
        unsigned long long live = __ballot(ok && (v_info & RI_KEEP) && v_end > p0 && v_pos <= plast);
        while (live) {
            const int j = __ffsll((long long)live) - 1;
            live &= live - 1;
            const uint32_t info = (uint32_t)__builtin_amdgcn_readlane((int)v_info, j);
            const int rpos = __builtin_amdgcn_readlane(v_pos, j), rend = __builtin_amdgcn_readlane(v_end, j);
            const int clip = __builtin_amdgcn_readlane(v_clip, j);
            if (info & RI_SIMPLE) {
                const bool cov = active && p >= rpos && p < rend;
                bool okb = cov && (!clip || p >= clip);
                if (P.min_qual) {
                    const uint64_t boff = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v_b8, j) << 3;
                    const int q = okb ? (int)g_qual[boff + (uint64_t)(p - rpos)] : 0;
                    okb = okb && q >= P.min_qual;
                }
                cover += cov ? 1u : 0u;
                cnt += okb ? 1u : 0u;
            } else {
                const int cend = (info & RI_UNMAP_SPAN) ? rpos + 1 : rend;           // bam_endpos
                cover += (active && p >= rpos && p < cend) ? 1u : 0u;
                const bool in = active && p >= rpos && p < rend;
                if (__ballot(in) == 0) continue;
                if (in) {
                    const int64_t r = b0 + j;
                    const int lq = R.l_qseq[r];
                    const uint8_t *qual = R.qual + ((uint64_t)R.base_off8[r] << 3);
                    int32_t i = rpos, spos = 0;
                    for (uint32_t k = R.cig_off[r]; k < R.cig_off[r + 1]; ++k) {
                        const uint32_t c = R.cigar[k];
                        const int op = c & 0xf; const int32_t oplen = (int32_t)(c >> 4);
                        if (op == CG_D || op == CG_N) {
                            if (p >= i && p < i + oplen) {
                                if (op == CG_D && !P.skip_del) {
                                    const bool okq = spos < lq ? (int)qual[spos] >= P.min_qual : true;
                                    if (okq && (!clip || p >= clip)) cnt++;
                                }
                                break;
                            }
                            i += oplen;
                        } else if (cg_is_mop(op)) {
                            if (p >= i && p < i + oplen) {
                                const int q = spos + (p - i);
                                const bool okq = !P.min_qual || (q < lq ? (int)qual[q] >= P.min_qual : true);
                                if (okq && (!clip || p >= clip)) cnt++;
                                break;
                            }
                            spos += oplen; i += oplen;
                        } else if (op == CG_I || op == CG_S) spos += oplen;
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ void depth_read_range(const StaReadsDev &R, int p0, int p1, int64_t &rlo, int64_t &rhi)
{
    if (R.n == 0) { rlo = rhi = 0; return; }
    rlo = wave_upper_bound(R.maxend, R.n, p0);
    rhi = wave_upper_bound(R.pos, R.n, p1);
    if (rlo > rhi) rlo = rhi;
}
// the next tile's range from the previous one's: both bounds only move forward (see wave_read_range_next in kernels_plp.hip)
__device__ __forceinline__ void depth_read_range_next(const StaReadsDev &R, int p0, int p1, int64_t &rlo, int64_t &rhi, bool have_prev)
{
    if (R.n == 0) { rlo = rhi = 0; return; }
    if (!have_prev) { depth_read_range(R, p0, p1, rlo, rhi); return; }
    const int lane = threadIdx.x & 63;
    bool found = false;
    for (int it = 0; it < 3 && !found; ++it) {
        const int64_t idx = rlo + lane;
        const unsigned long long m = __ballot(idx < R.n ? R.maxend[idx] > p0 : true);
        if (m) { rlo += __ffsll((long long)m) - 1; found = true; } else rlo += 64;
    }
    if (!found) rlo = wave_upper_bound(R.maxend, R.n, p0);
    if (rlo > R.n) rlo = R.n;
    found = false;
    for (int it = 0; it < 3 && !found; ++it) {
        const int64_t idx = rhi + lane;
        const unsigned long long m = __ballot(idx < R.n ? R.pos[idx] > p1 : true);
        if (m) { rhi += __ffsll((long long)m) - 1; found = true; } else rhi += 64;
    }
    if (!found) rhi = wave_upper_bound(R.pos, R.n, p1);
    if (rhi > R.n) rhi = R.n;
    if (rlo > rhi) rlo = rhi;
}

__global__ void __launch_bounds__(256) k_depth_fused(StaWinDev W, DepthDevPar P, DepthFusedArgs A)
{
    __shared__ unsigned int s_batch;
    __shared__ unsigned long long s_wtot[4][2];
    __shared__ unsigned long long s_base[2];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_batch = atomicAdd(A.ticket, 1u);      // one ticket per batch of consecutive tiles (dev_lookback.h)
    __syncthreads();
    const unsigned int batch = s_batch;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const uint32_t lb = (uint32_t)wid * A.per_wave;
    int64_t rlo1 = 0, rhi1 = 0; bool have_range = false;

    for (unsigned ti = 0; ti < A.tiles_per_batch; ++ti) {
        const unsigned int tile = batch * A.tiles_per_batch + ti;
        if (tile >= A.n_tiles) break;
        __syncthreads();
        const int64_t c0 = ((int64_t)tile * 4 + wid) * 64;
        const bool wave_on = c0 < ncols;
        const int p0 = W.col_beg + (int)(wave_on ? c0 : 0);
        const int p = p0 + lane;
        const bool active = wave_on && p < W.col_end;
        const int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
        const int64_t apos = W.origin + p;
        const int64_t col = c0 + lane;

        // ---- COUNT ----
        uint32_t len = 0; bool covered = false, exists = false;
        if (wave_on) {
            uint32_t cover = 0, digits = 0;
            for (int f = 0; f < W.nfiles; ++f) {
                const StaReadsDev &R = W.files[f];
                int64_t rlo, rhi;
                if (W.nfiles == 1) { depth_read_range_next(R, p0, plast, rlo1, rhi1, have_range); have_range = true; rlo = rlo1; rhi = rhi1; }
                else depth_read_range(R, p0, plast, rlo, rhi);
                uint32_t cnt = 0;
                depth_walk(R, P, A.has_clip, p0, plast, p, active, rlo, rhi, cnt, cover);
                if (active) A.counts[(int64_t)f * (ncols + 1) + col] = (int32_t)cnt;
                digits += 1 + (uint32_t)dec_digits_u32(cnt);
            }
            if (active) A.counts[(int64_t)W.nfiles * (ncols + 1) + col] = (int32_t)cover;
            covered = active && cover > 0;
            exists = active && (covered || (P.all_pos && apos < W.tlen));
            if (exists && W.has_bed) exists = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
            if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + digits + 1;
        }
        uint32_t incl = len;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const uint32_t excl = incl - len;
        const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const unsigned long long n_rows = (unsigned long long)__popcll(__ballot(exists)), n_cov = (unsigned long long)__popcll(__ballot(covered));
        if (lane == 0) { s_wtot[wid][0] = wave_total; s_wtot[wid][1] = (n_rows << 31) | n_cov; }
        __syncthreads();
        if (wid == 0) {
            const unsigned long long agg0 = s_wtot[0][0] + s_wtot[1][0] + s_wtot[2][0] + s_wtot[3][0];
            const unsigned long long agg1 = s_wtot[0][1] + s_wtot[1][1] + s_wtot[2][1] + s_wtot[3][1];
            unsigned long long ex0, ex1;
            tile_lookback(A.status, tile, agg0, agg1, ex0, ex1);
            if (lane == 0) {
                s_base[0] = ex0; s_base[1] = agg0;
                if (tile + 1 == A.n_tiles) {
                    A.ctr->out_bytes = ex0 + agg0;
                    A.ctr->n_lines = (ex1 + agg1) >> 31;
                    A.ctr->n_data_cols = (ex1 + agg1) & 0x7fffffffull;
                }
            }
        }
        __syncthreads();
        const unsigned long long wg_off = s_base[0], wg_bytes = s_base[1];
        if (wg_off + wg_bytes > A.capacity) { if (threadIdx.x == 0) A.ctr->overflow = 1; continue; }
        if (!wave_on || wave_total == 0) continue;
        unsigned long long wave_off = wg_off;
        for (int w = 0; w < wid; ++w) wave_off += s_wtot[w][0];

        // ---- EMIT: rows into the wave's LDS line buffer (rounds of consecutive rows when many input files make them long) ----
        int a = 0;
        while (a < 64) {
            const uint32_t start = (uint32_t)__shfl((int)excl, a);
            const bool fits = lane >= a && incl - start <= A.lbuf;
            const int nb = __popcll(__ballot(fits));
            if (nb == 0) {
                if (lane == a && exists) { DSink<false> s; s.cur = 0; s.g = A.out + wave_off + excl; depth_row_write<false>(W, A.counts, ncols, col, s); }
                a += 1;
                continue;
            }
            const int b = a + nb;
            const uint32_t rbytes = (uint32_t)__shfl((int)incl, b - 1) - start;
            if (rbytes) {
                char *dst = A.out + wave_off + start;
                const uint32_t mis = (uint32_t)((uintptr_t)dst & 15);
                wave_lds_sync();
                if (lane >= a && lane < b && exists) { DSink<true> s; s.g = nullptr; s.cur = lb + mis + (excl - start); depth_row_write<true>(W, A.counts, ncols, col, s); }
                wave_lds_sync();
                wave_flush_text(lds_dtext + lb + mis, dst, rbytes);
            }
            a = b;
        }
    }
}

size_t sta_depth_fused_status_bytes(int64_t ncols) { return (size_t)((ncols + 255) / 256) * 16 + 16; }

void sta_launch_depth_fused(hipStream_t s, const StaWinDev &w, const sta_depth_params &p, void *status, int32_t *counts, char *out,
                            uint64_t capacity, StaCounters *ctr, uint32_t lbuf)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    const int64_t n_tiles = (ncols + 255) / 256;
    hipMemsetAsync(status, 0, sta_depth_fused_status_bytes(ncols), s);
    DepthFusedArgs a;
    a.status = (unsigned long long *)status;
    a.ticket = (unsigned int *)((char *)status + (size_t)n_tiles * 16);
    a.out = out; a.capacity = capacity; a.counts = counts; a.ctr = ctr;
    a.lbuf = lbuf; a.per_wave = ((lbuf + 16 + 15) & ~15u) + 16; a.n_tiles = (uint32_t)n_tiles;
    a.has_clip = p.remove_overlaps ? 1 : 0;
    DepthDevPar d{ p.min_qual, p.skip_del, p.all_pos };
    int64_t tpb = 1;      // consecutive tiles per workgroup serialise the look-back chain (measured: 400x slower); kept as an experiment knob
    { static const char *ev = getenv("STA_FUSED_TPB"); if (ev && atoi(ev) > 0) tpb = atoi(ev); }
    a.tiles_per_batch = (uint32_t)tpb;
    hipLaunchKernelGGL(k_depth_fused, dim3((unsigned)((n_tiles + tpb - 1) / tpb)), dim3(256), (size_t)4 * a.per_wave, s, w, d, a);
}
