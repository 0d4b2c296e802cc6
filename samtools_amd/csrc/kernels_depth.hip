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
    int32_t pos = R.pos[r], end = R.end[r];
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
