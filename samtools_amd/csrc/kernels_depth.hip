// kernels_depth.hip -- samtools depth on the device (gfx950).
//
// Replaces add_depth()/incr_hist[_qual]() (bam2depth.c:165-195, :209-477) and the row formatter
// (:219-245, :289-316, zero_region :88-118).  The reference keeps a ring histogram and bumps
// hist[i]++ for every aligned base; here ONE kernel (k_depth_fused, below) counts every column from
// +1 / -1 difference marks in LDS, places the rows with a decoupled look-back and writes the text.
#include "dev_util.h"
#include <cstdlib>
#include "dev_lookback.h"

extern __shared__ __attribute__((aligned(16))) char lds_dtext[];

struct DepthDevPar { int32_t min_qual, skip_del, all_pos; };

// rows: counts[f][c] for f < nfiles, counts[nfiles][c] = number of covering reads
__device__ __forceinline__ bool depth_row_exists(const StaWinDev &W, const DepthDevPar &P, const int32_t *counts,
                                                 int64_t ncols, int64_t c, int64_t apos, bool &covered)
{
    covered = counts[(int64_t)W.nfiles * (ncols + 1) + c] > 0;
    bool ex = covered || (P.all_pos && apos < W.tlen);
    if (ex && W.has_bed) ex = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
    return ex;
}

// digits of a coordinate: nearly always below 2^32, where a compare chain replaces a loop of 64-bit divisions
__device__ __forceinline__ int depth_dec_digits(unsigned long long u) { return u <= 0xffffffffull ? dec_digits_u32((uint32_t)u) : dec_digits(u); }

template <bool LDS> struct DSink {
    uint32_t cur; char *g;
    __device__ __forceinline__ void put(char c) { if (LDS) lds_dtext[cur++] = c; else *g++ = c; }
    __device__ __forceinline__ void put_dec(unsigned long long u)
    {
        int n = depth_dec_digits(u);
        if (LDS && u <= 0xffffffffull) {
            uint32_t w = (uint32_t)u; const uint32_t e = cur + n;         // 32-bit digits: multiply-high + shift per digit
            for (uint32_t q = e; q > cur;) { const uint32_t d = w / 10u; lds_dtext[--q] = (char)('0' + (w - d * 10u)); w = d; }
            cur = e;
        }
        else if (LDS) { uint32_t e = cur + n; for (uint32_t q = e; q > cur;) { lds_dtext[--q] = (char)('0' + u % 10); u /= 10; } cur = e; }
        else { char *e = g + n; for (char *q = e; q > g;) { *--q = (char)('0' + u % 10); u /= 10; } g = e; }
    }
};

// the contig name, fetched once per wave (wave-uniform loads) instead of once per row: up to 16 characters in two registers
struct DName { unsigned long long lo, hi; int len; };
__device__ __forceinline__ DName depth_name(const StaWinDev &W)
{
    DName n; n.lo = 0; n.hi = 0; n.len = W.tname_len;
    if (n.len <= 16) for (int t = 0; t < n.len; ++t) { const unsigned long long ch = (unsigned char)W.tname[t]; if (t < 8) n.lo |= ch << (8 * t); else n.hi |= ch << (8 * (t - 8)); }
    return n;
}

template <bool LDS>
__device__ __forceinline__ void depth_row_write(const StaWinDev &W, const DName &nm, const int32_t *counts, int64_t ncols, int64_t c, DSink<LDS> &s)
{
    int64_t apos = W.origin + W.col_beg + c;
    if (nm.len <= 16) for (int t = 0; t < nm.len; ++t) s.put((char)((t < 8 ? nm.lo >> (8 * t) : nm.hi >> (8 * (t - 8))) & 0xff));
    else for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
    s.put('\t');
    s.put_dec((unsigned long long)(apos + 1));
    for (int f = 0; f < W.nfiles; ++f) { s.put('\t'); s.put_dec((uint32_t)counts[(int64_t)f * (ncols + 1) + c]); }
    s.put('\n');
}



// ================================================================================================
// Single-pass depth: per-column counts, row lengths, offsets (decoupled look-back, dev_lookback.h) and text in ONE launch.
//
// A workgroup takes one ticket for 4 x 512 consecutive columns; each of its waves owns 512 of them, EIGHT consecutive columns per
// lane.  COUNT: the reads that can touch the wave's span (one contiguous index range) are taken ONE LANE PER READ; every lane drops
// +1 / -1 difference marks for its read's counted runs into a 513-entry LDS array (clipped to the span; a plain 150M read costs two
// LDS atomics); each lane then sums its eight entries and one wave scan turns the marks into the 512 column counts.  That is O(reads)
// work instead of O(reads x columns / 64), and one memory round trip per phase for 512 columns.  The counts go to `counts`
// ([nfiles + 1][ncols + 1] int32, last row = covering reads: sta_depth_counts_dev) and give the row lengths; ONE look-back per
// workgroup places its text; EMIT: every lane formats its eight rows back to back into the wave's LDS line buffer, 16-byte flush.
#define DF_CPL 8                      // columns per lane
#define DF_SPAN (64 * DF_CPL)         // columns per wave
struct DepthFusedArgs {
    unsigned long long *status; unsigned int *ticket;
    char *out; unsigned long long capacity;
    int32_t *counts;
    StaCounters *ctr;
    uint32_t lbuf, per_wave, n_tiles;
    int32_t has_clip;
    int32_t diag;          // timing diagnostics only (STA_DEPTH_DIAG; wrong text): 1 = no EMIT phase, 2 = no look-back wait, 3 = no COUNT marks
    uint32_t *lens;        // split form: the row length of every column (k_depth_fused<1> -> <2>)
};

__device__ __forceinline__ void lds_mark(int *row, int a, int b, int p0)
{
    // [a, b) clipped to the wave's span [p0, p0 + DF_SPAN)
    a = a < p0 ? p0 : a;
    b = b > p0 + DF_SPAN ? p0 + DF_SPAN : b;
    if (b <= a) return;
    atomicAdd(&row[a - p0], 1);
    atomicAdd(&row[b - p0], -1);
}

// difference marks of the reads [rlo, rhi) of file R for the span starting at p0 (bam2depth.c:396-424 rules: M/=/X counted
// under -q, D only with -J and judged by the quality of the next query base, N never; -s clips below `clip`)
__device__ __forceinline__ void depth_marks(const StaReadsDev &R, const DepthDevPar &P, int has_clip, int p0, int plast,
                                            int64_t rlo, int64_t rhi, int *d_file, int *d_cover)
{
    const int lane = threadIdx.x & 63;
    for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
        const int64_t r = b0 + lane;
        if (r >= rhi) continue;
        const uint32_t info = R.info[r];
        const int rpos = R.pos[r], rend = R.end[r];
        if (!(info & RI_KEEP) || rend <= p0 || rpos > plast) continue;
        lds_mark(d_cover, rpos, (info & RI_UNMAP_SPAN) ? rpos + 1 : rend, p0);          // bam_endpos
        const int clip = has_clip ? R.clip[r] : 0;
        if ((info & RI_SIMPLE) && !P.min_qual) { lds_mark(d_file, clip && rpos < clip ? clip : rpos, rend, p0); continue; }
        const int lq = R.l_qseq[r];
        const uint8_t *qual = R.qual + ((uint64_t)R.base_off8[r] << 3);
        int32_t i = rpos, spos = 0;
        for (uint32_t k = R.cig_off[r]; k < R.cig_off[r + 1] && i <= plast; ++k) {
            const uint32_t c = R.cigar[k];
            const int op = c & 0xf; const int32_t oplen = (int32_t)(c >> 4);
            if (op == CG_D || op == CG_N) {
                if (op == CG_D && !P.skip_del) {
                    int32_t a = i;
                    if (clip && a < clip) a = clip;
                    const bool okq = spos < lq ? (int)qual[spos] >= P.min_qual : true;
                    if (okq) lds_mark(d_file, a, i + oplen, p0);
                }
                i += oplen;
            } else if (cg_is_mop(op)) {
                int32_t a = i, b = i + oplen;
                if (clip && a < clip) a = clip;
                if (a < b) {
                    if (!P.min_qual) lds_mark(d_file, a, b, p0);
                    else {
                        // run-length encode the passing bases inside the span
                        const int32_t lo = a < p0 ? p0 : a, hi = b > p0 + DF_SPAN ? p0 + DF_SPAN : b;
                        int32_t run = -1;
                        for (int32_t x = lo; x < hi; ++x) {
                            const int q = spos + (x - i);
                            const bool okq = q < lq ? (int)qual[q] >= P.min_qual : true;
                            if (okq) { if (run < 0) run = x; }
                            else if (run >= 0) { lds_mark(d_file, run, x, p0); run = -1; }
                        }
                        if (run >= 0) lds_mark(d_file, run, hi, p0);
                    }
                }
                spos += oplen; i += oplen;
            } else if (op == CG_I || op == CG_S) spos += oplen;
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

__device__ __forceinline__ int wave_excl_scan_i32(int v, int &total)
{
    const int lane = threadIdx.x & 63;
    int x = v;
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

// marks -> counts for the lane's DF_CPL consecutive columns (entries d[DF_CPL * lane ..]); written to the counts row
__device__ __forceinline__ void depth_counts_from_marks(const int *d, int32_t *row, int64_t c0, int64_t ncols)
{
    const int lane = threadIdx.x & 63;
    int v[DF_CPL], sum = 0;
#pragma unroll
    for (int k = 0; k < DF_CPL; ++k) { v[k] = d[DF_CPL * lane + k]; sum += v[k]; }
    int tot;
    int run = wave_excl_scan_i32(sum, tot);
#pragma unroll
    for (int k = 0; k < DF_CPL; ++k) {
        run += v[k];
        const int64_t col = c0 + DF_CPL * lane + k;
        if (col < ncols) row[col] = run;
    }
}

// row length of column `col` from the stored counts (0: the row is not printed); `covered` = some read spans the column
__device__ __forceinline__ uint32_t depth_row_len(const StaWinDev &W, const DepthDevPar &P, const int32_t *counts, int64_t ncols, int64_t col,
                                                  bool active, bool &covered)
{
    covered = false;
    if (!active) return 0;
    const int64_t apos = W.origin + W.col_beg + col;
    bool ex = depth_row_exists(W, P, counts, ncols, col, apos, covered);
    if (!ex) return 0;
    uint32_t len = (uint32_t)W.tname_len + 1 + (uint32_t)depth_dec_digits((unsigned long long)(apos + 1)) + 1;
    for (int f = 0; f < W.nfiles; ++f) len += 1 + (uint32_t)dec_digits_u32((uint32_t)counts[(int64_t)f * (ncols + 1) + col]);
    return len;
}

// PHASE 0: the single launch described above.  The SPLIT form (round 6; STA_DEPTH_FORM=split) is the same code in two launches with a
// one-workgroup scan between them: PHASE 1 = COUNT, the wave's text bytes and rows into status[2 (4 tile + wave)], every column's row
// length into `lens`; k_depth_wave_scan turns the bytes into offsets and adds up the window's totals; PHASE 2 = EMIT from the stored
// lengths.  No ticket (2 048 atomics on one address: 23 us) and no look-back chain (51 us with every tile resident and reaching it
// together: profiles/r05_depth_phases.md); two more launches and 8 bytes per column through memory instead.
template <int PHASE>
__global__ void __launch_bounds__(256) k_depth_fused(StaWinDev W, DepthDevPar P, DepthFusedArgs A)
{
    __shared__ unsigned int s_tile;
    __shared__ unsigned long long s_wtot[4][2];
    __shared__ unsigned long long s_base[2];
    __shared__ uint32_t s_len[4][DF_SPAN];
    // the difference marks of the COUNT phase and the line buffers of the EMIT phase are never live at the same time (the
    // look-back barrier lies between them): they share the dynamic LDS block, which keeps a workgroup at 16.5 KB
    int *s_diff = reinterpret_cast<int *>(lds_dtext);                 // [4 waves][2][DF_SPAN + 4]
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned int tile;
    if (PHASE != 0) tile = blockIdx.x;
    else if (A.ticket) {
        if (threadIdx.x == 0) s_tile = atomicAdd(A.ticket, 1u);      // tiles are handed out in start order (dev_lookback.h)
        __syncthreads();
        tile = s_tile;
    } else tile = blockIdx.x;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = ((int64_t)tile * 4 + wid) * DF_SPAN;          // this wave's first column
    const bool wave_on = c0 < ncols;
    const uint32_t lb = (uint32_t)wid * A.per_wave;
    int *d_file = s_diff + (size_t)(wid * 2) * (DF_SPAN + 4), *d_cover = d_file + (DF_SPAN + 4);

    // ---- COUNT ----
    uint32_t len[DF_CPL]; uint32_t lane_len = 0;
    unsigned long long n_rows = 0, n_cov = 0;
#pragma unroll
    for (int k = 0; k < DF_CPL; ++k) len[k] = 0;
    if (PHASE == 2) {
        if (wave_on) {
#pragma unroll
            for (int k = 0; k < DF_CPL; ++k) { const int64_t col = c0 + DF_CPL * lane + k; len[k] = col < ncols ? A.lens[col] : 0u; lane_len += len[k]; }
        }
    } else
    if (wave_on) {
        const int p0 = W.col_beg + (int)c0;
        const int plast = c0 + DF_SPAN < ncols ? p0 + DF_SPAN - 1 : W.col_end - 1;
#pragma unroll
        for (int k = 0; k < DF_CPL; ++k) d_cover[DF_CPL * lane + k] = 0;
        if (lane == 0) d_cover[DF_SPAN] = 0;
        for (int f = 0; f < W.nfiles; ++f) {
            const StaReadsDev &R = W.files[f];
            int64_t rlo, rhi;
            depth_read_range(R, p0, plast, rlo, rhi);
#pragma unroll
            for (int k = 0; k < DF_CPL; ++k) d_file[DF_CPL * lane + k] = 0;
            if (lane == 0) d_file[DF_SPAN] = 0;
            wave_lds_sync();
            if (A.diag != 3) depth_marks(R, P, A.has_clip, p0, plast, rlo, rhi, d_file, d_cover);
            wave_lds_sync();
            depth_counts_from_marks(d_file, A.counts + (int64_t)f * (ncols + 1), c0, ncols);
            wave_lds_sync();                                         // d_file is zeroed again for the next file
        }
        depth_counts_from_marks(d_cover, A.counts + (int64_t)W.nfiles * (ncols + 1), c0, ncols);
#pragma unroll
        for (int k = 0; k < DF_CPL; ++k) {
            const int64_t col = c0 + DF_CPL * lane + k;
            bool covered;
            len[k] = depth_row_len(W, P, A.counts, ncols, col, col < ncols, covered);
            lane_len += len[k];
            n_rows += len[k] > 0; n_cov += covered;
        }
    }
    int wave_total_i;
    (void)wave_excl_scan_i32((int)lane_len, wave_total_i);
    const uint32_t wave_total = (uint32_t)wave_total_i;
    if (PHASE == 1) {
        n_rows = wave_sum_u64(n_rows); n_cov = wave_sum_u64(n_cov);
        if (lane == 0) { unsigned long long *st = A.status + 2 * ((size_t)tile * 4 + wid); st[0] = wave_total; st[1] = (n_rows << 31) | n_cov; }
        if (wave_on) {
#pragma unroll
            for (int k = 0; k < DF_CPL; ++k) { const int64_t col = c0 + DF_CPL * lane + k; if (col < ncols) A.lens[col] = len[k]; }
        }
        return;
    }
    unsigned long long off;
    if (PHASE == 2) {
        off = A.status[2 * ((size_t)tile * 4 + wid)];              // (k_depth_wave_scan: the exclusive prefix of the waves' bytes)
        if (off + wave_total > A.capacity) return;                  // counted, not written: the host retries with room (the scan set the flag)
        if (!wave_on || wave_total == 0) return;
    } else {
    n_rows = wave_sum_u64(n_rows); n_cov = wave_sum_u64(n_cov);
    if (lane == 0) { s_wtot[wid][0] = wave_total; s_wtot[wid][1] = (n_rows << 31) | n_cov; }
    __syncthreads();
    if (wid == 0) {
        const unsigned long long agg0 = s_wtot[0][0] + s_wtot[1][0] + s_wtot[2][0] + s_wtot[3][0];
        const unsigned long long agg1 = s_wtot[0][1] + s_wtot[1][1] + s_wtot[2][1] + s_wtot[3][1];
        unsigned long long ex0, ex1;
        if (A.diag == 2) { ex0 = (unsigned long long)tile * 30000ull; ex1 = 0; }
        else tile_lookback(A.status, tile, agg0, agg1, ex0, ex1);
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
    if (wg_off + wg_bytes > A.capacity) { if (threadIdx.x == 0) A.ctr->overflow = 1; return; }     // counted, not written: the host retries with room
    if (!wave_on || wave_total == 0 || A.diag == 1) return;
    off = wg_off;
    for (int w = 0; w < wid; ++w) off += s_wtot[w][0];
    }

    // ---- EMIT: 64 consecutive rows at a time, lane j formats row j of the group.  (A lane formatting its own 8 consecutive rows
    // puts the lanes 8 x 16 = 128 bytes apart in the line buffer: with the usual 16-byte rows every byte store of the wave hit
    // two LDS banks, 32-way conflicts, and the kernel was LDS-bound -- SQ_LDS_BANK_CONFLICT 118 M of 129 M active cycles.
    // Neighbouring lanes now write neighbouring rows: 4-way.)  Row lengths travel through LDS in column order. ----
#pragma unroll
    for (int k = 0; k < DF_CPL; ++k) s_len[wid][DF_CPL * lane + k] = len[k];
    wave_lds_sync();
    const DName dname = depth_name(W);
    uint32_t gbase = 0;                                              // bytes of this wave's earlier groups
    for (int k = 0; k < DF_CPL; ++k) {
        const int m = k * 64 + lane;
        const uint32_t l = s_len[wid][m];
        int gtot_i;
        const uint32_t ro = (uint32_t)wave_excl_scan_i32((int)l, gtot_i);
        const uint32_t gtot = (uint32_t)gtot_i;
        if (gtot == 0) continue;
        char *dst = A.out + off + gbase;
        const int64_t col = c0 + m;
        if (gtot <= A.lbuf) {
            const uint32_t mis = (uint32_t)((uintptr_t)dst & 15);
            wave_lds_sync();
            if (l) { DSink<true> sk; sk.g = nullptr; sk.cur = lb + mis + ro; depth_row_write<true>(W, dname, A.counts, ncols, col, sk); }
            wave_lds_sync();
            wave_flush_text(lds_dtext + lb + mis, dst, gtot);
        } else if (l) {
            // the 64 rows exceed the line buffer (hundreds of input files): straight to global memory
            DSink<false> sk; sk.cur = 0; sk.g = dst + ro;
            depth_row_write<false>(W, dname, A.counts, ncols, col, sk);
        }
        gbase += gtot;
    }
}

struct alignas(16) DU64x2 { unsigned long long x, y; };
// the split form's scan: status[2 i] = bytes of wave i (four waves per tile) -> their exclusive prefix; the window's totals (and the
// overflow flag) into the counters.  One workgroup; a thread takes whole tiles (64 contiguous bytes each, asked for together), the
// per-thread sums are scanned wave by wave (shuffles) and across the sixteen waves by one of them: two barriers in all.
__global__ void __launch_bounds__(1024) k_depth_wave_scan(unsigned long long *__restrict__ status, int64_t n_tiles, unsigned long long capacity, StaCounters *ctr)
{
    __shared__ unsigned long long s_w[16][2];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int64_t per = (n_tiles + 1023) / 1024, a = (int64_t)t * per, b = a + per < n_tiles ? a + per : n_tiles;
    unsigned long long sum = 0, rows = 0;
    for (int64_t i = a; i < b; ++i) {
        const DU64x2 *q = reinterpret_cast<const DU64x2 *>(status + 8 * i);
        const DU64x2 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
        sum += w0.x + w1.x + w2.x + w3.x; rows += w0.y + w1.y + w2.y + w3.y;
    }
    unsigned long long isum = sum, irows = rows;                   // inclusive scans over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long v = __shfl_up(isum, o), r = __shfl_up(irows, o);
        if (lane >= o) { isum += v; irows += r; }
    }
    if (lane == 63) { s_w[wid][0] = isum; s_w[wid][1] = irows; }
    __syncthreads();
    unsigned long long base = 0, tot = 0, tot_rows = 0;
    for (int w = 0; w < 16; ++w) { const unsigned long long v = s_w[w][0]; if (w < wid) base += v; tot += v; tot_rows += s_w[w][1]; }
    unsigned long long run = base + isum - sum;                    // bytes in front of this thread's first tile
    for (int64_t i = a; i < b; ++i) {
        DU64x2 *q = reinterpret_cast<DU64x2 *>(status + 8 * i);
#pragma unroll
        for (int w = 0; w < 4; ++w) { DU64x2 v = q[w]; const unsigned long long bytes = v.x; v.x = run; q[w] = v; run += bytes; }
    }
    if (t == 0) {
        ctr->out_bytes = tot;
        ctr->n_lines = tot_rows >> 31; ctr->n_data_cols = tot_rows & 0x7fffffffull;
        if (tot > capacity) ctr->overflow = 1;
    }
}

// (two words per WAVE for the split form -- four waves per tile -- and the ticket behind them)
size_t sta_depth_fused_status_bytes(int64_t ncols) { const int64_t t = 4 * DF_SPAN; return (size_t)((ncols + t - 1) / t) * 64 + 16; }

void sta_launch_depth_fused(hipStream_t s, const StaWinDev &w, const sta_depth_params &p, void *status, int32_t *counts, char *out,
                            uint64_t capacity, StaCounters *ctr, uint32_t lbuf, bool status_zeroed, uint32_t *lens)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    const int64_t tcols = 4 * DF_SPAN;
    const int64_t n_tiles = (ncols + tcols - 1) / tcols;
    // which form: the single launch.  The split one (STA_DEPTH_FORM=split) was measured at bench size in round 6 (profiles/r06_sessionK_depth_forms.log):
    // count 0.107 + wave scan 0.026 + emit 0.076 = 0.210 ms against 0.219 ms -- the ticket and the look-back chain do go away, but a
    // one-workgroup launch between the two halves costs most of what they cost; kept for A/B runs and as the tests' second witness.
    bool split = false;
    if (const char *f = getenv("STA_DEPTH_FORM")) split = lens != nullptr && f[0] == 's';
    if (!split && !status_zeroed) hipMemsetAsync(status, 0, sta_depth_fused_status_bytes(ncols), s);
    DepthFusedArgs a;
    a.lens = lens;
    a.status = (unsigned long long *)status;
    a.ticket = (unsigned int *)((char *)status + (size_t)n_tiles * 16);
    a.out = out; a.capacity = capacity; a.counts = counts; a.ctr = ctr;
    a.lbuf = lbuf; a.per_wave = ((lbuf + 16 + 15) & ~15u) + 16; a.n_tiles = (uint32_t)n_tiles;
    a.has_clip = p.remove_overlaps ? 1 : 0;
    { const char *dg = getenv("STA_DEPTH_DIAG"); a.diag = dg ? atoi(dg) : 0; }
    DepthDevPar d{ p.min_qual, p.skip_del, p.all_pos };
    // STA_DEPTH_TICKET=0: tile = blockIdx.x (workgroups are dispatched in index order and never preempted, so a tile's
    // predecessors are finished or running); the default ticket does not rely on that
    static const bool use_ticket = !(getenv("STA_DEPTH_TICKET") && atoi(getenv("STA_DEPTH_TICKET")) == 0);
    if (!use_ticket) a.ticket = nullptr;
    const size_t marks = (size_t)4 * 2 * (DF_SPAN + 4) * sizeof(int), text = (size_t)4 * a.per_wave;
    if (split) {
        a.ticket = nullptr;
        hipLaunchKernelGGL(k_depth_fused<1>, dim3((unsigned)n_tiles), dim3(256), marks, s, w, d, a);
        hipLaunchKernelGGL(k_depth_wave_scan, dim3(1), dim3(1024), 0, s, a.status, n_tiles, (unsigned long long)capacity, ctr);
        if (a.diag != 1) hipLaunchKernelGGL(k_depth_fused<2>, dim3((unsigned)n_tiles), dim3(256), text, s, w, d, a);
        return;
    }
    hipLaunchKernelGGL(k_depth_fused<0>, dim3((unsigned)n_tiles), dim3(256), marks > text ? marks : text, s, w, d, a);
}
