// kernels_glf.hip -- per-column genotype-likelihood packer.
//
// Replaces bcf_call_glfgen (bam2bcf.c:65-123) + HTSlib errmod_cal, which tview calls once per column on the iterator's
// entries: one wave owns 64 columns, one lane per column (reads walked uniformly as in the pileup kernels).  A lane
//   1. filters / caps its entries exactly like the reference loop (deletions, ref skips, base quality, mapping quality cap
//      60 with 255 -> 20, clamp to [4,63]) and adds the qualities into qsum[] in pileup order (float adds: the order is part
//      of the result);
//   2. instead of sorting the packed values q<<5 | strand<<4 | base as errmod_cal does, counts them: a value has only
//      60 x 2 x 5 possible keys, so the lane keeps one byte counter per key in its column of an LDS tile (600 x 64 B per wave)
//      and a 60-bit mask of the qualities seen per (strand, base) in registers;
//   3. runs errmod_cal's accumulation in the sorted order anyway -- the running sums of different bases are independent, so
//      per base the keys are visited from the highest quality down, reverse strand first, `count` times each (fp64, dependent
//      on running counts) -- then the 5x5 genotype table, from coefficient tables computed once on the host (32 MB of
//      beta[q][n][k] in HBM, read sparsely).  O(n + distinct keys) LDS traffic per column instead of the O(n^2) of a sort.
// Byte/integer work plus a short fp64 recurrence per column: HBM/latency bound, no MFMA.
#include "dev_util.h"

struct GlfPar { int32_t min_baseQ, capQ; const char *ref; int64_t ref_len; const double *fk, *beta, *lhet; };
struct GlfCol { int32_t n_plp, n, flags; float qsum[4]; float p[25]; };      // = sta_glf_col

#define GLF_MAXB 255

__device__ __forceinline__ int nt16_to_2bit(int c) { return c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 4; }   // seq_nt16_int

#define GLF_KEYS 600            // (q - 4) in 0..59, strand, base 0..4: key = ((q - 4) * 2 + strand) * 5 + base

__global__ void __launch_bounds__(64) k_glf_cols(StaWinDev W, GlfPar P, GlfCol *out)
{
    extern __shared__ uint8_t tile[];             // [GLF_KEYS][64] byte counters: lane's counter of a key at tile[key * 64 + lane]
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    if (c0 >= ncols) return;
    {
        uint32_t *tw = reinterpret_cast<uint32_t *>(tile);
        for (int i = lane; i < GLF_KEYS * 64 / 4; i += 64) tw[i] = 0u;     // one wave per block: no barrier needed
    }
    const int p0 = W.col_beg + (int)c0;
    const int p = p0 + lane;
    const bool active = p < W.col_end;
    const int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
    // reference base of the column -> 4-bit code (seq_nt16_table): 15 when unknown
    int ref4 = 15;
    if (active && P.ref) { int64_t a = W.origin + p; if (a >= 0 && a < P.ref_len) ref4 = nt16_from_char((unsigned char)P.ref[a]); }

    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int n_plp = 0, cnt = 0;
        float qs0 = 0.f, qs1 = 0.f, qs2 = 0.f, qs3 = 0.f;
        // qualities seen per (strand, base): bit (q - 4); forward strand mf*, reverse strand mr*
        uint64_t mf0 = 0, mf1 = 0, mf2 = 0, mf3 = 0, mf4 = 0, mr0 = 0, mr1 = 0, mr2 = 0, mr3 = 0, mr4 = 0;
        if (R.n) {
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
                    if (is_del || is_refskip) continue;                       // bam2bcf.c:88
                    const int lq = R.l_qseq[r];
                    const uint64_t boff = (uint64_t)R.base_off8[r] << 3;
                    int mapQ = (int)((info >> RI_MAPQ_SHIFT) & 0xff);
                    if (mapQ >= 255) mapQ = 20;                                // DEF_MAPQ
                    int q = qpos < lq ? (int)R.qual[boff + (uint64_t)qpos] : 0;
                    if (q < P.min_baseQ) continue;
                    if (q > 99) q = 99;
                    if (mapQ > P.capQ) mapQ = P.capQ;
                    if (q > mapQ) q = mapQ;
                    if (q > 63) q = 63;
                    if (q < 4) q = 4;
                    int b = 4;
                    if (qpos < lq) {
                        int c = (R.seq[(boff >> 1) + ((uint64_t)qpos >> 1)] >> ((~qpos & 1) << 2)) & 0xf;
                        b = nt16_to_2bit(c ? c : ref4);
                    }
                    const int rev = (info & RI_REV) ? 1 : 0;
                    if (cnt < GLF_MAXB) {
                        tile[(((q - 4) * 2 + rev) * 5 + b) * 64 + lane]++;
                        const uint64_t bit = 1ull << (q - 4);
                        const uint64_t fb = rev ? 0ull : bit, rb = rev ? bit : 0ull;
                        mf0 |= b == 0 ? fb : 0ull; mf1 |= b == 1 ? fb : 0ull; mf2 |= b == 2 ? fb : 0ull; mf3 |= b == 3 ? fb : 0ull; mf4 |= b == 4 ? fb : 0ull;
                        mr0 |= b == 0 ? rb : 0ull; mr1 |= b == 1 ? rb : 0ull; mr2 |= b == 2 ? rb : 0ull; mr3 |= b == 3 ? rb : 0ull; mr4 |= b == 4 ? rb : 0ull;
                    }
                    cnt++;
                    const float qf = (float)q;
                    qs0 += b == 0 ? qf : 0.f; qs1 += b == 1 ? qf : 0.f; qs2 += b == 2 ? qf : 0.f; qs3 += b == 3 ? qf : 0.f;
                }
            }
        }
        const int n = cnt > GLF_MAXB ? GLF_MAXB : cnt;
        GlfCol o;
        o.n_plp = n_plp; o.n = cnt; o.flags = cnt > GLF_MAXB ? 1 : 0;
        o.qsum[0] = qs0; o.qsum[1] = qs1; o.qsum[2] = qs2; o.qsum[3] = qs3;
        for (int i = 0; i < 25; ++i) o.p[i] = 0.f;
        double fsum[5] = { 0, 0, 0, 0, 0 }, bsum[5] = { 0, 0, 0, 0, 0 };
        int c[5] = { 0, 0, 0, 0, 0 };
        const uint64_t mf[5] = { mf0, mf1, mf2, mf3, mf4 }, mr[5] = { mr0, mr1, mr2, mr3, mr4 };
        // errmod_cal's running sums, per base, keys in descending order (quality, then reverse strand before forward)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            int wf = 0, wr = 0;
            uint64_t u = mf[b] | mr[b];
            while (u) {
                const int qi = 63 - __clzll((long long)u);
                u &= ~(1ull << qi);
                const int qual = qi + 4;
#pragma unroll
                for (int s = 1; s >= 0; --s) {
                    if (!(((s ? mr[b] : mf[b]) >> qi) & 1ull)) continue;
                    const int key = ((qi * 2 + s) * 5 + b) * 64 + lane;
                    const int reps = tile[key];
                    tile[key] = 0;                                   // leave the tile clean for the next file
                    for (int t = 0; t < reps; ++t) {
                        const int wv = s ? wr : wf;
                        const double fk = P.fk[wv];
                        fsum[b] += fk;
                        bsum[b] += fk * P.beta[(size_t)qual << 16 | (size_t)n << 8 | (size_t)c[b]];
                        ++c[b];
                        if (s) ++wr; else ++wf;
                    }
                }
            }
        }
        (void)fsum;
        if (n > 0) {
            // genotype table (float accumulators exactly as in the reference: float += double)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                float tmp1 = 0.f; int tmp2 = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) { if (k == j) continue; tmp1 = (float)((double)tmp1 + bsum[k]); tmp2 += c[k]; }
                if (tmp2) o.p[j * 5 + j] = tmp1;
#pragma unroll
                for (int k = j + 1; k < 5; ++k) {
                    const int cjk = c[j] + c[k];
                    tmp1 = 0.f; tmp2 = 0;
#pragma unroll
                    for (int i = 0; i < 5; ++i) { if (i == j || i == k) continue; tmp1 = (float)((double)tmp1 + bsum[i]); tmp2 += c[i]; }
                    const double het = -4.343 * P.lhet[cjk << 8 | c[k]];
                    const float v = tmp2 ? (float)(het + (double)tmp1) : (float)het;
                    o.p[j * 5 + k] = v; o.p[k * 5 + j] = v;
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) if (o.p[j * 5 + k] < 0.0f) o.p[j * 5 + k] = 0.0f;
            }
        }
        if (active) out[(size_t)(c0 + lane) * (size_t)W.nfiles + (size_t)f] = o;
    }
}

void sta_launch_glf_cols(hipStream_t s, const StaWinDev &w, int min_baseQ, int capQ, const char *ref, int64_t ref_len,
                         const double *fk, const double *beta, const double *lhet, void *out)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    GlfPar p{ min_baseQ, capQ, ref, ref_len, fk, beta, lhet };
    hipLaunchKernelGGL(k_glf_cols, dim3((unsigned)((ncols + 63) / 64)), dim3(64), GLF_KEYS * 64, s, w, p, (GlfCol *)out);
}
