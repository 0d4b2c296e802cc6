// kernels_glf.hip -- per-column genotype-likelihood packer.
//
// Replaces bcf_call_glfgen (bam2bcf.c:65-123) + HTSlib errmod_cal, which tview calls once per column on the iterator's
// entries: one wave owns 64 columns, one lane per column (reads walked uniformly as in the pileup kernels).  A lane
//   1. filters / caps its entries exactly like the reference loop (deletions, ref skips, base quality, mapping quality cap
//      60 with 255 -> 20, clamp to [4,63]), packs them as q<<5 | strand<<4 | base into its column of an LDS tile and adds the
//      qualities into qsum[] in pileup order (float adds: the order is part of the result);
//   2. sorts its <=255 packed values (insertion sort in LDS; the values are 16-bit keys, so any sort gives HTSlib's order);
//   3. runs errmod_cal's accumulation from the highest quality down (fp64, dependent on running counts) and the 5x5
//      genotype table, from coefficient tables computed once on the host (32 MB of beta[q][n][k] in HBM, read sparsely).
// Byte/integer work plus a short fp64 recurrence per column: HBM/latency bound, no MFMA.
#include "dev_util.h"

struct GlfPar { int32_t min_baseQ, capQ; const char *ref; int64_t ref_len; const double *fk, *beta, *lhet; };
struct GlfCol { int32_t n_plp, n, flags; float qsum[4]; float p[25]; };      // = sta_glf_col

#define GLF_MAXB 255

__device__ __forceinline__ int nt16_to_2bit(int c) { return c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 4; }   // seq_nt16_int

__global__ void __launch_bounds__(64) k_glf_cols(StaWinDev W, GlfPar P, GlfCol *out)
{
    extern __shared__ uint16_t tile[];            // [GLF_MAXB][64]: lane's values at tile[i * 64 + lane]
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    if (c0 >= ncols) return;
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
        float qsum[4] = { 0.f, 0.f, 0.f, 0.f };
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
                    if (cnt < GLF_MAXB) tile[cnt * 64 + lane] = (uint16_t)(q << 5 | ((info & RI_REV) ? 1 : 0) << 4 | b);
                    cnt++;
                    if (b < 4) qsum[b] += (float)q;
                }
            }
        }
        if (!active) continue;
        GlfCol o;
        o.n_plp = n_plp; o.n = cnt; o.flags = cnt > GLF_MAXB ? 1 : 0;
        for (int i = 0; i < 4; ++i) o.qsum[i] = qsum[i];
        for (int i = 0; i < 25; ++i) o.p[i] = 0.f;
        const int n = cnt > GLF_MAXB ? GLF_MAXB : cnt;
        if (n > 0) {
            // ascending insertion sort of the lane's values
            for (int i = 1; i < n; ++i) {
                uint16_t v = tile[i * 64 + lane];
                int j = i - 1;
                while (j >= 0 && tile[j * 64 + lane] > v) { tile[(j + 1) * 64 + lane] = tile[j * 64 + lane]; --j; }
                tile[(j + 1) * 64 + lane] = v;
            }
            // errmod_cal: running sums from the highest quality down
            double fsum[5] = { 0, 0, 0, 0, 0 }, bsum[5] = { 0, 0, 0, 0, 0 };
            int c[5] = { 0, 0, 0, 0, 0 }, wf[5] = { 0, 0, 0, 0, 0 }, wr[5] = { 0, 0, 0, 0, 0 };
            for (int j = n - 1; j >= 0; --j) {
                const int v = tile[j * 64 + lane];
                int qual = v >> 5; if (qual < 4) qual = 4; if (qual > 63) qual = 63;
                const int base = v & 0xf, rev = (v >> 4) & 1;       // base is 0..4 here
                const int wv = rev ? wr[base] : wf[base];
                const double fk = P.fk[wv];
                fsum[base] += fk;
                bsum[base] += fk * P.beta[(size_t)qual << 16 | (size_t)n << 8 | (size_t)c[base]];
                ++c[base];
                if (rev) ++wr[base]; else ++wf[base];
            }
            // genotype table (float accumulators exactly as in the reference: float += double)
            for (int j = 0; j < 5; ++j) {
                float tmp1 = 0.f; int tmp2 = 0;
                for (int k = 0; k < 5; ++k) { if (k == j) continue; tmp1 = (float)((double)tmp1 + bsum[k]); tmp2 += c[k]; }
                if (tmp2) o.p[j * 5 + j] = tmp1;
                for (int k = j + 1; k < 5; ++k) {
                    const int cjk = c[j] + c[k];
                    tmp1 = 0.f; tmp2 = 0;
                    for (int i = 0; i < 5; ++i) { if (i == j || i == k) continue; tmp1 = (float)((double)tmp1 + bsum[i]); tmp2 += c[i]; }
                    const double het = -4.343 * P.lhet[cjk << 8 | c[k]];
                    const float v = tmp2 ? (float)(het + (double)tmp1) : (float)het;
                    o.p[j * 5 + k] = v; o.p[k * 5 + j] = v;
                }
                for (int k = 0; k < 5; ++k) if (o.p[j * 5 + k] < 0.0f) o.p[j * 5 + k] = 0.0f;
            }
        }
        out[(size_t)(c0 + lane) * (size_t)W.nfiles + (size_t)f] = o;
    }
}

void sta_launch_glf_cols(hipStream_t s, const StaWinDev &w, int min_baseQ, int capQ, const char *ref, int64_t ref_len,
                         const double *fk, const double *beta, const double *lhet, void *out)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    GlfPar p{ min_baseQ, capQ, ref, ref_len, fk, beta, lhet };
    hipLaunchKernelGGL(k_glf_cols, dim3((unsigned)((ncols + 63) / 64)), dim3(64), GLF_MAXB * 64 * sizeof(uint16_t), s, w, p, (GlfCol *)out);
}
