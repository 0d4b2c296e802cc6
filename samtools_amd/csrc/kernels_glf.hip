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

// The per-column counters.  S == 0: one byte per key, tile[key * 64 + lane] -- 37.5 KB of LDS per wave, i.e. FOUR waves per CU, one per SIMD: the
// kernel ran at 1.4 % of the HBM roof with nothing to hide its latencies behind (6.6 ms for 4 M columns at 30x).  A column holds few DISTINCT
// keys (a handful of quality values x two strands x the reference base and an error or two): S > 0 keeps S slots (key << 8 | count) per column,
// open addressing with linear probing, slots[slot * 64 + lane] -- 16 KB per wave at S = 64, ten waves per CU.  A column that needs more than
// S slots marks its group of 64 columns in `redo` and a second launch of the S == 0 form computes those groups (and leaves at once elsewhere).
template <int S> struct GlfCnt {
    uint32_t *t; int lane; bool over;
    __device__ __forceinline__ void clear() { for (int i = lane; i < (S ? S * 64 : GLF_KEYS * 64 / 4); i += 64) t[i] = S ? 0xffffffffu : 0u; over = false; }
    __device__ __forceinline__ void add(int key)
    {
        if (S == 0) { reinterpret_cast<uint8_t *>(t)[key * 64 + lane]++; return; }
        int h = key & (S - 1);
        for (int probe = 0; probe < S; ++probe, h = (h + 1) & (S - 1)) {
            const uint32_t w = t[h * 64 + lane];
            if (w == 0xffffffffu) { t[h * 64 + lane] = (uint32_t)key << 8 | 1u; return; }
            if ((int)(w >> 8) == key) { t[h * 64 + lane] = w + 1u; return; }
        }
        over = true;
    }
    // the count of a key that was added (S == 0: and the counter left clean for the next file)
    __device__ __forceinline__ int take(int key)
    {
        if (S == 0) { uint8_t *b = reinterpret_cast<uint8_t *>(t) + key * 64 + lane; const int r = *b; *b = 0; return r; }
        int h = key & (S - 1);
        for (int probe = 0; probe < S; ++probe, h = (h + 1) & (S - 1)) {
            const uint32_t w = t[h * 64 + lane];
            if ((int)(w >> 8) == key) return (int)(w & 255u);
            if (w == 0xffffffffu) break;
        }
        return 0;
    }
};

// Which form a window starts with is decided on the device from a sample of its qualities (k_glf_tier: the number of distinct values that
// can reach the counters, and the depth the host knows): tier[0] = the level, 16 | 32 | 64 slots | the byte tile.  The four forms are launched one behind the other; a form
// below the chosen one leaves at once, the chosen one takes every group, a form above it the groups the one before it marked.
template <int S>
__global__ void __launch_bounds__(64) k_glf_cols(StaWinDev W, GlfPar P, GlfCol *out, uint8_t *redo, const uint32_t *tier)
{
    extern __shared__ uint8_t tile[];             // S == 0: [GLF_KEYS][64] byte counters; S > 0: [S][64] slot words
    if (tier) {
        const int start = (int)*tier;             // level of the first form: 0 = 16 slots, 1 = 32, 2 = 64, 3 = the byte tile for every column
        const int mine = S == 0 ? 3 : S == 64 ? 2 : S == 32 ? 1 : 0;
        if (mine < start) return;
        if (mine > start && !redo[blockIdx.x]) return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    if (c0 >= ncols) return;
    GlfCnt<S> cn; cn.t = reinterpret_cast<uint32_t *>(tile); cn.lane = lane;
    cn.clear();                                    // one wave per block: no barrier needed
    bool over_any = false;
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
                        cn.add(((q - 4) * 2 + rev) * 5 + b);
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
                    const int reps = cn.take((qi * 2 + s) * 5 + b);
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
        if (S) { over_any |= cn.over; if (f + 1 < W.nfiles) cn.clear(); }
    }
    if (S && redo) { const bool any = __ballot(over_any) != 0; if (lane == 0) redo[blockIdx.x] = any ? 1 : 0; }      // (S == 0 holds every key)
}

// distinct quality values (as they reach the counters: below min_baseQ dropped, clamped to [4, 63]) among the first 64 KiB of the first file's
// qualities; with the window's mean depth: how many distinct (quality, strand, base) keys a column can be expected to hold at most
__global__ void __launch_bounds__(1024) k_glf_tier(StaWinDev W, int min_baseQ, float depth, uint32_t *tier)
{
    __shared__ unsigned long long mask;
    if (threadIdx.x == 0) mask = 0;
    __syncthreads();
    unsigned long long m = 0;
    if (W.nfiles > 0 && W.files[0].n) {
        const StaReadsDev &R = W.files[0];
        const uint64_t nb = R.n_bases_total < 65536 ? (uint64_t)R.n_bases_total : 65536ull;
        for (uint64_t i = threadIdx.x; i < nb; i += blockDim.x) {
            int q = R.qual[i];
            if (q < min_baseQ) continue;
            q = q > 63 ? 63 : q < 4 ? 4 : q;
            m |= 1ull << q;
        }
    }
    for (int d = 32; d; d >>= 1) m |= __shfl_xor(m, d);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(&mask, m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nq = __popcll(mask);
        const float nmax = depth + 3.f * sqrtf(depth) + 2.f;          // entries of a column (Poisson tail)
        float keys = 2.5f * (float)nq + 2.f;                          // two strands, the column's base and an error or two
        if (keys > nmax) keys = nmax;
        *tier = keys <= 12.f ? 0u : keys <= 24.f ? 1u : keys <= 48.f ? 2u : 3u;
    }
}

size_t sta_glf_redo_bytes(const StaWinDev &w) { const int64_t ncols = (int64_t)w.col_end - w.col_beg; return 64 + (ncols > 0 ? (size_t)((ncols + 63) / 64) : 0); }

// redo: sta_glf_redo_bytes() -- the chosen form (one word, 64 bytes reserved) + one byte per group of 64 columns.
// STA_GLF_SLOTS = 0 (the byte tile for every column, as before round 6) | 16 | 32 | 64: that form first whatever the sample says
void sta_launch_glf_cols(hipStream_t s, const StaWinDev &w, int min_baseQ, int capQ, const char *ref, int64_t ref_len,
                         const double *fk, const double *beta, const double *lhet, void *out, uint8_t *redo, double mean_depth)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    GlfPar p{ min_baseQ, capQ, ref, ref_len, fk, beta, lhet };
    static const int forced = [] { const char *e = getenv("STA_GLF_SLOTS"); return e && *e ? atoi(e) : -1; }();
    const dim3 grid((unsigned)((ncols + 63) / 64));
    if (forced == 0 || !redo) { hipLaunchKernelGGL(k_glf_cols<0>, grid, dim3(64), GLF_KEYS * 64, s, w, p, (GlfCol *)out, (uint8_t *)nullptr, (const uint32_t *)nullptr); return; }
    uint32_t *tier = reinterpret_cast<uint32_t *>(redo);
    uint8_t *flags = redo + 64;
    if (forced == 16 || forced == 32 || forced == 64) { const uint32_t v = forced == 16 ? 0u : forced == 32 ? 1u : 2u; hipMemcpyAsync(tier, &v, 4, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
    else hipLaunchKernelGGL(k_glf_tier, dim3(1), dim3(1024), 0, s, w, min_baseQ, (float)mean_depth, tier);
    hipLaunchKernelGGL(k_glf_cols<16>, grid, dim3(64), 16 * 64 * 4, s, w, p, (GlfCol *)out, flags, (const uint32_t *)tier);
    hipLaunchKernelGGL(k_glf_cols<32>, grid, dim3(64), 32 * 64 * 4, s, w, p, (GlfCol *)out, flags, (const uint32_t *)tier);
    hipLaunchKernelGGL(k_glf_cols<64>, grid, dim3(64), 64 * 64 * 4, s, w, p, (GlfCol *)out, flags, (const uint32_t *)tier);
    hipLaunchKernelGGL(k_glf_cols<0>, grid, dim3(64), GLF_KEYS * 64, s, w, p, (GlfCol *)out, flags, (const uint32_t *)tier);
}
