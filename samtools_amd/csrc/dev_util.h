// dev_util.h -- small device helpers shared by the kernels (gfx950, wave64).
#pragma once
#include "sta_dev.h"
#include "plp_entry.h"

#define WAVE 64

// BAQ window geometry of realn.c: reference window [xb, xe) and the band width that
// probaln_glocal ends up using.  Shared by k_prep_reads (scratch sizing) and k_baq.
struct BaqGeo { long long xb, xe; int bw, l_ref; bool ok; };
__device__ __forceinline__ BaqGeo baq_geometry(const uint32_t *cigar, int n_cigar, long long rpos, int lq,
                                                const char *ref, long long ref_len)
{
    BaqGeo g; g.ok = false; g.bw = 7; g.l_ref = 0; g.xb = g.xe = -1;
    long long x = rpos, xb = -1, xe = -1; int y = 0, yb = -1, ye = -1;
    for (int k = 0; k < n_cigar; ++k) {
        int op = cigar[k] & 0xf, l = (int)(cigar[k] >> 4);
        if (cg_is_mop(op)) {
            if (yb < 0) yb = y;
            if (xb < 0) xb = x;
            ye = y + l; xe = x + l;
            x += l; y += l;
        } else if (op == CG_S || op == CG_I) y += l;
        else if (op == CG_D) x += l;
    }
    if (xb < 0) return g;
    int bw = 7;
    long long d = (xe - xb) - (ye - yb); if (d < 0) d = -d;
    if (d > bw) bw = (int)d + 3;
    xb -= yb + bw / 2; if (xb < 0) xb = 0;
    xe += lq - ye + bw / 2;
    if (xe - xb - lq > bw) { xb += (xe - xb - lq - bw) / 2; xe -= (xe - xb - lq - bw) / 2; }
    (void)ref;                      // the staged contig has an explicit length and no NUL bytes
    if (xe > ref_len) xe = ref_len > xb ? ref_len : xb;
    int l_ref = (int)(xe - xb);
    g.xb = xb; g.xe = xe; g.l_ref = l_ref;
    // probaln_glocal: bw = min(max(l_ref, l_query), bw); bw = max(bw, |l_ref - l_query|)
    int b2 = l_ref > lq ? l_ref : lq;
    if (b2 > bw) b2 = bw;
    int dd = l_ref - lq; if (dd < 0) dd = -dd;
    if (b2 < dd) b2 = dd;
    g.bw = b2;
    g.ok = l_ref > 0 && lq > 0;
    return g;
}

// wave-cooperative search on a non-decreasing int32 array: first index in [0,n) with a[i] > key
// (a "64-ary" search: every step the 64 lanes probe 64 evenly spaced points)
__device__ __forceinline__ int64_t wave_upper_bound(const int32_t *a, int64_t n, int32_t key)
{
    int lane = threadIdx.x & (WAVE - 1);
    int64_t lo = 0, hi = n;          // answer in [lo, hi]
    while (hi - lo > 0) {
        int64_t span = hi - lo;
        int64_t step = (span + WAVE - 1) / WAVE;
        int64_t idx = lo + (int64_t)lane * step;
        bool gt = (idx < hi) ? (a[idx] > key) : true;
        unsigned long long m = __ballot(gt);
        int first = m ? __ffsll((long long)m) - 1 : WAVE;   // first lane whose probe is > key
        // answer lies in (probe[first-1], probe[first]]
        int64_t nlo = first == 0 ? lo : lo + (int64_t)(first - 1) * step + 1;
        int64_t nhi = lo + (int64_t)first * step;
        if (nhi > hi) nhi = hi;
        if (step == 1) { lo = nhi; hi = nhi; break; }
        lo = nlo; hi = nhi;
    }
    return lo;
}


// stateless HTSlib resolve_cigar2 for one (read, column) -- SURVEY.md A.2 (shared by the entry and coverage kernels)
__device__ __forceinline__ void plp_resolve(const uint32_t *cig, int n, int rpos, int p, int &qpos, int &indel, int &k_out,
                                            bool &is_del, bool &is_refskip)
{
    int x = rpos, y = 0, k = 0, op = 0, l = 0;
    for (k = 0; k < n; ++k) {
        uint32_t c = cig[k];
        op = c & 0xf; l = (int)(c >> 4);
        if (cg_is_refop(op)) {
            if (p < x + l) break;
            if (cg_is_mop(op)) y += l;
            x += l;
        } else if (cg_is_qop(op)) y += l;
    }
    k_out = k; indel = 0; is_del = false; is_refskip = false;
    if (x + l - 1 == p && k + 1 < n) {
        int op2 = cig[k + 1] & 0xf, l2 = (int)(cig[k + 1] >> 4);
        if (op2 == CG_D && op != CG_D) {
            indel = -l2;
            for (int j = k + 2; j < n; ++j) { if ((cig[j] & 0xf) == CG_D) indel -= (int)(cig[j] >> 4); else break; }
        } else if (op2 == CG_I) {
            indel = l2;
            for (int j = k + 2; j < n; ++j) {
                int o = cig[j] & 0xf;
                if (o == CG_I) indel += (int)(cig[j] >> 4);
                else if (o != CG_P) break;
            }
        } else if (op2 == CG_P && k + 2 < n) {
            int l3 = 0;
            for (int j = k + 2; j < n; ++j) {
                int o = cig[j] & 0xf;
                if (o == CG_I) l3 += (int)(cig[j] >> 4);
                else if (cg_is_refop(o)) break;
            }
            if (l3 > 0) indel = l3;
        }
    }
    if (cg_is_mop(op)) qpos = y + (p - x);
    else { is_del = true; qpos = y; is_refskip = (op == CG_N); }
}


// Inclusive prefix sum over the 64 lanes of a wave, every lane taking part (full EXEC): six v_add_u32 with DPP operands (row_shr 1 / 2 / 4 / 8
// inside the rows of sixteen, then row_bcast:15 and row_bcast:31 across them) -- no LDS crossbar round trips as with six __shfl_up.
// Lanes without a source read `old` = 0 (bound_ctrl off).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return (uint32_t)x;
}

// Block-wide reduction of N per-thread values followed by ONE global atomic per value and block
// (sum for k < NSUM, max for the rest).  Keeps contended device atomics off the per-wave path: a
// counter word sustains only ~90 atomics/us, so one atomic per wave (65k+ waves) costs milliseconds.
template <int N, int NSUM>
__device__ __forceinline__ void block_reduce_atomic(unsigned long long (&v)[N], unsigned long long *const (&dst)[N])
{
    __shared__ unsigned long long red_[16][N];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        unsigned long long x = v[k];
        for (int o = 32; o; o >>= 1) {
            unsigned long long y = __shfl_down(x, o);
            x = k < NSUM ? x + y : (y > x ? y : x);
        }
        if (lane == 0) red_[wid][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int k = threadIdx.x;
        unsigned long long x = red_[0][k];
        for (int w = 1; w < nw; ++w) { unsigned long long y = red_[w][k]; x = k < NSUM ? x + y : (y > x ? y : x); }
        if (x) { if (k < NSUM) atomicAdd(dst[k], x); else atomicMax(dst[k], x); }
    }
    __syncthreads();      // the staging array is reused by the next call in the same kernel (k_cov_cols calls this per file)
}
