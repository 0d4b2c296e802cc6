// dev_util.h -- small device helpers shared by the kernels (gfx950, wave64).
#pragma once
#include "sta_dev.h"

#define WAVE 64

#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16

enum { CG_M = 0, CG_I, CG_D, CG_N, CG_S, CG_H, CG_P, CG_EQ, CG_X, CG_B };

__device__ __forceinline__ bool cg_is_refop(int op) { return (0x18Du >> op) & 1; }   // M D N = X  -> bits 0,2,3,7,8
__device__ __forceinline__ bool cg_is_mop(int op) { return (0x181u >> op) & 1; }     // M = X
__device__ __forceinline__ bool cg_is_qop(int op) { return op == CG_I || op == CG_S; }

__device__ __forceinline__ int dec_digits_u32(uint32_t v)
{
    return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u)
             + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
__device__ __forceinline__ int dec_digits(unsigned long long v)
{
    int n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}

// merged, sorted, disjoint intervals: does [beg,end) overlap any?  (bedidx.c:159-197 semantics)
__device__ __forceinline__ bool bed_overlap_dev(const int64_t *bbeg, const int64_t *bend, int64_t n, int64_t beg, int64_t end)
{
    // first interval with bend > beg
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (bend[mid] > beg) hi = mid; else lo = mid + 1;
    }
    return lo < n && bbeg[lo] < end;
}

// seq_nt16_table restricted to what FASTA text can hold (hts.c)
__device__ __forceinline__ int nt16_from_char(unsigned char c)
{
    switch (c) {
    case '=': return 0;
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'M': case 'm': return 3;
    case 'G': case 'g': return 4;
    case 'R': case 'r': return 5;
    case 'S': case 's': return 6;
    case 'V': case 'v': return 7;
    case 'T': case 't': return 8;
    case 'W': case 'w': return 9;
    case 'Y': case 'y': return 10;
    case 'H': case 'h': return 11;
    case 'K': case 'k': return 12;
    case 'D': case 'd': return 13;
    case 'B': case 'b': return 14;
    case '0': return 1;
    case '1': return 2;
    case '2': return 4;
    case '3': return 8;
    default: return 15;
    }
}

__device__ __forceinline__ int seq_nib(const uint8_t *seq, uint64_t seq_byte0, int i)
{
    return (seq[seq_byte0 + (uint64_t)(i >> 1)] >> ((~i & 1) << 2)) & 0xf;
}

__device__ __forceinline__ char lower_c(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }
__device__ __forceinline__ char upper_c(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

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


// a read reached bam_plp_push and was not dropped by the -d cap (it moved the iterator's max_pos)
__device__ __forceinline__ bool read_advances_iterator(const StaReadsDev &R, int64_t j)
{
    uint32_t info = R.info[j];
    bool dropped = (info & RI_PUSHED) && !(info & RI_KEEP) && R.end[j] > R.pos[j];
    return (info & RI_PUSHED) && !dropped;
}

// Quality a deletion / ref-skip placeholder of read r shows at column p (bam_plcmd.c:676-679 reads qual[qpos] of the NEXT base).
// HTSlib resolves a mate pair when the second mate is pushed, and a column is handed out as soon as some read starting
// beyond it has been pushed -- so a column before the mate's start sees the resolved quality only if the mate itself is
// that first read.  Everything else about the overlap pass is order independent; this is the one place where it is not.
__device__ __forceinline__ int placeholder_qual(const StaReadsDev &R, int64_t r, int qpos, int lq, uint64_t boff, int p)
{
    if (qpos >= lq) return 0;
    int q = R.qual[boff + (uint64_t)qpos];
    if (!R.fix_y || R.fix_y[r] != qpos) return q;
    const int64_t mate = R.fix_mate[r];
    if (p >= R.pos[mate]) return q;
    // first read (file order) starting beyond p that advances the iterator
    int64_t lo = 0, hi = R.n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (R.pos[mid] > p) hi = mid; else lo = mid + 1; }
    int64_t j = lo;
    while (j < R.n && !read_advances_iterator(R, j)) ++j;
    return j == mate ? q : (int)R.fix_q[r];
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
