// kernels_cons.hip -- the consensus path on the device (gfx950): one lane per read for the read-shaped steps, one lane per
// (position, nth) column for the callers.  The arithmetic lives in cons_core.h / cons_window.h (shared with the CPU harness of
// the tests); this file is the launch geometry.  Built with -ffp-contract=off: the Bayesian caller's fp64 sums and the
// fast_log2 polynomial must round exactly as the reference's unfused x86-64 code does.
//
// Bound: integer / byte work with data-dependent control flow, HBM traffic = entry words (4 B, + 4 B in the Bayesian mode
// with mapping qualities) written once by the read walk and read once by the column caller, plus the staged reads.  No MFMA.
#include "sta_dev.h"
#include "cons_window.h"

using namespace cons;

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

__device__ __forceinline__ unsigned long long wave_sum_ull(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;       // lane 0 holds the sum
}

__global__ void __launch_bounds__(256) k_cons_read_a(Win w, Par o, const Tables *t, int walk_all)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int code = 0;
    if (r < w.n_reads) code = step_read_a(w, o, *t, r, [](uint32_t *p, uint32_t v) { atomicMax(p, v); }, false, walk_all != 0);
    // one global atomic per workgroup, not per read: a single counter word takes ~90 atomics per microsecond
    __shared__ unsigned int s_kept, s_bad;
    if (threadIdx.x == 0) { s_kept = 0; s_bad = 0; }
    __syncthreads();
    const unsigned long long kept = __ballot(code > 0), bad = __ballot(code < 0);
    if ((threadIdx.x & 63) == 0) {
        if (kept) atomicAdd(&s_kept, (unsigned int)__popcll(kept));
        if (bad) atomicAdd(&s_bad, (unsigned int)__popcll(bad));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_kept) atomicAdd(&w.counters[0], (unsigned long long)s_kept);
        if (s_bad) atomicAdd(&w.counters[1], (unsigned long long)s_bad);
    }
}

// nm_init for the Bayesian mode: one wave per 64 consecutive reads.  A lane working through its own read touches its
// qualities / bases / nm words byte by byte, several passes; straight from HBM that is 64 different cache lines per load
// instruction.  The wave therefore copies the (contiguous) pool slice of its 64 reads into LDS with coalesced loads, every
// lane runs read_prepare() on LDS, and the nm words (and rewritten qualities) go back with coalesced stores.
#define PREP_CAP 10240                  // pool bytes of 64 reads the LDS slice holds (64 x 160)
__global__ void __launch_bounds__(64) k_cons_prepare(Win w, Par o, const Tables *t)
{
    extern __shared__ unsigned char lds[];
    uint8_t *lq = lds;                                   // PREP_CAP
    uint8_t *ls = lds + PREP_CAP;                        // PREP_CAP / 2
    int32_t *ln = (int32_t *)(lds + PREP_CAP + PREP_CAP / 2);      // PREP_CAP words
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = r0 + 64 < w.n_reads ? r0 + 64 : w.n_reads;
    const int64_t r = r0 + lane;
    const int64_t first = (int64_t)w.base_off8[r0] * 8;
    const int64_t end = (int64_t)w.base_off8[r1 - 1] * 8 + ((w.l_qseq[r1 - 1] + 7) & ~7);
    const bool mine = r < r1 && (w.r_keep[r] & 1u);
    const char *md = nullptr; int md_len = 0;
    if (mine) md_of(w, r, md, md_len);
    const int64_t B = end - first;
    bool staged = B > 0 && B <= PREP_CAP;
    if (staged) {
        // every read of the slice must lie inside it, in order
        const int64_t off = r < r1 ? (int64_t)w.base_off8[r] * 8 - first : 0;
        const bool ok = r >= r1 || (off >= 0 && off + ((w.l_qseq[r] + 7) & ~7) <= B);
        staged = __all(ok);
    }
    if (!staged) {
        if (mine) { ReadView v = view_of(w, r, false); read_prepare(o, *t, v, w.qual + (size_t)w.base_off8[r] * 8, md, md_len, w.nm + (size_t)w.base_off8[r] * 8); }
        return;
    }
    const uint32_t *gq = (const uint32_t *)(w.qual_in + first);          // pool offsets are multiples of 8
    const uint32_t *gs = (const uint32_t *)(w.seq + first / 2);
    for (int64_t i = lane; i < B / 4; i += 64) ((uint32_t *)lq)[i] = gq[i];
    for (int64_t i = lane; i < B / 8; i += 64) ((uint32_t *)ls)[i] = gs[i];
    __syncthreads();
    if (mine) {
        const int64_t off = (int64_t)w.base_off8[r] * 8 - first;
        ReadView v = view_of(w, r, false);
        v.seq = ls + off / 2; v.qual = lq + off;
        read_prepare(o, *t, v, lq + off, md, md_len, ln + off);
    }
    __syncthreads();
    int32_t *gn = w.nm + first;
    for (int64_t i = lane; i < B; i += 64) gn[i] = ln[i];
    if (o.homopoly_on) { uint32_t *gw = (uint32_t *)(w.qual + first); for (int64_t i = lane; i < B / 4; i += 64) gw[i] = ((uint32_t *)lq)[i]; }
}

// The default configuration (no homopolymer fixing, not the 1.16 mode) needs no sequential walk: prepare_granule() gives the nm
// words of 8 bases from their neighbourhood.  One thread per staged base: k_cons_granules first notes, for every 8-base granule of
// the pool (reads start on granule boundaries), which read it belongs to; k_cons_prepare_base then runs over the pool, one
// thread per granule, with nothing sequential between bases.
__global__ void __launch_bounds__(256) k_cons_granules(Win w, int32_t *gran2read)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= w.n_reads || !(w.r_keep[r] & 1u)) return;
    const uint32_t g0 = w.base_off8[r], g1 = g0 + (uint32_t)((w.l_qseq[r] + 7) >> 3);
    for (uint32_t g = g0; g < g1; ++g) gran2read[g] = (int32_t)r;
}
__global__ void __launch_bounds__(256) k_cons_prepare_base(Win w, Par o, const int32_t *gran2read, int64_t n_gran)
{
    // one thread per granule of 8 bases: the chain gran2read -> read record -> qualities is several dependent memory
    // latencies deep, and a thread that stops after one base spends its whole life waiting on it
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_gran) return;
    const int32_t r = gran2read[g];
    if (r < 0) return;
    const int i0 = (int)((g - (int64_t)w.base_off8[r]) * 8);
    const ReadView v = view_of(w, r, false);
    int32_t out[8];
    prepare_granule(o, v, i0, out);
    int4 *nm = (int4 *)(w.nm + g * 8);
    nm[0] = make_int4(out[0], out[1], out[2], out[3]); nm[1] = make_int4(out[4], out[5], out[6], out[7]);
}
// ... followed by the soft-clip / MD costs, one lane per read that carries an MD tag
__global__ void __launch_bounds__(256) k_cons_prepare_md(Win w, Par o)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= w.n_reads || !(w.r_keep[r] & 1u)) return;
    const char *md; int md_len;
    md_of(w, r, md, md_len);
    const ReadView v = view_of(w, r, false);
    read_prepare_md(o, v, md, md_len, w.nm + (size_t)w.base_off8[r] * 8);
}

// lengths for the column-index scan: position i of the window owns 1 + ins[i + 1] columns (ins[0] is the look-back position)
__global__ void __launch_bounds__(256) k_cons_collen(const uint32_t *ins, uint32_t *len, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) len[i] = 1u + ins[i + 1];
}

__global__ void __launch_bounds__(256) k_cons_read_b(Win w)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t alive = 0;
    const bool walk = r < w.n_reads ? step_read_b(w, r, alive) : false;
    // reads that need the cursor walk are appended to clist (any order): slots are handed out per workgroup
    __shared__ unsigned int s_n[4];
    __shared__ unsigned long long s_base, s_alive;
    const unsigned long long m = __ballot(walk);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_alive = 0;
    if (lane == 0) s_n[wid] = (unsigned int)__popcll(m);
    __syncthreads();
    const unsigned long long tot = wave_sum_ull(alive);
    if (lane == 0 && tot) atomicAdd(&s_alive, tot);
    if (threadIdx.x == 0) { const unsigned int all = s_n[0] + s_n[1] + s_n[2] + s_n[3]; s_base = all ? atomicAdd(&w.counters[2], (unsigned long long)all) : 0ull; }
    __syncthreads();
    if (walk) {
        unsigned long long at = s_base;
        for (int k = 0; k < wid; ++k) at += s_n[k];
        w.clist[at + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)r;
    }
    if (threadIdx.x == 0 && s_alive) atomicAdd(&w.counters[3], s_alive);
}

__global__ void __launch_bounds__(256) k_cons_colpos(Win w, int64_t W)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W) step_colpos(w, i);
}

__global__ void __launch_bounds__(256) k_cons_walk(Win w, Par o, int64_t n_list, int so_words)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_list) step_walk(w, o, w.clist[k], so_words != 0);
}

// The Bayesian caller looks up eleven table entries per read and column (q2p, mqual_pow_1m, nine log-probabilities): the
// parameter set(s) and the two small tables are copied into LDS once per workgroup (10 KB, 18 KB in the mixed mode).
template <int KIND> __global__ void __launch_bounds__(256) k_cons_col(Win w, Par o, const Tables *t, int64_t n_cols)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t c0 = (int32_t)(c & ~63ll), c1 = (int32_t)(c0 + 63 < n_cols ? c0 + 63 : n_cols - 1);      // this wave's columns
    if (KIND == 0) {
        if (c < n_cols) step_col<0>(w, o, *t, t->recall, t->recall, t->q2p, t->mqual_pow_1m, c, c0, c1);
        return;
    }
    __shared__ Probs s_cp1;
    __shared__ Probs s_cp2[KIND == 2 ? 1 : 0 + (KIND == 2 ? 0 : 1)];      // one element either way; only filled in the mixed mode
    __shared__ double s_q2p[101], s_mq[256];
    {
        const Probs &g1 = first_probs(o, *t);
        const double *src = (const double *)&g1; double *dst = (double *)&s_cp1;
        for (int i = threadIdx.x; i < (int)(sizeof(Probs) / 8); i += 256) dst[i] = src[i];
        if (KIND == 2) { src = (const double *)&t->recall; dst = (double *)&s_cp2[0]; for (int i = threadIdx.x; i < (int)(sizeof(Probs) / 8); i += 256) dst[i] = src[i]; }
        if (threadIdx.x < 101) s_q2p[threadIdx.x] = t->q2p[threadIdx.x];
        s_mq[threadIdx.x] = t->mqual_pow_1m[threadIdx.x];
    }
    __syncthreads();
    if (c < n_cols) step_col<KIND>(w, o, *t, s_cp1, s_cp2[0], s_q2p, s_mq, c, c0, c1);
}

__global__ void __launch_bounds__(256) k_cons_text(Win w, Par o, int64_t n_cols)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t c0 = (int32_t)(c & ~63ll), c1 = (int32_t)(c0 + 63 < n_cols ? c0 + 63 : n_cols - 1);
    if (c < n_cols) step_text(w, o, c, c0, c1);
}


void sta_launch_cons_read_a(hipStream_t s, const Win &w, const Par &o, const Tables *t, bool walk_all)
{
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_read_a, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w, o, t, walk_all ? 1 : 0);
}
void sta_launch_cons_prepare(hipStream_t s, const Win &w, const Par &o, const Tables *t, int32_t *gran2read, int64_t n_bases)
{
    if (w.n_reads > 0 && prepare_is_per_base(o)) {
        // (a window whose records hold no base -- SEQ "*" throughout: five such reads in a row under STA_WINDOW_READS=5 -- has no granules: a grid
        //  of 0 workgroups is hipErrorInvalidConfiguration and the whole run failed; found on the device by hunt5 / hunt6 in round 6)
        if (n_bases > 0) {
            hipMemsetAsync(gran2read, 0xff, (size_t)((n_bases + 7) / 8) * 4, s);
            hipLaunchKernelGGL(k_cons_granules, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w, gran2read);
            hipLaunchKernelGGL(k_cons_prepare_base, dim3(blocks_for((n_bases + 7) / 8)), dim3(256), 0, s, w, o, gran2read, (n_bases + 7) / 8);
        }
        hipLaunchKernelGGL(k_cons_prepare_md, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w, o);
        return;
    }
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_prepare, dim3((unsigned)((w.n_reads + 63) / 64)), dim3(64), PREP_CAP * 5 + PREP_CAP / 2, s, w, o, t);
}
void sta_launch_cons_collen(hipStream_t s, const uint32_t *ins, uint32_t *len, int64_t W)
{
    if (W > 0) hipLaunchKernelGGL(k_cons_collen, dim3(blocks_for(W)), dim3(256), 0, s, ins, len, W);
}
void sta_launch_cons_read_b(hipStream_t s, const Win &w)
{
    if (w.n_reads > 0) hipLaunchKernelGGL(k_cons_read_b, dim3(blocks_for(w.n_reads)), dim3(256), 0, s, w);
}
void sta_launch_cons_colpos(hipStream_t s, const Win &w)
{
    const int64_t W = (int64_t)w.col_end - w.col_beg;
    if (W > 0) hipLaunchKernelGGL(k_cons_colpos, dim3(blocks_for(W)), dim3(256), 0, s, w, W);
}
void sta_launch_cons_walk(hipStream_t s, const Win &w, const Par &o, int64_t n_list, bool so_words)
{
    if (n_list > 0) hipLaunchKernelGGL(k_cons_walk, dim3(blocks_for(n_list)), dim3(256), 0, s, w, o, n_list, so_words ? 1 : 0);
}
void sta_launch_cons_col(hipStream_t s, const Win &w, const Par &o, const Tables *t, int64_t n_cols)
{
    if (n_cols <= 0) return;
    const int kind = col_kind(o);
    if (kind == 0) hipLaunchKernelGGL(k_cons_col<0>, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, o, t, n_cols);
    else if (kind == 1) hipLaunchKernelGGL(k_cons_col<1>, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, o, t, n_cols);
    else hipLaunchKernelGGL(k_cons_col<2>, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, o, t, n_cols);
}
void sta_launch_cons_text(hipStream_t s, const Win &w, const Par &o, int64_t n_cols)
{
    if (n_cols > 0) hipLaunchKernelGGL(k_cons_text, dim3(blocks_for(n_cols)), dim3(256), 0, s, w, o, n_cols);
}
