// kernels_common.hip -- read preparation, quality preparation and scans (gfx950).
//
// k_prep_reads      : mplp_func read-level filters (bam_plcmd.c:413-458) + reference span
//                     (bam_cigar2rlen) + overlap eligibility (HTSlib overlap_push conditions,
//                     SURVEY.md A.3) for every staged read; one thread per read, SoA loads coalesce.
// k_prep_reads_depth: fastdepth_core read filters (bam2depth.c:552-571) + qlen_used (:124-159).
// k_qual_prep       : -6 shift (bam_plcmd.c:428-433) and BQ:Z tag application (realn.c, A.4) as one
//                     elementwise pass over the quality pool, 16 B per lane.
// scans             : three-phase block scans (reduce / scan of block sums / rescan), wave64 shuffles.
#include "dev_util.h"
#include "baq_band7s.h"
#include <cstdlib>


// ------------------------------------------------------------------------------------------------
// Block-level scan step shared by the three-phase scans below and by the preparation kernels' in-kernel prefix maximum.
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

struct OpSumU64 { typedef unsigned long long T; static __device__ T id() { return 0; } static __device__ T f(T a, T b) { return a + b; } };
struct OpMaxI32 { typedef int T; static __device__ T id() { return INT32_MIN; } static __device__ T f(T a, T b) { return a > b ? a : b; } };
struct OpSumI32 { typedef int T; static __device__ T id() { return 0; } static __device__ T f(T a, T b) { return a + b; } };

template <class Op> __device__ typename Op::T block_scan_incl(typename Op::T v, typename Op::T *wave_tot /*LDS[5]*/, typename Op::T &block_total)
{
    typedef typename Op::T T;
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        T u = __shfl_up(v, o);
        if (lane >= o) v = Op::f(u, v);
    }
    if (lane == 63) wave_tot[wid] = v;
    __syncthreads();
    T pre = Op::id();
    for (int w = 0; w < wid; ++w) pre = Op::f(pre, wave_tot[w]);
    T tot = Op::id();
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) tot = Op::f(tot, wave_tot[w]);
    block_total = tot;
    __syncthreads();
    return Op::f(pre, v);
}


// R.maxend inside the preparation kernels (one launch instead of prep + three scan launches per file and window).  Every
// workgroup owns `chunk` consecutive reads: it writes the running maximum of its own reads, publishes the chunk's maximum, takes
// the maximum of every chunk in front of it (all of them read at once, 256 per round: a window has at most 2048 chunks) and raises
// the entries that maximum exceeds.  Chunks are numbered by a ticket, so a workgroup only ever waits for workgroups that started
// before it; neither the ticket nor the published words are cleared between launches -- the host passes the ticket's base and a
// launch number (`epoch`) that the words carry in their upper half.
struct ChunkScan {
    unsigned long long *words;       // [0] ticket counter (monotone), [1 + b] = epoch << 32 | bits of chunk b's maximum
    unsigned long long ticket_base;
    uint32_t epoch;
    int32_t nchunks;                 // tickets beyond the chunks do the launch's other work (k_prep_reads: the column -> read index)
    int64_t chunk;                   // reads per chunk, a multiple of 256
};

__device__ __forceinline__ int chunk_ticket(const ChunkScan &cs)
{
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = (int)(atomicAdd(&cs.words[0], 1ull) - cs.ticket_base);
    __syncthreads();
    return s_ticket;
}

// one tile of 256 reads: `ke` = the read's end if it is kept, INT32_MIN otherwise (also for threads beyond the chunk)
__device__ __forceinline__ void chunk_tile(int ke, int &run, int32_t *maxend, int64_t i, bool in)
{
    __shared__ int s_wt[SCAN_BLOCK / 64];
    int tot;
    const int inc = block_scan_incl<OpMaxI32>(ke, s_wt, tot);
    if (in) maxend[i] = inc > run ? inc : run;
    run = tot > run ? tot : run;
}

__device__ __forceinline__ void chunk_finish(const ChunkScan &cs, int b, int run, int32_t *maxend, int64_t i0, int64_t i1)
{
    __shared__ int s_pre[SCAN_BLOCK / 64];
    if (threadIdx.x == 0) __hip_atomic_store(&cs.words[1 + b], ((unsigned long long)cs.epoch << 32) | (uint32_t)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int pre = INT32_MIN;
    for (int j = threadIdx.x; j < b; j += SCAN_BLOCK) {
        unsigned long long v;
        // (the word IS the value: one device-scope store, no fence pair -- as in dev_lookback.h)
        while ((uint32_t)((v = __hip_atomic_load(&cs.words[1 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != cs.epoch) __builtin_amdgcn_s_sleep(2);
        const int m = (int)(uint32_t)v;
        pre = m > pre ? m : pre;
    }
    for (int o = 32; o; o >>= 1) { const int y = __shfl_down(pre, o); pre = y > pre ? y : pre; }
    if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = pre;
    __syncthreads();
    pre = s_pre[0];
    for (int w = 1; w < SCAN_BLOCK / 64; ++w) pre = s_pre[w] > pre ? s_pre[w] : pre;
    if (pre == INT32_MIN) return;
    // (every thread re-reads what it wrote itself; maxend is non-decreasing, so only the chunk's first entries change)
    for (int64_t i = i0 + threadIdx.x; i < i1; i += SCAN_BLOCK) if (maxend[i] < pre) maxend[i] = pre;
}

// ------------------------------------------------------------------------------------------------
struct PrepArgs {
    int32_t min_mq, rflag_require, rflag_filter, flag, all, baq_force_slow, min_qlen, baq_class_s;
};

// wfirst[f][w] = first read of file f that starts at or beyond column col_beg + 64 w (w = 0 .. nwaves): one thread per entry, a
// binary search each over the INPUT positions.  The tile kernels find their reads from it with one more coalesced load instead of
// two 64-ary searches (eight dependent loads at the head of every wave).
__device__ __forceinline__ void wave_first_entry(const StaWinDev &W, uint32_t *__restrict__ wfirst, int64_t nwaves, int64_t i)
{
    if (i >= (nwaves + 1) * W.nfiles) return;
    const int f = (int)(i / (nwaves + 1)); const int64_t w = i - (int64_t)f * (nwaves + 1);
    const StaReadsDev &R = W.files[f];
    const int64_t key = (int64_t)W.col_beg + 64 * w;
    int64_t lo = 0, hi = R.n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)R.pos[mid] >= key) hi = mid; else lo = mid + 1; }
    wfirst[i] = (uint32_t)lo;
}

__global__ void __launch_bounds__(256) k_prep_reads(StaReadsDev R, StaWinDev W, PrepArgs P, StaCounters *ctr, ChunkScan cs, uint32_t *wfirst, int64_t nwaves)
{
    // a chunk of consecutive reads per workgroup (see ChunkScan); the counters are reduced ONCE per block at the end (a counter word
    // sustains only ~90 atomics/us, so one reduction per 256 reads cost more than the reads themselves)
    const int b = chunk_ticket(cs);
    if (b >= cs.nchunks) {       // the launch's other work: the tile kernels' column -> read index of every file (first file's launch only)
        wave_first_entry(W, wfirst, nwaves, (int64_t)(b - cs.nchunks) * 256 + threadIdx.x);
        return;
    }
    // class-S candidates per read length: counted in LDS over the block's whole chunk, one device atomic per block and distinct length at the
    // end (one per WAVE cost 0.1 ms at bench size: 13 000 atomics on the one word of the common length)
    __shared__ int s_hist[STA_SLIST_BINS];
    const bool want_hist = P.baq_class_s && R.s_ws != nullptr;
    if (want_hist) { for (int k = threadIdx.x; k < STA_SLIST_BINS; k += blockDim.x) s_hist[k] = 0; __syncthreads(); }
    unsigned long long piled = 0, kept = 0;
    unsigned long long c_baq = 0, c_fast = 0, c_bw8 = 0, c_gen = 0, m_lqf = 0, m_lq = 0, m_bw = 0, c_s = 0, m_lqs = 0, c_bw7l = 0, c_olap = 0;
    const int64_t i0 = (int64_t)b * cs.chunk, i1 = i0 + cs.chunk < R.n ? i0 + cs.chunk : R.n;
    int run = INT32_MIN;
    for (int64_t base = i0; base < i1; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool in = i < i1;
        int ke = INT32_MIN;
        if (in) {
        int32_t pos = R.pos[i];
        uint32_t flag = R.flag[i];
        uint32_t c0 = R.cig_off[i], c1 = R.cig_off[i + 1];
        int32_t rlen = 0;
        bool has_m = false, has_n = false;
        for (uint32_t k = c0; k < c1; ++k) {
            uint32_t c = R.cigar[k];
            int op = c & 0xf;
            if (cg_is_refop(op)) rlen += (int32_t)(c >> 4);
            has_m |= cg_is_mop(op);
            has_n |= (op == CG_N);
        }
        int32_t end = pos + rlen;
        int32_t lq = R.l_qseq[i];
        uint32_t aux = R.aux[i];
        uint32_t mapq = R.mapq[i];
        int64_t apos = W.origin + pos;
        bool pushed = !(flag & BAM_FUNMAP);
        if (P.rflag_require && !(P.rflag_require & flag)) pushed = false;
        if (P.rflag_filter && (P.rflag_filter & flag)) pushed = false;
        if (pushed && W.has_bed && P.all == 0) {
            int64_t endpos = apos + (rlen > 0 ? rlen : 1);
            pushed = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, endpos);
        }
        if (aux & STA_AUX_SKIP) pushed = false;
        bool has_ref = W.ref != nullptr;
        if (has_ref && W.ref_len <= apos && !(P.flag & STA_MPLP_INT_CALMD)) pushed = false;   // "Skipping because ... is outside of ..."
        if (P.min_qlen) {
            // coverage -l: bam_cigar2qlen (coverage.c:189)
            int32_t ql = 0;
            for (uint32_t k = c0; k < c1; ++k) { int op = R.cigar[k] & 0xf; if (op == CG_M || op == CG_I || op == CG_S || op == CG_EQ || op == CG_X) ql += (int32_t)(R.cigar[k] >> 4); }
            if (ql < P.min_qlen) pushed = false;
        }
        if ((int32_t)mapq < P.min_mq) pushed = false;
        else if ((P.flag & STA_MPLP_NO_ORPHAN) && (flag & BAM_FPAIRED) && !(flag & BAM_FPROPER_PAIR)) pushed = false;

        bool keep = pushed && rlen > 0;
        bool simple = (c1 - c0 == 1) && cg_is_mop(R.cigar[c0] & 0xf) && lq == rlen;   // one M op covering the whole query
        uint32_t info = (pushed ? RI_PUSHED : 0) | (keep ? RI_KEEP : 0) | (simple ? RI_SIMPLE : 0)
                      | ((flag & BAM_FREVERSE) ? RI_REV : 0) | (mapq << RI_MAPQ_SHIFT);
        // overlap_push eligibility
        if (keep && (P.flag & STA_MPLP_SMART_OVERLAPS) && !(flag & BAM_FMUNMAP) && (flag & BAM_FPROPER_PAIR)) {
            int32_t mtid = R.mtid[i];
            int64_t mpos = R.mpos[i];
            long long isz = R.isize[i]; if (isz < 0) isz = -isz;
            bool no = (mtid >= 0 && mtid != W.tid) || (isz >= 2ll * lq && mpos >= W.origin + end);
            if (!no) { info |= RI_OLAP_EL; c_olap += 1; }
        }
        // realn.c's first early return (unmapped, no bases, qual[0] == 0xff) comes BEFORE it looks for BQ:Z: such a record keeps its
        // qualities whatever its tag says.  k_qual_prep applied the tag pool to every byte; put this read's bytes back.  (Found by
        // scripts/hunt6.py on the CPU emulation, round 5: calmd -r -A printed 255 - (BQ - 64) for reads without QUAL.)
        if ((P.flag & STA_MPLP_REALN) && W.ref != nullptr && !(P.flag & STA_MPLP_REDO_BAQ) && R.bq != nullptr && (aux & STA_AUX_HAS_BQ)
            && R.qual != R.qual_in && lq > 0) {
            const uint64_t bo = (uint64_t)R.base_off8[i] << 3;
            if ((flag & BAM_FUNMAP) || (R.qual_in[bo] == 0xff && !(P.flag & STA_MPLP_ILLUMINA13)))
                for (int32_t k = 0; k < lq; ++k) R.qual[bo + k] = R.qual_in[bo + k];
        }
        // BAQ needed? (realn.c early returns, A.4)  baq_cls: 0 none, 1 band width 7 in place, 2 through the list, 3 class-S candidate
        int baq_cls = 0, baq_bw = 0, baq_gbw = 0;
        if (pushed && (P.flag & STA_MPLP_REALN) && has_ref && lq > 0 && has_m && !has_n) {
            bool redo = (P.flag & STA_MPLP_REDO_BAQ) != 0;
            // realn.c gives up when qual[0] == 0xff -- tested on the record mplp_func has ALREADY shifted for -6 (bam_plcmd.c:431-435 runs
            // before :451): a read without qualities is then 224 everywhere and does get realigned (found by scripts/hunt5.py on the
            // CPU emulation, round 5)
            bool q_absent = R.qual_in[(uint64_t)R.base_off8[i] << 3] == 0xff && !(P.flag & STA_MPLP_ILLUMINA13);
            if (!q_absent && !(aux & STA_AUX_HAS_ZQ) && (redo || !(aux & STA_AUX_HAS_BQ))) {
                BaqGeo g = baq_geometry(R.cigar + c0, (int)(c1 - c0), apos, lq, W.ref, W.ref_len);
                if (g.ok) {         // !ok: probaln_glocal returns 0 and the qualities stay as they are
                    info |= RI_BAQ;
                    c_baq += 1;
                    bool band = (g.bw == 7 || g.bw == 8) && lq <= STA_BAQ7_LQ_MAX && !P.baq_force_slow;
                    baq_bw = band ? g.bw : 0; baq_gbw = g.bw;
                    baq_cls = (band && g.bw == 7) ? 1 : 2;
                    if (baq_cls == 1 && P.baq_class_s && baq7s::classify(R.cigar + c0, (int)(c1 - c0), apos, lq, W.ref_len).ok && g.l_ref == lq + 6) baq_cls = 3;
                }
            }
            // both BQ and ZQ without redo: ZQ is dropped and BQ applied by k_qual_prep
        }
        if (P.baq_class_s) {
            // class S wants ONE read length per group of 64 lanes (its parameters live in scalar registers).  Round 4 formed the groups from
            // 64 CONSECUTIVE reads and sent every candidate of another length than the group's first -- and every lane next to an indel
            // read stayed idle -- through the list kernels: trimmed reads (many lengths) lost the fast kernel altogether.  Now every
            // candidate is taken: the lengths are counted here (one atomic per wave and distinct length), the host lays the list out
            // per length and k_baq7s_gather fills it (kernels_baq.hip).
            if (baq_cls == 1) baq_cls = 2;
            if (R.s_ws) {
                unsigned long long todo = __ballot(baq_cls == 3);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    const int l0 = __shfl(lq, src);
                    const unsigned long long same = __ballot(baq_cls == 3 && lq == l0);
                    if ((int)(threadIdx.x & 63) == src) atomicAdd(&s_hist[l0], (int)__popcll(same));
                    todo &= ~same;
                }
            } else if (baq_cls == 3) baq_cls = 2;
        }
        if (baq_cls) {
            if (baq_bw) {
                info |= (uint32_t)baq_bw << RI_BAQ_BW_SHIFT;
                if ((unsigned long long)lq > m_lqf) m_lqf = (unsigned long long)lq;
            } else {
                if ((unsigned long long)lq > m_lq) m_lq = (unsigned long long)lq;
                if ((unsigned long long)baq_gbw > m_bw) m_bw = (unsigned long long)baq_gbw;
            }
            if (baq_cls == 3) { info |= RI_BAQ_S; c_s += 1; if ((unsigned long long)lq > m_lqs) m_lqs = (unsigned long long)lq; }
            else if (baq_cls == 1) c_fast += 1;
            else {
                if (baq_bw == 7) c_bw7l += 1; else if (baq_bw == 8) c_bw8 += 1; else c_gen += 1;
                info |= RI_BAQ_SLOW;
                int slot = atomicAdd(&R.chain[0], 1);
                R.chain[1 + slot] = (int32_t)i;
            }
        }
        R.end[i] = end;
        R.info[i] = info;
        if (keep) {
            kept += 1;
            ke = end;
            int32_t ca = pos > W.col_beg ? pos : W.col_beg, cb = end < W.col_end ? end : W.col_end;
            if (cb > ca) piled += (unsigned long long)(cb - ca);
        }
        }
        chunk_tile(ke, run, R.maxend, i, in);
    }
    chunk_finish(cs, b, run, R.maxend, i0, i1);
    if (want_hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < STA_SLIST_BINS; k += blockDim.x) if (s_hist[k]) atomicAdd(&R.s_ws[k], s_hist[k]);
    }
    // block reduce, then one atomic per counter and block
    unsigned long long v[13] = { piled, kept, c_baq, c_fast, c_bw8, c_gen, c_s, c_bw7l, c_olap, m_lqf, m_lq, m_bw, m_lqs };
    unsigned long long *const dst[13] = { &ctr->piled_bases, &ctr->n_kept, &ctr->n_baq, &ctr->n_baq_fast, &ctr->n_baq_bw8, &ctr->n_baq_general,
                                          &ctr->n_baq_s, &ctr->n_baq_bw7l, &ctr->n_olap_el, &ctr->max_lq_fast, &ctr->max_lq, &ctr->max_bw, &ctr->max_lq_s };
    block_reduce_atomic<13, 9>(v, dst);
}

// chunks of a launch over n reads: at most 2048, each a multiple of 256 reads
static void chunk_geometry(int64_t n, ChunkScan &cs, StaChunkState &st)
{
    int64_t nc = (n + 255) / 256;
    if (nc > 2048) nc = 2048;
    cs.chunk = ((n + nc - 1) / nc + 255) / 256 * 256;
    cs.nchunks = (int32_t)((n + cs.chunk - 1) / cs.chunk);
    if (++st.epoch == 0) st.epoch = 1;
    cs.words = st.words; cs.ticket_base = st.tickets; cs.epoch = st.epoch;
}

bool sta_launch_prep_reads(hipStream_t s, const StaWinDev &w, const StaReadsDev *files_host, int nfiles,
                           const sta_mplp_params &p, StaCounters *ctr, StaChunkState &st, uint32_t *wfirst)
{
    // STA_BAQ_CLASS_S=0: the round-3 kernels take every band-width-7 read in place (A/B measurements, tests of both paths)
    static const int class_s = [] { const char *e = getenv("STA_BAQ_CLASS_S"); return e ? atoi(e) : 1; }();
    PrepArgs a{ p.min_mq, p.rflag_require, p.rflag_filter, p.flag, p.all, getenv("STA_BAQ_FORCE_SLOW") ? 1 : 0, p.min_qlen, class_s && !getenv("STA_BAQ_FORCE_SLOW") };
    const int64_t ncols = (int64_t)w.col_end - w.col_beg, nwaves = (ncols + 63) / 64;
    bool wf_done = !(wfirst && ncols > 0 && w.nfiles > 0);
    for (int f = 0; f < nfiles; ++f) {
        const StaReadsDev &R = files_host[f];
        if (R.n == 0) continue;
        ChunkScan cs;
        chunk_geometry(R.n, cs, st);
        int64_t nb = cs.nchunks;
        const bool wf = !wf_done;            // the first launch also builds the column -> read index of EVERY file (input positions only)
        if (wf) nb += ((nwaves + 1) * w.nfiles + 255) / 256;
        hipMemsetAsync(R.chain, 0, 4, s);
        hipLaunchKernelGGL(k_prep_reads, dim3((unsigned)nb), dim3(256), 0, s, R, w, a, ctr, cs, wf ? wfirst : nullptr, nwaves);
        st.tickets += (unsigned long long)nb;
        wf_done = wf_done || wf;
    }
    return wf_done;
}

// ------------------------------------------------------------------------------------------------
// -C / --adjust-MQ: HTSlib realn.c sam_cap_mapq (absent from the reference tree; call site bam_plcmd.c:453-457, run AFTER
// BAQ, so it sees the BAQ-adjusted qualities).  One thread per read that is still pushed: count mismatches with quality >= 13
// over the aligned bases, turn them into a cap for the mapping quality, then re-apply the -q filter (bam_plcmd.c:458).
// sam_cap_mapq for read r: the cap, or -1 when the read's mismatch score exceeds the threshold
__device__ __forceinline__ int cap_mapq_of(const StaReadsDev &R, const StaWinDev &W, int64_t r, int thres)
{
    const uint8_t *qual = R.qual + ((uint64_t)R.base_off8[r] << 3);
    const uint8_t *seq = R.seq + ((uint64_t)R.base_off8[r] << 2);
    int mm = 0, q = 0, len = 0, clip_l = 0, clip_q = 0, y = 0;
    // A record without SEQ (l_qseq 0 under a CIGAR with M operations) makes HTSlib's loop read the bytes behind the record: undefined
    // there.  Here, and in the oracle, a base that is not there is neither a mismatch nor a clipped quality.
    const int lq = R.l_qseq[r];
    long long x = W.origin + R.pos[r];
    bool stop = false;
    if (thres < 0) thres = 40;
    for (uint32_t k = R.cig_off[r]; k < R.cig_off[r + 1] && !stop; ++k) {
        int op = R.cigar[k] & 0xf, l = (int)(R.cigar[k] >> 4);
        if (cg_is_mop(op)) {
            int j;
            for (j = 0; j < l; ++j) {
                int z = y + j;
                if (x + j >= W.ref_len) break;
                if (z >= lq) continue;
                int c1 = (seq[z >> 1] >> ((~z & 1) << 2)) & 0xf, c2 = nt16_from_char((unsigned char)W.ref[x + j]);
                if (c2 != 15 && c1 != 15 && qual[z] >= 13) {
                    ++len;
                    if (c1 && c1 != c2 && qual[z] >= 13) { ++mm; q += qual[z] > 33 ? 33 : qual[z]; }
                }
            }
            if (j < l) { stop = true; break; }
            x += l; y += l; len += l;
        } else if (op == CG_D) {
            if (x + l > W.ref_len) { stop = true; break; }
            x += l;
        } else if (op == CG_S) {
            for (int j = 0; j < l; ++j) if (y + j < lq) clip_q += qual[y + j];
            clip_l += l; y += l;
        } else if (op == CG_H) { clip_q += 13 * l; clip_l += l; }
        else if (op == CG_I) y += l;
        else if (op == CG_N) x += l;
    }
    (void)clip_l;
    double t = 1;
    for (int i = 0; i < mm; ++i) t *= (double)len / (i + 1);
    t = q - 4.343 * log(t) + clip_q / 5.;
    if (t > thres) return -1;
    if (t < 0) t = 0;
    t = sqrt((thres - t) / thres) * thres;
    return (int)(t + .499);
}

// calmd -C (bam_md.c:480-483): the cap of every staged read, for the host to apply to the record's MAPQ field
__global__ void __launch_bounds__(256) k_cap_mapq_vals(StaReadsDev R, StaWinDev W, int thres, int16_t *cap)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R.n) cap[r] = (int16_t)cap_mapq_of(R, W, r, thres);
}

__global__ void __launch_bounds__(256) k_cap_mapq(StaReadsDev R, StaWinDev W, int thres, int min_mq, StaCounters *ctr)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long d_kept = 0, d_piled = 0;
    if (r < R.n) {
        uint32_t info = R.info[r];
        if (info & RI_PUSHED) {
            const int cap = cap_mapq_of(R, W, r, thres);
            int mapq = (int)((info >> RI_MAPQ_SHIFT) & 0xff);
            bool drop = cap < 0;
            if (!drop && mapq > cap) mapq = cap;
            if (!drop && mapq < min_mq) drop = true;
            if (drop) {
                if (info & RI_KEEP) {
                    d_kept = 1;
                    int32_t a = R.pos[r] > W.col_beg ? R.pos[r] : W.col_beg, b = R.end[r] < W.col_end ? R.end[r] : W.col_end;
                    if (b > a) d_piled = (unsigned long long)(b - a);
                }
                info &= ~(RI_PUSHED | RI_KEEP | RI_OLAP_EL);
            }
            info = (info & ~(0xffu << RI_MAPQ_SHIFT)) | ((uint32_t)mapq << RI_MAPQ_SHIFT);
            R.info[r] = info;
        }
    }
    // the counters were filled by k_prep_reads: take the dropped reads out again (unsigned wrap-around subtraction)
    unsigned long long v[2] = { 0ull - d_piled, 0ull - d_kept };
    unsigned long long *const dst[2] = { &ctr->piled_bases, &ctr->n_kept };
    block_reduce_atomic<2, 2>(v, dst);
}

void sta_launch_cap_mapq_vals(hipStream_t s, const StaReadsDev &R, const StaWinDev &w, int thres, int16_t *cap)
{
    if (R.n == 0) return;
    hipLaunchKernelGGL(k_cap_mapq_vals, dim3((unsigned)((R.n + 255) / 256)), dim3(256), 0, s, R, w, thres, cap);
}

void sta_launch_cap_mapq(hipStream_t s, const StaReadsDev &R, const StaWinDev &w, int thres, int min_mq, StaCounters *ctr)
{
    if (R.n == 0) return;
    hipLaunchKernelGGL(k_cap_mapq, dim3((unsigned)((R.n + 255) / 256)), dim3(256), 0, s, R, w, thres, min_mq, ctr);
}

// ------------------------------------------------------------------------------------------------
struct PrepDepthArgs { int32_t flag, incl_flag, require_flag, min_mqual, min_len; };

__global__ void __launch_bounds__(256) k_prep_reads_depth(StaReadsDev R, StaWinDev W, PrepDepthArgs P, StaCounters *ctr, uint4 *zero, int64_t zero_n16, ChunkScan cs)
{
    // (also clears the look-back status words of the depth kernel that follows: one small launch less per window)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n16; i += (int64_t)gridDim.x * blockDim.x) zero[i] = make_uint4(0, 0, 0, 0);
    unsigned long long piled = 0, kept = 0;
    const int b = chunk_ticket(cs);
    const int64_t i0 = (int64_t)b * cs.chunk, i1 = i0 + cs.chunk < R.n ? i0 + cs.chunk : R.n;
    int run = INT32_MIN;
    for (int64_t base = i0; base < i1; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool in = i < i1;
        int ke = INT32_MIN;
        if (in) {
        int32_t pos = R.pos[i];
        uint32_t flag = R.flag[i];
        uint32_t c0 = R.cig_off[i], c1 = R.cig_off[i + 1];
        int32_t rlen = 0;
        int64_t qused = 0;
        for (uint32_t k = c0; k < c1; ++k) {
            uint32_t c = R.cigar[k];
            int op = c & 0xf;
            if (cg_is_refop(op)) rlen += (int32_t)(c >> 4);
            if (op == CG_M || op == CG_I || op == CG_EQ || op == CG_X) qused += (c >> 4);
        }
        int32_t lq = R.l_qseq[i];
        bool ok = true;
        if (flag & P.flag) ok = false;
        if (P.incl_flag && (flag & P.incl_flag) == 0) ok = false;
        if ((flag & P.require_flag) != (uint32_t)P.require_flag) ok = false;
        if ((int32_t)R.mapq[i] < P.min_mqual) ok = false;
        if (ok && P.min_len) {
            // qlen_used (bam2depth.c:124-159)
            int64_t l;
            if (lq) {
                l = lq;
                uint32_t kl, kr;
                for (kl = c0; kl < c1; kl++) { if ((R.cigar[kl] & 0xf) == CG_S) l -= (R.cigar[kl] >> 4); else break; }
                for (kr = c1; kr > kl + 1; kr--) { if ((R.cigar[kr - 1] & 0xf) == CG_S) l -= (R.cigar[kr - 1] >> 4); else break; }
            } else l = qused;
            if (l < P.min_len) ok = false;
        }
        // bam_endpos: pos + max(rlen,1) (unmapped-flagged reads count as length 1)
        int32_t span = (flag & BAM_FUNMAP) ? 0 : rlen;
        int32_t end = pos + (span > 0 ? span : 1);
        // the column walker finds a read through [pos, R.end): the CIGAR's own reach, which is longer than bam_endpos for an
        // unmapped-flagged record that still carries a CIGAR (add_depth counts along the CIGAR whatever the flag says);
        // RI_UNMAP_SPAN keeps the "row is covered" span at bam_endpos for those
        const int32_t cig_end = pos + (rlen > 0 ? rlen : 1);
        const bool simple = (c1 - c0 == 1) && cg_is_mop(R.cigar[c0] & 0xf) && lq == rlen && !(flag & BAM_FUNMAP);
        R.end[i] = cig_end;
        R.info[i] = (ok ? (RI_PUSHED | RI_KEEP) : 0) | ((flag & BAM_FREVERSE) ? RI_REV : 0) | (simple ? RI_SIMPLE : 0) | (cig_end != end ? RI_UNMAP_SPAN : 0);
        {
            // depth -s with the caller's name hash (sta_reads.olap_clip): the clip column, window relative
            int32_t cl = 0;
            if (R.clip_in) {
                const long long c = (long long)R.clip_in[i];
                if (c) { const long long rel = c - (long long)W.origin; cl = (int32_t)(rel > INT32_MAX ? INT32_MAX : (rel < INT32_MIN + 1 ? INT32_MIN + 1 : rel)); }
            }
            R.clip[i] = cl;
        }
        if (ok) {
            kept += 1;
            ke = cig_end;
            int32_t ca = pos > W.col_beg ? pos : W.col_beg, cb = end < W.col_end ? end : W.col_end;
            if (cb > ca) piled += (unsigned long long)(cb - ca);
        }
        }
        chunk_tile(ke, run, R.maxend, i, in);
    }
    chunk_finish(cs, b, run, R.maxend, i0, i1);
    unsigned long long v[2] = { piled, kept };
    unsigned long long *const dst[2] = { &ctr->piled_bases, &ctr->n_kept };
    block_reduce_atomic<2, 2>(v, dst);
}

bool sta_launch_prep_reads_depth(hipStream_t s, const StaWinDev &w, const StaReadsDev *files_host, int nfiles,
                                 const sta_depth_params &p, StaCounters *ctr, StaChunkState &st, void *zero, size_t zero_bytes)
{
    PrepDepthArgs a{ p.flag, p.incl_flag, p.require_flag, p.min_mqual, p.min_len };
    bool zeroed = false;                       // `zero` (16-byte aligned, a multiple of 16 bytes) is cleared by the first launch
    for (int f = 0; f < nfiles; ++f) {
        const StaReadsDev &R = files_host[f];
        if (R.n == 0) continue;
        ChunkScan cs;
        chunk_geometry(R.n, cs, st);
        const bool z = !zeroed && zero && zero_bytes % 16 == 0;
        hipLaunchKernelGGL(k_prep_reads_depth, dim3((unsigned)cs.nchunks), dim3(256), 0, s, R, w, a, ctr, (uint4 *)(z ? zero : nullptr), (int64_t)(z ? zero_bytes / 16 : 0), cs);
        st.tickets += (unsigned long long)cs.nchunks;
        zeroed = zeroed || z;
    }
    return zeroed;
}

// ------------------------------------------------------------------------------------------------
// Working-quality pool = f(input pool): 16 bytes per lane, fully coalesced.
__global__ void __launch_bounds__(256) k_qual_prep(const uint8_t *__restrict__ qin, const uint8_t *__restrict__ bq,
                                                   uint8_t *__restrict__ qout, uint64_t nbytes, int illumina13)
{
    uint64_t n16 = nbytes >> 4;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 v = reinterpret_cast<const uint4 *>(qin)[i];
        uint4 b = bq ? reinterpret_cast<const uint4 *>(bq)[i] : make_uint4(0, 0, 0, 0);
        uint32_t w[4] = { v.x, v.y, v.z, v.w }, bb[4] = { b.x, b.y, b.z, b.w };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t q = (w[j] >> (8 * k)) & 0xff;
                if (illumina13) q = q > 31 ? q - 31 : 0;
                if (bq) {
                    uint32_t t = (bb[j] >> (8 * k)) & 0xff;
                    q = (q + 64 < t) ? 0 : (q - (t - 64)) & 0xff;
                }
                o |= q << (8 * k);
            }
            w[j] = o;
        }
        reinterpret_cast<uint4 *>(qout)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // tail (pool is padded to 8 bases per read, so at most 8 trailing bytes)
    if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15)) {
        uint64_t i = (n16 << 4) + threadIdx.x;
        uint32_t q = qin[i];
        if (illumina13) q = q > 31 ? q - 31 : 0;
        if (bq) { uint32_t t = bq[i]; q = (q + 64 < t) ? 0 : (q - (t - 64)) & 0xff; }
        qout[i] = (uint8_t)q;
    }
}

void sta_launch_qual_prep(hipStream_t s, const StaReadsDev &r, int illumina13)
{
    if (r.n_bases_total == 0) return;
    uint64_t n16 = r.n_bases_total >> 4;
    uint64_t nb = (n16 + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_qual_prep, dim3((unsigned)nb), dim3(256), 0, s, r.qual_in, r.bq, r.qual, r.n_bases_total, illumina13);
}

// ------------------------------------------------------------------------------------------------
// Generic three-phase scan.  TILE elements per 256-thread block.

// Loader functors turn input element i into Op::T
struct LoadU32 { const uint32_t *p; __device__ unsigned long long operator()(int64_t i) const { return p[i] & 0x7fffffffu; } };   // bit 31 of a line length = "column has data"
struct LoadI32 { const int32_t *p; __device__ int operator()(int64_t i) const { return p[i]; } };
struct LoadKeptEnd { const int32_t *end; const uint32_t *info;
    __device__ int operator()(int64_t i) const { return (info[i] & RI_KEEP) ? end[i] : INT32_MIN; } };

template <class Op, class Load> __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_reduce(Load ld, int64_t n, typename Op::T *block_sums)
{
    typedef typename Op::T T;
    __shared__ T wt[SCAN_BLOCK / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T acc = Op::id();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < n) acc = Op::f(acc, ld(base + j));
    T tot;
    block_scan_incl<Op>(acc, wt, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block sums in place; total stored at [nb]
template <class Op> __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_sums(typename Op::T *block_sums, int64_t nb)
{
    typedef typename Op::T T;
    __shared__ T wt[SCAN_BLOCK / 64];
    T carry = Op::id();
    for (int64_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
        int64_t i = b0 + threadIdx.x;
        T v = i < nb ? block_sums[i] : Op::id();
        T tot;
        T inc = block_scan_incl<Op>(v, wt, tot);
        // exclusive = carry (+) (inc without v) -> recompute from neighbours: use shuffle of inc
        T prev = __shfl_up(inc, 1);
        int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        __shared__ T last_of_wave[SCAN_BLOCK / 64];
        if (lane == 63) last_of_wave[wid] = inc;
        __syncthreads();
        T ex = lane ? prev : (wid ? last_of_wave[wid - 1] : Op::id());
        if (i < nb) block_sums[i] = Op::f(carry, ex);
        carry = Op::f(carry, tot);
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry;
}

template <class Op, class Load, class Tout, bool EXCL> __global__ void __launch_bounds__(SCAN_BLOCK)
k_scan_apply(Load ld, int64_t n, const typename Op::T *block_sums, Tout *out)
{
    typedef typename Op::T T;
    __shared__ T wt[SCAN_BLOCK / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = Op::id();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = (base + j < n) ? ld(base + j) : Op::id(); acc = Op::f(acc, v[j]); }
    T tot;
    T inc = block_scan_incl<Op>(acc, wt, tot);
    // exclusive prefix of this thread = block prefix (+) (inclusive scan of previous thread)
    T prev = __shfl_up(inc, 1);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __shared__ T last_of_wave[SCAN_BLOCK / 64];
    if (lane == 63) last_of_wave[wid] = inc;
    __syncthreads();
    T ex = lane ? prev : (wid ? last_of_wave[wid - 1] : Op::id());
    T run = Op::f(block_sums[blockIdx.x], ex);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) {
            if (EXCL) { out[base + j] = (Tout)run; run = Op::f(run, v[j]); }
            else { run = Op::f(run, v[j]); out[base + j] = (Tout)run; }
        }
    }
    if (EXCL && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        // grand total at out[n]
        out[n] = (Tout)block_sums[gridDim.x];
    }
}

size_t sta_scan_tmp_bytes(int64_t n)
{
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    return (size_t)(nb + 2) * sizeof(unsigned long long);
}

template <class Op, class Load, class Tout, bool EXCL>
static void run_scan(hipStream_t s, Load ld, int64_t n, Tout *out, void *tmp)
{
    typedef typename Op::T T;
    if (n <= 0) {
        return;
    }
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    T *sums = reinterpret_cast<T *>(tmp);
    hipLaunchKernelGGL((k_scan_reduce<Op, Load>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, ld, n, sums);
    hipLaunchKernelGGL((k_scan_sums<Op>), dim3(1), dim3(SCAN_BLOCK), 0, s, sums, nb);
    hipLaunchKernelGGL((k_scan_apply<Op, Load, Tout, EXCL>), dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, ld, n, sums, out);
}

void sta_launch_maxend_scan(hipStream_t s, const StaReadsDev &r, void *tmp, size_t)
{
    LoadKeptEnd ld{ r.end, r.info };
    run_scan<OpMaxI32, LoadKeptEnd, int32_t, false>(s, ld, r.n, r.maxend, tmp);
}

// inclusive running maximum of an int32 array (consensus: last column of the reads so far)
void sta_launch_scan_max_i32(hipStream_t s, const int32_t *in, int32_t *out, int64_t n, void *tmp)
{
    LoadI32 ld{ in };
    run_scan<OpMaxI32, LoadI32, int32_t, false>(s, ld, n, out, tmp);
}

__global__ void k_set_u64(uint64_t *p, uint64_t v) { *p = v; }

void sta_launch_len_scan(hipStream_t s, const uint32_t *len, uint64_t *offs, int64_t n, void *tmp, size_t)
{
    if (n <= 0) { hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, s, offs, (uint64_t)0); return; }
    LoadU32 ld{ len };
    run_scan<OpSumU64, LoadU32, uint64_t, true>(s, ld, n, offs, tmp);
}

// Column statistics after the scan: max over waves (64 columns) of the output bytes a wave must stage
// (sizes the emit kernels' LDS), number of output rows, number of columns with data.  Grid-stride,
// block-reduced: a few thousand atomics in total.
__global__ void __launch_bounds__(256) k_col_stats(const uint64_t *offs, const uint32_t *line_len, int64_t ncols, StaCounters *ctr)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long lines = 0, data = 0, mx = 0;
    for (int64_t c = tid; c < ncols; c += stride) {
        uint32_t u = line_len[c];
        lines += (u & 0x7fffffffu) != 0;
        data += u >> 31;
    }
    const int64_t nw = (ncols + 63) / 64;
    for (int64_t w = tid; w < nw; w += stride) {
        int64_t c0 = w * 64, c1 = c0 + 64 < ncols ? c0 + 64 : ncols;
        unsigned long long b = offs[c1] - offs[c0];
        mx = b > mx ? b : mx;
    }
    unsigned long long v[3] = { lines, data, mx };
    unsigned long long *const dst[3] = { &ctr->n_lines, &ctr->n_data_cols, &ctr->max_wave_bytes };
    block_reduce_atomic<3, 2>(v, dst);
}

void sta_launch_wave_bytes_max(hipStream_t s, const uint64_t *offs, const uint32_t *line_len, int64_t ncols, StaCounters *ctr)
{
    if (ncols <= 0) return;
    int64_t nb = (ncols + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_col_stats, dim3((unsigned)nb), dim3(256), 0, s, offs, line_len, ncols, ctr);
}
