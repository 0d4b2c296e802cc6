// kernels_plp.hip -- the mpileup column kernels (gfx950, wave64).
//
// Replaces the reference's per-column hot loops: HTSlib bam_plp64_next + resolve_cigar2
// (callers bam_plcmd.c:607) and samtools' mpileup()/pileup_seq() text assembly
// (bam_plcmd.c:54-169, :663-868), plus print_empty_pileup (:372-398).
//
// Layout of the work: one wavefront owns 64 consecutive reference columns, one lane per column.
// The reads that can touch those columns form a contiguous index range of the position-sorted
// read arrays, found with a wave-cooperative 64-ary search on `maxend` (prefix max of read ends)
// and `pos`.  The wave then walks that range *uniformly* (read metadata becomes scalar loads);
// each lane tests coverage of its own column, resolves its CIGAR position, applies the base
// quality filter and appends its token.  Because reads are walked in file order and every lane
// appends to its own line, the reference's "column entries in file order" rule holds by
// construction, with no sort.
//
//  k_mplp_len  : measuring pass -> bytes of every output line (0 = column not printed)
//  (scan)      : exclusive scan of line lengths -> line offsets
//  k_mplp_emit : writes the text.  A wave's lines are contiguous in the output, so they are
//                assembled in LDS and flushed with coalesced 16-byte stores; a wave whose
//                lines exceed the LDS slice writes straight to the text, eight bytes per store.
//                Since round 5 one measuring walk + one writing walk with a cursor per string
//                of the row (emit_column_1walk); one walk per string for rows with more than
//                GEN_NX extra columns (emit_column).
// These two are the any-option-set path (--output-extra / -O / tag columns / --output-mods); windows without such
// columns take the tile / read-major kernels further down.
#include "dev_util.h"
#include "plp_tile.h"
#include "dev_lookback.h"
#include <cstdlib>
#include <type_traits>

// seq_nt16_table (hts.c): character -> 4-bit code, 15 for anything else (a table: the switch in nt16_from_char costs ~300
// scalar instructions of exec-mask juggling per wave when every lane holds a different character)
__constant__ unsigned char c_nt16_of_char[256] = {
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,  1, 2, 4, 8, 15,15,15,15, 15,15,15,15, 15, 0,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15, 15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    15, 1,14, 2, 13,15,15, 4, 11,15,15,12, 15, 3,15,15, 15,15, 5, 6,  8,15, 7, 9, 15,10,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15,
    15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15, 15,15,15,15 };

#define EXTRA_MASK (STA_MPLP_PRINT_MAPQ_CHAR | STA_MPLP_PRINT_QPOS | STA_MPLP_PRINT_QNAME | STA_MPLP_PRINT_FLAG | \
                    STA_MPLP_PRINT_RNAME | STA_MPLP_PRINT_POS | STA_MPLP_PRINT_MAPQ | STA_MPLP_PRINT_RNEXT | STA_MPLP_PRINT_PNEXT | \
                    STA_MPLP_PRINT_RLEN | STA_MPLP_PRINT_QPOS5)

// extra per-read columns (bam_plcmd.c:727-796)
__device__ __forceinline__ long long extra_value(const StaReadsDev &R, const StaWinDev &W, int kind, const Entry &e)
{
    switch (kind) {
    case STA_MPLP_PRINT_QPOS: return e.rs.qpos + 1;
    case STA_MPLP_PRINT_QPOS5: return (e.info & RI_REV) ? e.lq - e.rs.qpos + (e.rs.is_del ? 1 : 0) : e.rs.qpos + 1;
    case STA_MPLP_PRINT_FLAG: return R.flag[e.r];
    case STA_MPLP_PRINT_POS: return W.origin + e.rpos + 1;
    case STA_MPLP_PRINT_MAPQ: return (e.info >> RI_MAPQ_SHIFT) & 0xff;
    case STA_MPLP_PRINT_PNEXT: return R.mpos[e.r] + 1;
    case STA_MPLP_PRINT_RLEN: return e.lq;
    }
    return 0;
}
// host-formatted text column: RNEXT is column 0 when requested, tag t is column (RNEXT ? 1 : 0) + t
__device__ __forceinline__ int xcol_index(const MplpDevPar &P, int kind)
{
    int rn = (P.flag & STA_MPLP_PRINT_RNEXT) ? 1 : 0;
    return kind == STA_MPLP_PRINT_RNEXT ? 0 : rn + (kind - TAGKIND);
}
__device__ __forceinline__ int extra_len(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int kind, const Entry &e)
{
    if (kind == STA_MPLP_PRINT_RNEXT || kind >= TAGKIND) {
        const uint32_t *o = R.xcol_off + (uint64_t)e.r * (uint64_t)R.n_xcols + (uint64_t)xcol_index(P, kind);
        return (int)(o[1] - o[0]);
    }
    if (kind == STA_MPLP_PRINT_MAPQ_CHAR) return 1;
    if (kind == STA_MPLP_PRINT_QNAME) return (int)(R.name_off[e.r + 1] - R.name_off[e.r]) - 1;
    if (kind == STA_MPLP_PRINT_RNAME) return W.tname_len;
    long long v = extra_value(R, W, kind, e);
    return (v < 0 ? 1 : 0) + dec_digits((unsigned long long)(v < 0 ? -v : v));
}
// l bytes of device text into a sink.  KIND 2: eight bytes per load and append; the last 0..7 as a 4-, a 2- and a 1-byte load asked for
// together (a byte loop is l dependent load latencies per entry: a read name cost the walker more than everything else of its entry).
typedef uint32_t __attribute__((aligned(1))) text_u32u;
typedef uint16_t __attribute__((aligned(1))) text_u16u;
template <int LDS>
__device__ __forceinline__ void put_text(Sink<LDS> &s, const char *src, int l)
{
    if (LDS != 2) { for (int t = 0; t < l; ++t) s.put(src[t]); return; }
    int t = 0;
    for (; t + 8 <= l; t += 8) s.put_n(*reinterpret_cast<const sink_u64u *>(src + t), 8);
    const int rem = l - t;
    if (rem) {
        const int o2 = t + (rem & 4), o1 = o2 + (rem & 2);
        uint64_t v4 = 0, v2 = 0, v1 = 0;
        if (rem & 4) v4 = *reinterpret_cast<const text_u32u *>(src + t);
        if (rem & 2) v2 = *reinterpret_cast<const text_u16u *>(src + o2);
        if (rem & 1) v1 = (unsigned char)src[o1];
        s.put_n(v4 | v2 << (8 * (rem & 4)) | v1 << (8 * (rem & 6)), (uint32_t)rem);
    }
}
template <int LDS>
__device__ __forceinline__ void extra_write(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int kind, const Entry &e, Sink<LDS> &s)
{
    if (kind == STA_MPLP_PRINT_RNEXT || kind >= TAGKIND) {
        const uint32_t *o = R.xcol_off + (uint64_t)e.r * (uint64_t)R.n_xcols + (uint64_t)xcol_index(P, kind);
        const uint32_t o0 = o[0], o1 = o[1];
        put_text<LDS>(s, R.xcol_text + o0, (int)(o1 - o0));
    } else if (kind == STA_MPLP_PRINT_MAPQ_CHAR) {
        int c = (int)((e.info >> RI_MAPQ_SHIFT) & 0xff) + 33;
        s.put((char)(c > 126 ? 126 : c));
    } else if (kind == STA_MPLP_PRINT_QNAME) {
        const auto n0 = R.name_off[e.r], n1 = R.name_off[e.r + 1];
        put_text<LDS>(s, R.names + n0, (int)(n1 - n0) - 1);
    } else if (kind == STA_MPLP_PRINT_RNAME) {
        put_text<LDS>(s, W.tname, W.tname_len);
    } else s.put_dec(extra_value(R, W, kind, e));
}

// ---- the uniform walk over one file's candidate reads ----
// MODE 0: measure (n_plp, cnt, seq_len, extras_len)   MODE 1: count only (n_plp, cnt)
// MODE 2: write seq tokens   MODE 3: write qual chars   MODE 4: write one extra column `kind`
struct Acc { uint32_t n_plp, cnt, seq_len, extras_len; };

template <int MODE, bool LDS>
__device__ __forceinline__ void file_pass(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int p, bool active,
                                          int64_t rlo, int64_t rhi, int kind, Acc &acc, Sink<LDS> &s)
{
    uint32_t nw = 0;    // entries written so far (for the ',' separators)
    for (int64_t r = rlo; r < rhi; ++r) {
        uint32_t info = R.info[r];
        if (!(info & RI_KEEP)) continue;
        int rpos = R.pos[r], rend = R.end[r];
        bool cov = active && rpos <= p && p < rend;
        if (__ballot(cov) == 0) continue;
        if (!cov) continue;
        Entry e;
        e.r = r; e.rpos = rpos; e.rend = rend; e.info = info;
        e.lq = R.l_qseq[r];
        e.boff = (uint64_t)R.base_off8[r] << 3;
        if (info & RI_SIMPLE) { e.rs.qpos = p - rpos; e.rs.indel = 0; e.rs.k = 0; e.rs.is_del = false; e.rs.is_refskip = false; }
        else e.rs = resolve_general(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), rpos, p);
        if (MODE <= 1) acc.n_plp++;
        int c = e.rs.is_del ? placeholder_qual(R, r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
        if (c < P.min_baseQ) continue;
        if (MODE == 0) {
            acc.cnt++;
            acc.seq_len += (uint32_t)token_len(R, P, e, p);
            uint32_t ex = (uint32_t)P.flag & EXTRA_MASK;
            while (ex) {
                int kd = (int)(ex & (~ex + 1)); ex &= ex - 1;
                acc.extras_len += (uint32_t)extra_len(R, W, P, kd, e);
            }
            for (int t = 0; t < P.n_tags; ++t) acc.extras_len += (uint32_t)extra_len(R, W, P, TAGKIND + t, e);
        } else if (MODE == 1) {
            acc.cnt++;
        } else if (MODE == 2) {
            token_write<LDS>(R, W, P, e, p, s);
        } else if (MODE == 3) {
            s.put((char)(c + 33 < 126 ? c + 33 : 126));
        } else {
            if (nw > 0 && kind != STA_MPLP_PRINT_MAPQ_CHAR) s.put(kind >= TAGKIND ? (char)P.tag_sep : ',');
            extra_write<LDS>(R, W, P, kind, e, s);
        }
        nw++;
    }
}

__device__ __forceinline__ void wave_read_range(const StaReadsDev &R, int p0, int p1 /*last col*/, int64_t &rlo, int64_t &rhi)
{
    if (R.n == 0) { rlo = rhi = 0; return; }
    rlo = wave_upper_bound(R.maxend, R.n, p0);      // first read with an end beyond the first column
    rhi = wave_upper_bound(R.pos, R.n, p1);         // first read starting beyond the last column
    if (rlo > rhi) rlo = rhi;
}

// bytes of "\t cnt \t seq \t qual [\t extra]*" for one file
__device__ __forceinline__ uint32_t file_text_len(const MplpDevPar &P, const Acc &a)
{
    uint32_t n_extra = (uint32_t)__popc((uint32_t)P.flag & EXTRA_MASK) + (uint32_t)P.n_tags;
    uint32_t len = 1 + (uint32_t)dec_digits_u32(a.cnt) + 1 + (a.seq_len ? a.seq_len : 1) + 1 + (a.cnt ? a.cnt : 1);
    if (n_extra) {
        // every extra column: '\t' + (fields + separators, or '*')
        uint32_t n_sep_cols = n_extra - ((P.flag & STA_MPLP_PRINT_MAPQ_CHAR) ? 1u : 0u);
        if (a.cnt) len += n_extra + a.extras_len + n_sep_cols * (a.cnt - 1);
        else len += 2 * n_extra;
    }
    return len;
}

__device__ __forceinline__ bool column_selected(const StaWinDev &W, int64_t apos)
{
    if (W.has_reg && (apos < W.reg_beg || apos >= W.reg_end)) return false;
    return true;
}

__global__ void __launch_bounds__(256) k_mplp_len(StaWinDev W, MplpDevPar P, uint32_t *line_len, StaCounters *ctr)
{
    int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    int lane = threadIdx.x & 63;
    int64_t ncols = (int64_t)W.col_end - W.col_beg;
    int64_t c0 = (int64_t)wave * 64;
    if (c0 >= ncols) return;
    int p0 = W.col_beg + (int)c0;
    int p = p0 + lane;
    bool active = p < W.col_end;
    int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
    int64_t apos = W.origin + p;

    uint32_t total = 0; bool any = false;
    Sink<false> dummy; dummy.g = nullptr; dummy.cur = 0;
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        Acc a{ 0, 0, 0, 0 };
        file_pass<0, false>(R, W, P, p, active, rlo, rhi, 0, a, dummy);
        any |= a.n_plp > 0;
        total += file_text_len(P, a);
    }
    bool in_reg = active && column_selected(W, apos);
    bool data = in_reg && any;
    bool exists = in_reg && (any || (P.all && apos < P.tlen));
    if (exists && W.has_bed) exists = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
    uint32_t len = 0;
    if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1 + 1 + total + 1;
    if (active) line_len[c0 + lane] = len | (data ? 0x80000000u : 0u);     // rows / data columns are counted by k_col_stats
    (void)ctr;
}

template <bool LDS>
__device__ __forceinline__ void emit_column(const StaWinDev &W, const MplpDevPar &P, int p0, int plast, int p, bool exists, Sink<LDS> &s)
{
    int64_t apos = W.origin + p;
    if (exists) {
        for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
        s.put('\t');
        s.put_dec(apos + 1);
        s.put('\t');
        s.put((W.ref && apos < W.ref_len) ? W.ref[apos] : 'N');
    }
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        Acc a{ 0, 0, 0, 0 };
        file_pass<1, LDS>(R, W, P, p, exists, rlo, rhi, 0, a, s);
        if (exists) { s.put('\t'); s.put_dec(a.cnt); s.put('\t'); }
        // seq
        file_pass<2, LDS>(R, W, P, p, exists && a.cnt, rlo, rhi, 0, a, s);
        // a column whose tokens are all empty cannot happen (every token has >= 1 char)
        if (exists) { if (!a.cnt) s.put('*'); s.put('\t'); }
        file_pass<3, LDS>(R, W, P, p, exists && a.cnt, rlo, rhi, 0, a, s);
        if (exists && !a.cnt) s.put('*');
        uint32_t ex = (uint32_t)P.flag & EXTRA_MASK;
        while (ex) {
            int kd = (int)(ex & (~ex + 1)); ex &= ex - 1;
            if (exists) s.put('\t');
            file_pass<4, LDS>(R, W, P, p, exists && a.cnt, rlo, rhi, kd, a, s);
            if (exists && !a.cnt) s.put('*');
        }
        for (int t = 0; t < P.n_tags; ++t) {           // aux-tag columns, in --output-extra order (bam_plcmd.c:798-852)
            if (exists) s.put('\t');
            file_pass<4, LDS>(R, W, P, p, exists && a.cnt, rlo, rhi, TAGKIND + t, a, s);
            if (exists && !a.cnt) s.put('*');
        }
    }
    if (exists) s.put('\n');
}

// ---- the single-walk form of emit_column (round 5) ----
// emit_column above walks a file's candidate reads once per string of the row: count, bases, qualities and one walk per extra
// column -- six walks for "-s -O --output-QNAME", each ~80 latency-bound instructions per entry.  Here the first walk measures every
// string separately (the measuring kernel kept only their sum), which fixes where each string starts inside the row, and ONE more
// walk writes an entry's token, quality and extra fields through a cursor per string.  Up to GEN_NX extra columns; rows with more
// keep the per-column walks.
#define GEN_NX 8
#ifndef GEN_OCC
#define GEN_OCC __attribute__((amdgpu_waves_per_eu(4, 8)))     // 128 registers: four waves per SIMD
#endif
struct AccX { uint32_t n_plp, cnt, seq_len, xlen[GEN_NX]; };
template <int LDS> __device__ __forceinline__ void sink_adv(Sink<LDS> &s, int32_t n) { if (LDS == 1) s.cur += (uint32_t)n; else s.g += n; }
// a byte at `off` from the cursor, the cursor stays (a KIND 2 sink must be empty: its g is then the cursor)
template <int LDS> __device__ __forceinline__ void sink_poke(const Sink<LDS> &s, int32_t off, char c) { if (LDS == 1) PLP_LDS[s.cur + (uint32_t)off] = c; else s.g[off] = c; }

// what the walk needs of a read before it knows whether any lane's column is covered: asked for one read ahead, so that the answer
// is there when the walk gets to it (five dependent load latencies per read otherwise -- the walker's time is latency, not issue)
struct WalkHdr { uint32_t info, b8; int pos, end, lq; };
__device__ __forceinline__ WalkHdr walk_hdr(const StaReadsDev &R, int64_t r)
{
    WalkHdr h; h.info = R.info[r]; h.pos = R.pos[r]; h.end = R.end[r]; h.lq = R.l_qseq[r]; h.b8 = R.base_off8[r];
    return h;
}

template <bool WRITE, int LDS>
__device__ __forceinline__ void file_walk_all(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int p, bool active,
                                              int64_t rlo, int64_t rhi, const int (&kinds)[GEN_NX], int nx, AccX &acc,
                                              Sink<LDS> &sq, Sink<LDS> &qs, Sink<LDS> (&xs)[GEN_NX])
{
    uint32_t nw = 0;
    WalkHdr ahead; ahead.info = 0; ahead.b8 = 0; ahead.pos = 0; ahead.end = 0; ahead.lq = 0;
    if (rlo < rhi) ahead = walk_hdr(R, rlo);
    for (int64_t r = rlo; r < rhi; ++r) {
        const WalkHdr h = ahead;
        if (r + 1 < rhi) ahead = walk_hdr(R, r + 1);
        uint32_t info = h.info;
        if (!(info & RI_KEEP)) continue;
        int rpos = h.pos, rend = h.end;
        bool cov = active && rpos <= p && p < rend;
        if (__ballot(cov) == 0) continue;
        if (!cov) continue;
        Entry e;
        e.r = r; e.rpos = rpos; e.rend = rend; e.info = info;
        e.lq = h.lq;
        e.boff = (uint64_t)h.b8 << 3;
        if (info & RI_SIMPLE) { e.rs.qpos = p - rpos; e.rs.indel = 0; e.rs.k = 0; e.rs.is_del = false; e.rs.is_refskip = false; }
        else e.rs = resolve_general(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), rpos, p);
        if (!WRITE) acc.n_plp++;
        int c = e.rs.is_del ? placeholder_qual(R, r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
        if (c < P.min_baseQ) continue;
        if (!WRITE) {
            acc.cnt++;
            acc.seq_len += (uint32_t)token_len(R, P, e, p);
#pragma unroll
            for (int k = 0; k < GEN_NX; ++k) if (k < nx) acc.xlen[k] += (uint32_t)extra_len(R, W, P, kinds[k], e);
        } else {
            token_write<LDS>(R, W, P, e, p, sq);
            qs.put((char)(c + 33 < 126 ? c + 33 : 126));
#pragma unroll
            for (int k = 0; k < GEN_NX; ++k) if (k < nx) {
                const int kd = kinds[k];
                if (nw > 0 && kd != STA_MPLP_PRINT_MAPQ_CHAR) xs[k].put(kd >= TAGKIND ? (char)P.tag_sep : ',');
                extra_write<LDS>(R, W, P, kd, e, xs[k]);
            }
        }
        nw++;
    }
}

// the extra columns in output order: the flag columns by ascending bit, then the aux-tag columns (bam_plcmd.c:727-852)
__device__ __forceinline__ int gen_kinds(const MplpDevPar &P, int (&kinds)[GEN_NX])
{
    uint32_t ex = (uint32_t)P.flag & EXTRA_MASK;
    const int nfl = __popc(ex);
#pragma unroll
    for (int k = 0; k < GEN_NX; ++k) {
        kinds[k] = k < nfl ? (int)(ex & (~ex + 1)) : TAGKIND + (k - nfl);
        ex &= ex - 1;
    }
    return nfl + P.n_tags;
}

// k_mplp_len for windows whose rows the single-walk emit writes (at most GEN_NX extra columns): the same line lengths, and what the
// measuring walk learnt about every string of every row is KEPT -- colinfo[file][column] = (entries, base-string bytes),
// xlen[file][extra column][column] = bytes of that column's fields -- so that k_mplp_emit goes straight to its writing walk
// (it used to repeat this walk: 2.1 of its 8.9 ms on mpileup30_B_sOx, profiles/r05_generic_walker.md).
__global__ void __launch_bounds__(256) k_mplp_len_x(StaWinDev W, MplpDevPar P, uint32_t *line_len, uint2 *__restrict__ colinfo, uint32_t *__restrict__ xlen)
{
    int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    int lane = threadIdx.x & 63;
    int64_t ncols = (int64_t)W.col_end - W.col_beg;
    int64_t c0 = (int64_t)wave * 64;
    if (c0 >= ncols) return;
    int p0 = W.col_beg + (int)c0;
    int p = p0 + lane;
    bool active = p < W.col_end;
    int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
    int64_t apos = W.origin + p;
    int kinds[GEN_NX];
    const int nx = gen_kinds(P, kinds);

    uint32_t total = 0; bool any = false;
    Sink<0> dummy; dummy.g = nullptr; dummy.cur = 0;
    Sink<0> dummies[GEN_NX];
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        AccX a; a.n_plp = a.cnt = a.seq_len = 0;
#pragma unroll
        for (int k = 0; k < GEN_NX; ++k) a.xlen[k] = 0;
        file_walk_all<false, 0>(R, W, P, p, active, rlo, rhi, kinds, nx, a, dummy, dummy, dummies);
        any |= a.n_plp > 0;
        Acc t{ a.n_plp, a.cnt, a.seq_len, 0 };
#pragma unroll
        for (int k = 0; k < GEN_NX; ++k) if (k < nx) t.extras_len += a.xlen[k];
        total += file_text_len(P, t);
        if (active) {
            colinfo[(int64_t)f * ncols + c0 + lane] = make_uint2(a.cnt, a.seq_len);
#pragma unroll
            for (int k = 0; k < GEN_NX; ++k) if (k < nx) xlen[((int64_t)f * nx + k) * ncols + c0 + lane] = a.xlen[k];
        }
    }
    bool in_reg = active && column_selected(W, apos);
    bool data = in_reg && any;
    bool exists = in_reg && (any || (P.all && apos < P.tlen));
    if (exists && W.has_bed) exists = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
    uint32_t len = 0;
    if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1 + 1 + total + 1;
    if (active) line_len[c0 + lane] = len | (data ? 0x80000000u : 0u);     // rows / data columns are counted by k_col_stats
}

template <int LDS>
__device__ __forceinline__ void emit_column_1walk(const StaWinDev &W, const MplpDevPar &P, int p0, int plast, int p, bool exists, Sink<LDS> &s, int diag,
                                                  const uint2 *__restrict__ colinfo, const uint32_t *__restrict__ xlen /* k_mplp_len_x's; NULL: measure here */)
{
    int64_t apos = W.origin + p;
    if (exists) {
        for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
        s.put('\t');
        s.put_dec(apos + 1);
        s.put('\t');
        s.put((W.ref && apos < W.ref_len) ? W.ref[apos] : 'N');
    }
    int kinds[GEN_NX];
    const int nx = gen_kinds(P, kinds);
    const int64_t ncols = (int64_t)W.col_end - W.col_beg, col = (int64_t)p - W.col_beg;
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        AccX a; a.n_plp = a.cnt = a.seq_len = 0;
#pragma unroll
        for (int k = 0; k < GEN_NX; ++k) a.xlen[k] = 0;
        Sink<LDS> xs[GEN_NX];
        if (xlen) {
            if (exists) {
                const uint2 ci = colinfo[(int64_t)f * ncols + col];
                a.cnt = ci.x; a.seq_len = ci.y;
#pragma unroll
                for (int k = 0; k < GEN_NX; ++k) if (k < nx) a.xlen[k] = xlen[((int64_t)f * nx + k) * ncols + col];
            }
        } else file_walk_all<false, LDS>(R, W, P, p, exists, rlo, rhi, kinds, nx, a, s, s, xs);
        if (exists) { s.put('\t'); s.put_dec(a.cnt); s.put('\t'); }
        s.flush();
        // where the strings of this file's part of the row start: bases | qualities | extra columns
        Sink<LDS> sq = s;
        const uint32_t sl = a.cnt ? a.seq_len : 1u, ql = a.cnt ? a.cnt : 1u;
        Sink<LDS> qs = s; sink_adv(qs, (int32_t)(sl + 1));
        Sink<LDS> end = qs; sink_adv(end, (int32_t)ql);
#pragma unroll
        for (int k = 0; k < GEN_NX; ++k) {
            xs[k] = end;
            if (k < nx) {
                sink_adv(xs[k], 1);
                const uint32_t xl = a.cnt ? a.xlen[k] + (kinds[k] != STA_MPLP_PRINT_MAPQ_CHAR ? a.cnt - 1 : 0u) : 1u;
                sink_adv(end, (int32_t)(1 + xl));
            }
        }
        if (exists) {                       // the separators, and '*' for the strings of a row without entries
            sink_poke(qs, -1, '\t');
            if (!a.cnt) { sink_poke(sq, 0, '*'); sink_poke(qs, 0, '*'); }
#pragma unroll
            for (int k = 0; k < GEN_NX; ++k) if (k < nx) {
                sink_poke(xs[k], -1, '\t');
                if (!a.cnt) sink_poke(xs[k], 0, '*');
            }
        }
        // (diag: STA_GENERIC_DIAG, timing only, wrong text: 1 = no writing walk, 3 = the writing walk without the extra columns, 4 = without its eight-byte stores)
        if (diag != 1) file_walk_all<true, LDS>(R, W, P, p, exists && a.cnt, rlo, rhi, kinds, diag == 3 ? 0 : nx, a, sq, qs, xs);
        sq.flush(); qs.flush();
#pragma unroll
        for (int k = 0; k < GEN_NX; ++k) if (k < nx) xs[k].flush();
        s = end;
    }
    if (exists) s.put('\n');
    s.flush();
}

template <bool ONE_WALK>
__global__ void __launch_bounds__(256) GEN_OCC k_mplp_emit(StaWinDev W, MplpDevPar P, const uint64_t *__restrict__ offs, char *out, uint32_t lds_cap, int diag,
                                                               const uint2 *__restrict__ colinfo, const uint32_t *__restrict__ xlen)
{
    int wid = threadIdx.x >> 6;
    int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    int lane = threadIdx.x & 63;
    int64_t ncols = (int64_t)W.col_end - W.col_beg;
    int64_t c0 = (int64_t)wave * 64;
    if (c0 >= ncols) return;
    int64_t c1 = c0 + 64 < ncols ? c0 + 64 : ncols;
    int p0 = W.col_beg + (int)c0;
    int p = p0 + lane;
    bool active = p < W.col_end;
    int plast = W.col_beg + (int)c1 - 1;
    uint64_t o0 = offs[c0], o1 = offs[c1];
    uint64_t my0 = active ? offs[c0 + lane] : o1;
    uint64_t my1 = active ? offs[c0 + lane + 1] : o1;
    bool exists = my1 > my0;
    uint64_t wbytes = o1 - o0;
    if (wbytes == 0) return;
    if (wbytes <= lds_cap) {
        uint32_t slice = (lds_cap + 16 + 15) & ~15u;
        uint32_t base = (uint32_t)wid * slice;
        uint32_t mis = (uint32_t)((uintptr_t)(out + o0) & 15);
        Sink<true> s; s.g = nullptr; s.cur = base + mis + (uint32_t)(my0 - o0);
        if (ONE_WALK) emit_column_1walk<true>(W, P, p0, plast, p, exists, s, diag, colinfo, xlen); else emit_column<true>(W, P, p0, plast, p, exists, s);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // flush: LDS offset == global address (mod 16)
        char *dst = out + o0;
        uint32_t n = (uint32_t)wbytes;
        uint32_t head = mis ? 16 - mis : 0; if (head > n) head = n;
        if ((uint32_t)lane < head) dst[lane] = lds_text[base + mis + lane];
        uint32_t body = (n - head) >> 4;
        const uint4 *src4 = reinterpret_cast<const uint4 *>(lds_text + base + mis + head);
        uint4 *dst4 = reinterpret_cast<uint4 *>(dst + head);
        for (uint32_t i = lane; i < body; i += 64) dst4[i] = src4[i];
        uint32_t done = head + (body << 4);
        if (done + lane < n) dst[done + lane] = lds_text[base + mis + done + lane];
    } else {
        if (ONE_WALK) { Sink<2> s; s.open(out + my0); s.dry = diag == 4; emit_column_1walk<2>(W, P, p0, plast, p, exists, s, diag, colinfo, xlen); }
        else { Sink<false> s; s.cur = 0; s.g = out + my0; emit_column<false>(W, P, p0, plast, p, exists, s); }
    }
}


// ================================================================================================
// Helpers of the fast kernels (windows without --output-extra / -O / -s columns): the tile kernels further down and the read-major
// deep kernel.  (Round 2's lane-per-column pair k_mplp_len_fast / k_mplp_emit_fast was retired in round 3: profiles/r03_tile_kernel_counters.md.)

// Pointers that reach a kernel through W.files[] (a struct read from memory) are "generic" to the compiler, which then
// emits flat_load (slower, and it couples vmcnt with lgkmcnt).  They always point to HBM: say so.
#define GPTR(T, p) ((const __attribute__((address_space(1))) T *)(p))

// Row offset of column c.  After k_mplp_len_rm the per-column offsets are relative to their measuring tile (LEN_TC columns) and
// tbase[tile] holds the tile's place in the text (k_tile_scan); tbase == nullptr: absolute offsets (the scan launches).
__device__ __forceinline__ uint64_t row_off(const uint64_t *__restrict__ offs, const uint64_t *__restrict__ tbase, int64_t c)
{
    return tbase ? offs[c] + tbase[c >> 10] : offs[c];
}
static_assert(LEN_TC == 1024, "row_off() shifts by the measuring tile's width");

__device__ __forceinline__ int rl_i(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }

// "=ACMGRSV" / "TWYHKDBN" with '.' in place of '=' (a base equal to the reference, or '=' in the read)
__device__ __forceinline__ char base_char_fast(int c, bool rev)
{
    const unsigned long long lo = 0x565352474D43412EULL, hi = 0x4E42444B48595754ULL;
    unsigned long long t = (c & 8) ? hi : lo;
    int ch = (int)((t >> ((c & 7) << 3)) & 0xff);
    if (rev) ch = c == 0 ? ',' : (ch | 0x20);
    return (char)ch;
}

// ------------------------------------------------------------------------------------------------
// Deep columns: the read-major emit kernel.  The tile kernel gives every column a lane and needs the wave's 64 rows in LDS
// (40 KB at 300x: one wave per SIMD, and 428 sequential read steps per wave).  Here a wave takes a strip of SIXTEEN consecutive
// columns and its lanes are the READS: 64 reads at a time, each lane works out what its read shows in each of the columns, the
// token offsets inside a base string are prefix counts over the lanes (ballots; a shuffle scan when a read with indels takes
// part), and every lane stores its few bytes straight into the row in global memory -- neighbouring lanes write neighbouring
// bytes, no LDS, full occupancy, ~5 read blocks per strip at 300x instead of 428 read steps per 64 columns.  The sixteen quality
// bytes and the packed bases a plain read shows in the strip come from two vector loads.  The fixed parts of a row (name,
// position, reference base, the per-file counts, separators, '*' placeholders) are written by lane k for column k.  The
// (count, base-string bytes) of every column and file come from the measuring pass, as for k_mplp_emit_tile.
// ------------------------------------------------------------------------------------------------
// Extra columns on the read-major path (round 6).  -O / --output-BP-5 / --output-extra print, per file and row, one
// more string per requested item with one field per entry that passed -Q (bam_plcmd.c:727-855).  Every field is either a constant of
// its read (name, flag, contig, position, mapping quality, mate contig / position, read length, an aux tag's text) or the decimal of a
// query position.  So the measuring pass counts a string's bytes the way it counts depth -- a difference mark of the field's length at
// a read's first column, the opposite mark behind its last, +-1 where a query position gains or loses a digit, and a point correction
// where an entry fails -Q -- and the read-major emit kernel (lanes = reads) places the fields with a prefix sum over the passing lanes.
// The -s column stays MplpDevPar.mq_col; --output-mods and rows with more than XF_NX such columns keep the generic walkers.
#define XF_NX 8
struct XfArgs { uint32_t *xlen; int nx; int kinds[XF_NX]; };       // xlen[file][x][column]: bytes of the fields of extra column x (separators not counted)
#define XF_FLAGS (EXTRA_MASK & ~STA_MPLP_PRINT_MAPQ_CHAR)
static int xf_kinds(const sta_mplp_params &p, int (&kinds)[XF_NX])
{
    uint32_t ex = (uint32_t)p.flag & (uint32_t)XF_FLAGS;
    const int nfl = __builtin_popcount(ex), nt = p.n_tags > 0 ? p.n_tags : 0;
    for (int k = 0; k < XF_NX; ++k) { kinds[k] = k < nfl ? (int)(ex & (~ex + 1)) : TAGKIND + (k - nfl); ex &= ex - 1; }
    return nfl + nt;
}
__device__ __forceinline__ bool xf_per_entry(int kind) { return kind == STA_MPLP_PRINT_QPOS || kind == STA_MPLP_PRINT_QPOS5; }
__device__ __forceinline__ int xf_digits_ll(long long v) { return (v < 0 ? 1 : 0) + dec_digits((unsigned long long)(v < 0 ? -v : v)); }
// what a one-op read stores about extra column `kind` for the measuring steps: the field's bytes (>= 0) if they are the same in every
// column, -1: the field is query index + 1, <= -2: it is (-code - 2) - query index (--output-BP-5 on the reverse strand)
__device__ __forceinline__ int xf_read_code(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int kind, long long r, int pos, uint32_t info, int lq)
{
    if (kind == STA_MPLP_PRINT_QPOS) return -1;
    if (kind == STA_MPLP_PRINT_QPOS5) return (info & RI_REV) ? -(lq + 2) : -1;
    Entry e; e.r = r; e.rpos = pos; e.rend = pos; e.info = info; e.lq = lq; e.boff = 0;
    e.rs.qpos = 0; e.rs.indel = 0; e.rs.k = 0; e.rs.is_del = false; e.rs.is_refskip = false;
    return extra_len(R, W, P, kind, e);
}
__device__ __forceinline__ int xf_code_len(int code, int qpos) { return code >= 0 ? code : dec_digits_u32((uint32_t)(code == -1 ? qpos + 1 : (-code - 2) - qpos)); }


// one field of an extra column as the emit kernel holds it: up to sixteen bytes in a register pair (first byte lowest); longer text stays
// in memory (src), a longer number in `num`
struct XfField { uint64_t lo, hi; long long num; const char *src; int L; bool is_num; };
__device__ __forceinline__ void xf_num_field(long long v, XfField &f)
{
    f.is_num = true; f.num = v; f.src = nullptr;
    if (v >= 0 && v < 100000000ll) {
        uint32_t w = (uint32_t)v; const int n = dec_digits_u32(w);
        uint64_t t = 0;
        for (int i = 0; i < n; ++i) { const uint32_t d = w / 10u; t = (t << 8) | (uint64_t)('0' + (w - d * 10u)); w = d; }
        f.lo = t; f.hi = 0; f.L = n;
        return;
    }
    const bool neg = v < 0;
    unsigned long long u = (unsigned long long)(neg ? -v : v);
    const int nd = dec_digits(u);
    f.L = nd + (neg ? 1 : 0); f.lo = 0; f.hi = 0;
    if (f.L > 16) return;
    for (int i = 0; i < nd; ++i) { const unsigned long long d = u / 10u; f.hi = (f.hi << 8) | (f.lo >> 56); f.lo = (f.lo << 8) | (uint64_t)('0' + (u - d * 10u)); u = d; }
    if (neg) { f.hi = (f.hi << 8) | (f.lo >> 56); f.lo = (f.lo << 8) | (uint64_t)'-'; }
}
// the first sixteen bytes of l bytes of text at src (the pool ends at pool_end: nothing behind it is read)
__device__ __forceinline__ void xf_text_field(const char *src, int l, const char *pool_end, XfField &f)
{
    f.is_num = false; f.num = 0; f.src = src; f.L = l; f.lo = 0; f.hi = 0;
    if (src + 16 <= pool_end) { f.lo = *reinterpret_cast<const sink_u64u *>(src); f.hi = *reinterpret_cast<const sink_u64u *>(src + 8); }
    else for (int i = 0; i < l && i < 16; ++i) { const uint64_t b = (uint64_t)(unsigned char)src[i]; if (i < 8) f.lo |= b << (8 * i); else f.hi |= b << (8 * (i - 8)); }
}
// a field whose text does not depend on the column (bam_plcmd.c:749-852)
__device__ __forceinline__ void xf_read_field(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int kind, long long r, int rpos, uint32_t info, int lq, XfField &f)
{
    if (kind == STA_MPLP_PRINT_RNEXT || kind >= TAGKIND) {
        const uint32_t *o = R.xcol_off + (uint64_t)r * (uint64_t)R.n_xcols + (uint64_t)xcol_index(P, kind);
        const uint32_t o0 = o[0], o1 = o[1];
        xf_text_field(R.xcol_text + o0, (int)(o1 - o0), R.xcol_text + R.xcol_off[(uint64_t)R.n * (uint64_t)R.n_xcols], f);
    } else if (kind == STA_MPLP_PRINT_QNAME) {
        const uint32_t n0 = R.name_off[r], n1 = R.name_off[r + 1];
        xf_text_field(R.names + n0, (int)(n1 - n0) - 1, R.names + R.name_off[R.n], f);
    } else if (kind == STA_MPLP_PRINT_RNAME) {
        xf_text_field(W.tname, W.tname_len, W.tname + W.tname_len, f);
    } else {
        Entry e; e.r = r; e.rpos = rpos; e.rend = rpos; e.info = info; e.lq = lq; e.boff = 0;
        e.rs.qpos = 0; e.rs.indel = 0; e.rs.k = 0; e.rs.is_del = false; e.rs.is_refskip = false;
        xf_num_field(extra_value(R, W, kind, e), f);
    }
}
// the field's L bytes at dst and nothing else: two overlapping stores of the widest size that fits
__device__ __forceinline__ void xf_store(char *dst, const XfField &f)
{
    const int L = f.L;
    if (L > 16) {
        if (f.is_num) { Sink<0> sk; sk.cur = 0; sk.g = dst; sk.put_dec(f.num); return; }
        for (int t = 0; t + 8 <= L; t += 8) *reinterpret_cast<sink_u64u *>(dst + t) = *reinterpret_cast<const sink_u64u *>(f.src + t);
        if (L & 7) *reinterpret_cast<sink_u64u *>(dst + L - 8) = *reinterpret_cast<const sink_u64u *>(f.src + L - 8);
    } else if (L >= 8) {
        *reinterpret_cast<sink_u64u *>(dst) = f.lo;
        if (L > 8) { const int sh = L - 8; *reinterpret_cast<sink_u64u *>(dst + sh) = sh == 8 ? f.hi : (f.lo >> (8 * sh)) | (f.hi << (64 - 8 * sh)); }
    } else if (L >= 4) {
        *reinterpret_cast<text_u32u *>(dst) = (uint32_t)f.lo;
        if (L > 4) *reinterpret_cast<text_u32u *>(dst + L - 4) = (uint32_t)(f.lo >> (8 * (L - 4)));
    } else if (L >= 2) {
        *reinterpret_cast<text_u16u *>(dst) = (uint16_t)f.lo;
        if (L == 3) dst[2] = (char)(f.lo >> 16);
    } else if (L == 1) dst[0] = (char)f.lo;
}

// the steps of the extra columns over the sixteen columns of a strip, the column index a compile-time constant of each
template <class F> __device__ __forceinline__ void xf_for_columns(F &&f)
{
    f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>()); f(std::integral_constant<int, 3>());
    f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>()); f(std::integral_constant<int, 6>()); f(std::integral_constant<int, 7>());
    f(std::integral_constant<int, 8>()); f(std::integral_constant<int, 9>()); f(std::integral_constant<int, 10>()); f(std::integral_constant<int, 11>());
    f(std::integral_constant<int, 12>()); f(std::integral_constant<int, 13>()); f(std::integral_constant<int, 14>()); f(std::integral_constant<int, 15>());
}
// v_writelane_b32: lane K of `old` takes the wave-uniform `val` (the CPU harness: a per-lane select -- every lane holds the uniform value)
template <int K> __device__ __forceinline__ int xf_writelane(int val, int old)
{
#if defined(HIPEMU)
    return (int)(threadIdx.x & 63) == K ? val : old;
#else
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(K));
    return old;
#endif
}
// OR of the low sixteen bits over the wave (which strip columns have entries from any lane)
__device__ __forceinline__ uint32_t xf_wave_or16(uint32_t v)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r |= __ballot((v >> k) & 1u) ? 1u << k : 0u;
    return r;
}

#define DEEP_STRIP 16
#include "deep_strip.h"

// the rare (read, strip) pairs that hold an indel / clip boundary / skip go through the general CIGAR resolution
__device__ __forceinline__ Entry deep_entry(const StaReadsDev &R, int64_t r, int rpos, int rend, uint32_t info, int lq, uint64_t boff, int p)
{
    Entry e; e.r = r; e.rpos = rpos; e.rend = rend; e.info = info; e.lq = lq; e.boff = boff;
    e.rs = resolve_general(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), rpos, p);
    return e;
}

// what a lane needs to know about its read before it can look at anything else of it
struct DeepPre { uint32_t info, b8, c0, c1; int pos, end, lq; };
__device__ __forceinline__ DeepPre deep_pre(const StaReadsDev &R, int64_t r, int64_t rhi)
{
    DeepPre d; d.info = 0; d.b8 = 0; d.c0 = 0; d.c1 = 0; d.pos = 0; d.end = 0; d.lq = 0;
    if (r < rhi) { d.info = R.info[r]; d.pos = R.pos[r]; d.end = R.end[r]; d.b8 = R.base_off8[r]; d.c0 = R.cig_off[r]; d.c1 = R.cig_off[r + 1]; d.lq = R.l_qseq[r]; }
    return d;
}

// [first, end) of the reads a strip has to look at, per (file, strip): found once by a thread here instead of by every wave of
// k_mplp_emit_deep in eight dependent probe rounds
__global__ void __launch_bounds__(256) k_mplp_strip_ranges(StaWinDev W, int64_t *__restrict__ rng, int64_t nstrips)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstrips * W.nfiles) return;
    const int f = (int)(i / nstrips); const int64_t sidx = i - (int64_t)f * nstrips;
    const StaReadsDev &R = W.files[f];
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = sidx * DEEP_STRIP;
    const int p0 = W.col_beg + (int)c0;
    const int plast = p0 + (int)(ncols - c0 < DEEP_STRIP ? ncols - c0 : DEEP_STRIP) - 1;
    int64_t lo = 0, hi = R.n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (R.maxend[mid] > p0) hi = mid; else lo = mid + 1; }
    const int64_t rlo = lo;
    hi = R.n;                                                      // pos is sorted and rhi >= rlo
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (R.pos[mid] > plast) hi = mid; else lo = mid + 1; }
    rng[2 * i] = rlo; rng[2 * i + 1] = lo;
}

// XCD-aware block -> tile mapping.  Workgroup b runs on XCD b % 8 (observed; MI355X_MICROARCH.md "Workgroup dispatch"), each XCD has its
// own 4 MiB L2, and consecutive 64-column groups read the same reads (a 150-bp read spans 3.3 of them): with the identity mapping every
// group's re-reads miss the L2 of the XCD it lands on.  With xcd_tile(b, n) XCD x owns the contiguous run of tiles [x * per, (x + 1) * per),
// dispatched in order.  n8 = the launched grid (a multiple of 8) or 0 for the identity mapping (STA_XCD_MAP=0).  Placement is for speed
// only: nothing relies on it.
__device__ __forceinline__ int64_t xcd_tile(unsigned b, unsigned n8) { return n8 ? (int64_t)(b & 7u) * (n8 >> 3) + (b >> 3) : (int64_t)b; }
static bool xcd_map_on() { const char *e = getenv("STA_XCD_MAP"); return !(e && atoi(e) == 0); }      // (per launch: the tests switch it inside one process)
static unsigned xcd_grid(int64_t nblocks) { return xcd_map_on() ? (unsigned)((nblocks + 7) / 8 * 8) : (unsigned)nblocks; }

// XF: the window prints extra columns (XfArgs above): after a block's bases and qualities are placed, every extra column's fields of the
// block are -- per column of the strip one prefix sum of (field bytes + separator) over the lanes whose entry passed -Q.
#ifndef XF_OCC
#define XF_OCC 3            // waves per SIMD the extra-column form is compiled for (A/B: -DXF_OCC=3 / 4)
#endif
template <bool XF>
__global__ void __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(XF ? XF_OCC : 4, 8))) k_mplp_emit_deep(StaWinDev W, MplpDevPar P, const uint64_t *__restrict__ offs, const uint2 *__restrict__ colinfo,
                                                        const int64_t *__restrict__ rng, char *out, uint32_t only_above, const uint64_t *__restrict__ tbase, unsigned n8, XfArgs X)
{
    const int lane = threadIdx.x & 63;
    // (readfirstlane: the compiler cannot see that the strip index is the same for the 64 lanes; with it the strip's bounds,
    // row offsets and cursors live in scalar registers and the row stores take a scalar base + 32-bit offset)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wave = xcd_tile(blockIdx.x, n8) * 4 + wv;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = wave * DEEP_STRIP;
    __shared__ uint32_t s_off[4][64][DEEP_STRIP];
    __shared__ uint8_t s_qc[4][64][DEEP_STRIP];
    __shared__ uint8_t s_list[4][64];
    __shared__ char s_chr[32];                                     // code -> character: forward strand, then reverse strand
    __shared__ uint32_t s_qp[XF ? 4 : 1][XF ? 64 : 1][DEEP_STRIP]; // XF: query index (bit 31: placeholder) of the entries that went through the general CIGAR resolution
    __shared__ __attribute__((aligned(16))) uint32_t s_xcur[XF ? 4 : 1][XF ? XF_NX : 1][DEEP_STRIP];      // XF: where the next field of (extra column, strip column) goes
    __shared__ uint32_t s_dec[XF ? 1000 : 1];                      // XF: the decimal of 0 .. 999: its characters, first in the lowest byte, and their number in the top byte
    if (XF) for (uint32_t i = threadIdx.x; i < 1000u; i += 256u) {
        const uint32_t h = i / 100u, t = (i / 10u) % 10u, o = i % 10u;
        s_dec[i] = i >= 100u ? (3u << 24) | ('0' + h) | (('0' + t) << 8) | (('0' + o) << 16) : i >= 10u ? (2u << 24) | ('0' + t) | (('0' + o) << 8) : (1u << 24) | ('0' + o);
    }
    if (threadIdx.x < 32) s_chr[threadIdx.x] = base_char_fast((int)(threadIdx.x & 15), threadIdx.x >= 16);
    __syncthreads();
    if (c0 >= ncols) return;
    if (only_above) {
        // beside k_mplp_emit_tile: only the 64-column groups whose rows did not fit that kernel's LDS slice (a deep amplicon inside
        // an ordinary window) are written here
        const int64_t g0 = c0 & ~(int64_t)63, g1 = g0 + 64 < ncols ? g0 + 64 : ncols;
        if (row_off(offs, tbase, g1) - row_off(offs, tbase, g0) <= (uint64_t)only_above) return;
    }
    uint32_t *const x_off = s_off[wv][lane]; uint8_t *const x_qc = s_qc[wv][lane];
    const int nk = ncols - c0 < DEEP_STRIP ? (int)(ncols - c0) : DEEP_STRIP;
    const int p0 = W.col_beg + (int)c0, plast = p0 + nk - 1;
    const bool ends = !P.no_ends, has_ref = W.ref != nullptr;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // everything the strip needs that does not depend on another load is asked for here, in one round trip
    const int64_t nstrips = (ncols + DEEP_STRIP - 1) / DEEP_STRIP;
    const uint64_t base0 = row_off(offs, tbase, c0);               // every offset of the strip below is relative to this row start
    const int64_t apos = W.origin + p0 + lane;
    uint64_t off_a = 0, off_b = 0; uint2 ci0 = make_uint2(0u, 0u); char rc = 'N';
    if (lane < nk) {
        off_a = row_off(offs, tbase, c0 + lane); off_b = row_off(offs, tbase, c0 + lane + 1);
        ci0 = colinfo[c0 + lane];
        if (has_ref && apos < W.ref_len) rc = W.ref[apos];
    }
    int64_t rlo0 = rng[2 * wave], rhi0 = rng[2 * wave + 1];
    char *const out0 = out + base0;

    // lane k < nk owns the fixed text of column k
    const bool my_ex = off_b > off_a; unsigned my_rb = 0;
    Sink<false> fx; fx.cur = 0; fx.g = nullptr;
    if (my_ex) {
        fx.g = out + off_a;
        for (int t = 0; t < W.tname_len; ++t) fx.put(W.tname[t]);
        fx.put('\t'); fx.put_dec(apos + 1); fx.put('\t');
        fx.put(rc);
        if (has_ref) my_rb = apos < W.ref_len ? nt16_arith((unsigned char)rc) : 15u;
    }
    const unsigned exm = (unsigned)(__ballot(my_ex) & 0xffffull);
    if (!exm) return;
    unsigned long long rbpack = 0;                                 // 4-bit reference code of column k at bits 4k
#pragma unroll
    for (int k = 0; k < DEEP_STRIP; ++k) rbpack |= (unsigned long long)((unsigned)__builtin_amdgcn_readlane((int)my_rb, k) & 15u) << (4 * k);

    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        // "\tcount\t" and where the two strings of this file start; the separators and '*' placeholders right away
        unsigned my_seq = 0, my_qual = 0, my_mqd = 0;
        if (my_ex) {
            const uint2 ci = f == 0 ? ci0 : colinfo[(int64_t)f * ncols + c0 + lane];
            const uint32_t cnt = ci.x, sl = ci.y ? ci.y : 1;
            fx.put('\t'); fx.put_dec(cnt); fx.put('\t');
            my_seq = (unsigned)(fx.g - out0);
            if (!cnt) { fx.put('*'); fx.put('\t'); fx.put('*'); if (P.mq_col) { fx.put('\t'); fx.put('*'); } }
            else {
                fx.g += sl; fx.put('\t'); my_qual = (unsigned)(fx.g - out0); fx.g += cnt;
                // -s: the mapping-quality string mirrors the quality string, one tab and `cnt` bytes further on
                if (P.mq_col) { fx.put('\t'); fx.g += cnt; my_mqd = cnt + 1; }
            }
            if (XF) {
                // "\t" + the extra column's fields and separators (or '*'), one after the other behind the strings above
                for (int x = 0; x < X.nx; ++x) {
                    fx.put('\t');
                    s_xcur[wv][x][lane] = (uint32_t)(fx.g - out0) - 1u;      // (the first unit starts ON this tab and carries one)
                    if (!cnt) fx.put('*'); else fx.g += X.xlen[((int64_t)f * X.nx + x) * ncols + c0 + lane] + (cnt - 1);
                }
            }
        }
        uint32_t seen = 0;                                         // XF: bit k = column k of the strip already holds an entry of this file
        unsigned seqcur[DEEP_STRIP], qualcur[DEEP_STRIP], mqd[DEEP_STRIP];          // wave-uniform
#pragma unroll
        for (int k = 0; k < DEEP_STRIP; ++k) {
            seqcur[k] = (unsigned)__builtin_amdgcn_readlane((int)my_seq, k); qualcur[k] = (unsigned)__builtin_amdgcn_readlane((int)my_qual, k);
            mqd[k] = P.mq_col ? (unsigned)__builtin_amdgcn_readlane((int)my_mqd, k) : 0u;
        }
        if (R.n == 0) continue;
        const int64_t rlo = f == 0 ? rlo0 : rng[2 * ((int64_t)f * nstrips + wave)], rhi = f == 0 ? rhi0 : rng[2 * ((int64_t)f * nstrips + wave) + 1];
        const auto g_qual = GPTR(uint8_t, R.qual); const auto g_seq = GPTR(uint8_t, R.seq);
        // the per-read words of the NEXT block of 64 reads are asked for while this block is worked on
        DeepPre cur = deep_pre(R, rlo + lane, rhi), nx;
        for (int64_t b0 = rlo; b0 < rhi; b0 += 64, cur = nx) {
            const int64_t r = b0 + lane;
            const bool ok = r < rhi;
            nx = deep_pre(R, r + 64, rhi);
            const uint32_t info = cur.info;
            const int rpos = cur.pos, rend = cur.end;
            const bool keep = ok && (info & RI_KEEP) && rend > p0 && rpos <= plast;
            if (!__ballot(keep)) continue;
            const uint64_t boff = (uint64_t)cur.b8 << 3;
            const bool simple = (info & RI_SIMPLE) != 0, rev = (info & RI_REV) != 0;
            // A read with indels / clips / skips is still PLAIN INSIDE THIS STRIP when the strip's columns all fall in one
            // M/=/X op and none of them carries an indel: then qpos = p - qshift exactly as for a one-op read (qshift = rpos there).
            bool fastl = keep && simple; int qshift = rpos; const int lq = cur.lq;
            const bool cplx = keep && !simple;
            if (__ballot(cplx)) {
                if (cplx) {
                    const uint32_t *cig = R.cigar + cur.c0; const int n = (int)(cur.c1 - cur.c0);
                    uint32_t cg[4];                                 // the first four ops in one round trip
#pragma unroll
                    for (int t = 0; t < 4; ++t) cg[t] = t < n ? cig[t] : 0u;
                    const int ca = p0 > rpos ? p0 : rpos, cb = plast < rend - 1 ? plast : rend - 1;
                    int x = rpos, y = 0, k = 0, op = 0, l = 0; bool found = false;
                    // the op that holds the strip's first covered column (as resolve_general)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (!found && t < n) {
                            op = (int)(cg[t] & 0xf); l = (int)(cg[t] >> 4); k = t;
                            if (cg_is_refop(op)) { if (ca < x + l) found = true; else { if (cg_is_mop(op)) y += l; x += l; } }
                            else if (cg_is_qop(op)) y += l;
                        }
                    }
                    if (!found) {
                        for (k = 4; k < n; ++k) {
                            const uint32_t c = cig[k];
                            op = c & 0xf; l = (int)(c >> 4);
                            if (cg_is_refop(op)) { if (ca < x + l) { found = true; break; } if (cg_is_mop(op)) y += l; x += l; }
                            else if (cg_is_qop(op)) y += l;
                        }
                    }
                    if (found && cg_is_mop(op) && y + l <= lq) {
                        bool quiet = cb < x + l - 1;
                        if (!quiet && cb == x + l - 1) {
                            if (k + 1 >= n) quiet = true;
                            else {
                                const uint32_t c2 = k + 1 == 1 ? cg[1] : k + 1 == 2 ? cg[2] : k + 1 == 3 ? cg[3] : cig[k + 1];
                                const int op2 = (int)(c2 & 0xf);
                                quiet = op2 != CG_D && op2 != CG_I && op2 != CG_P;
                            }
                        }
                        if (quiet) { fastl = true; qshift = x - y; }
                    }
                }
            }
            const bool slow = keep && !fastl;
            const unsigned long long sm = __ballot(slow);
            // a plain read: the strip's 16 qualities and packed bases in two vector loads (byte 0 = query index qb)
            uint32_t q4[4] = { 0, 0, 0, 0 }, s4[3] = { 0, 0, 0 }; int qb = 0;
            if (fastl) {
                qb = (p0 > rpos ? p0 : rpos) - qshift;
                const uint64_t qa = boff + (uint64_t)qb, sa = (boff >> 1) + (uint64_t)(qb >> 1);
                if (qa + 16 <= R.n_bases_total) __builtin_memcpy(q4, (const uint8_t *)g_qual + qa, 16);
                else for (int t = 0; t < 16 && qa + t < R.n_bases_total; ++t) q4[t >> 2] |= (uint32_t)g_qual[qa + t] << (8 * (t & 3));
                if (sa + 12 <= (R.n_bases_total >> 1)) __builtin_memcpy(s4, (const uint8_t *)g_seq + sa, 12);
                else for (int t = 0; t < 12 && sa + t < (R.n_bases_total >> 1); ++t) s4[t >> 2] |= (uint32_t)g_seq[sa + t] << (8 * (t & 3));
            }
            // one shift per block: column k of the strip finds its quality in byte k of qs and its base code in nibble k of nib
            // (0 where the base equals the reference) -- deep_strip.h
            uint32_t qs[4] = { 0, 0, 0, 0 }; uint64_t nib = 0;
            if (fastl) {
                const int d0 = (p0 > rpos ? p0 : rpos) - p0;
                deep_shift_quals(q4, d0, qs);
                nib = deep_shift_bases(s4, qb, d0, rbpack, has_ref);
            }
            const int chr_row = rev ? 16 : 0;
            const int mqc = (int)((info >> RI_MAPQ_SHIFT) & 0xff);
            const char mq_char = (char)(mqc > 93 ? 126 : mqc + 33);
            // the other reads: their (read, column) pairs are dealt out over the 64 lanes -- token length and quality now (LDS),
            // the text after the column loop has placed them
            const int n_pairs = sm ? __popcll(sm) * DEEP_STRIP : 0;
            if (sm) {
                if (slow) s_list[wv][__popcll(sm & lt)] = (uint8_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
                for (int j0 = 0; j0 < n_pairs; j0 += 64) {
                    const int j = j0 + lane; const bool valid = j < n_pairs;
                    const int src = valid ? (int)s_list[wv][j >> 4] : lane, k = j & (DEEP_STRIP - 1);
                    const int z_rpos = __shfl(rpos, src), z_rend = __shfl(rend, src), z_lq = __shfl(lq, src);
                    const uint32_t z_info = (uint32_t)__shfl((int)info, src);
                    const uint64_t z_boff = ((uint64_t)(uint32_t)__shfl((int)(boff >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)boff, src);
                    if (!valid) continue;
                    const int p = p0 + k;
                    int tl = 0, qc = 0;
                    if (((exm >> k) & 1u) && p >= z_rpos && p < z_rend) {
                        const Entry e = deep_entry(R, b0 + src, z_rpos, z_rend, z_info, z_lq, z_boff, p);
                        qc = e.rs.is_del ? placeholder_qual(R, e.r, e.rs.qpos, z_lq, z_boff, p) : (e.rs.qpos < z_lq ? (int)R.qual[z_boff + (uint64_t)e.rs.qpos] : 0);
                        if (qc >= P.min_baseQ) tl = token_len(R, P, e, p);
                        if (XF) s_qp[wv][src][k] = ((uint32_t)e.rs.qpos & 0x7fffffffu) | (e.rs.is_del ? 0x80000000u : 0u);
                    }
                    s_off[wv][src][k] = (uint32_t)tl; s_qc[wv][src][k] = (uint8_t)qc;   // the length now, the destination (never 0) once it is known
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            uint32_t passbits = 0;                                 // XF: bit k = this lane's entry in column k passed -Q
#pragma unroll
            for (int k = 0; k < DEEP_STRIP; ++k) {
                if (!((exm >> k) & 1u)) continue;
                const int p = p0 + k;
                const bool cov = keep && p >= rpos && p < rend;
                bool pass = false; int qc = 0; uint32_t bc = 0;
                if (cov && fastl) {
                    qc = (int)((qs[k >> 2] >> (8 * (k & 3))) & 255u);
                    pass = qc >= P.min_baseQ;
                    bc = (uint32_t)(nib >> (4 * k)) & 15u;
                }
                const bool cx = sm && cov && slow;
                int tl = 0;
                if (sm) {
                    if (cx) { tl = (int)x_off[k]; qc = x_qc[k]; pass = tl > 0; }
                }
                const unsigned long long m = __ballot(pass);
                if (!m) continue;
                if (XF) passbits |= pass ? 1u << k : 0u;
                const bool head = pass && !cx && ends && p == rpos, tail = pass && !cx && ends && p == rend - 1;
                const unsigned pm = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));   // passing lanes below this one
                unsigned excl, total;
                if (sm && __ballot(cx)) {
                    if (!cx) tl = pass ? 1 + (head ? 2 : 0) + (tail ? 1 : 0) : 0;
                    int incl = tl;
                    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
                    excl = (unsigned)(incl - tl); total = (unsigned)__builtin_amdgcn_readlane(incl, 63);
                } else {
                    const unsigned long long mh = __ballot(head), mt = __ballot(tail);
                    excl = pm + 2u * __builtin_amdgcn_mbcnt_hi((unsigned)(mh >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mh, 0u))
                              + __builtin_amdgcn_mbcnt_hi((unsigned)(mt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mt, 0u));
                    total = (unsigned)(__popcll(m) + 2 * __popcll(mh) + __popcll(mt));
                }
                if (pass) {
                    uint32_t o = seqcur[k] + excl;
                    if (!cx) {
                        if (head) { out0[o] = '^'; out0[o + 1] = mq_char; o += 2; }
                        out0[o] = s_chr[chr_row + (int)bc];
                        if (tail) out0[o + 1] = '$';
                    } else x_off[k] = o;
                    out0[qualcur[k] + pm] = (char)(qc + 33 < 126 ? qc + 33 : 126);
                    if (P.mq_col) out0[qualcur[k] + pm + mqd[k]] = mq_char;
                }
                seqcur[k] += total; qualcur[k] += (unsigned)__popcll(m);
            }
            if (XF) {
                // The extra columns of this block.  A field travels with its separator in front of it (a "unit" of w = field + 1 bytes; the row's
                // first unit starts on the tab in front of the string and carries a tab instead), so that a lane writes its unit with two
                // overlapping stores of the widest size that fits and nothing else.  Per strip column: one prefix sum of w over the passing lanes.
                // (The column index is a compile-time constant of each step -- xf_for_columns -- so that lane k of the cursor register is read and
                // written with v_readlane / v_writelane; the steps hold no wave-uniform state of their own in scalar registers.)
                const uint32_t blockm = (uint32_t)__builtin_amdgcn_readfirstlane((int)xf_wave_or16(passbits));      // columns of the strip that got entries from this block
                if (blockm) {
                    const uint32_t *const x_qp = s_qp[wv][lane];
                    for (int x = 0; x < X.nx; ++x) {
                        const int kind = X.kinds[x];
                        const uint32_t sep = kind >= TAGKIND ? (uint32_t)(unsigned char)P.tag_sep : (uint32_t)',';
                        int myxc = lane < DEEP_STRIP ? (int)s_xcur[wv][x][lane] : 0;         // lane k: where column k's next unit starts
                        const int myxc0 = myxc;
                        if (!xf_per_entry(kind)) {
                            // ---- the field is a constant of the read: unit, length and store class once per block ----
                            XfField fld; fld.lo = 0; fld.hi = 0; fld.num = 0; fld.src = nullptr; fld.L = 0; fld.is_num = false;
                            if (keep) xf_read_field(R, W, P, kind, r, rpos, info, lq, fld);
                            const uint32_t w = keep ? (uint32_t)fld.L + 1u : 0u;
                            // unit = separator + field: sixteen bytes in u0 .. u3 (fields of up to fifteen bytes; longer ones take the slow class)
                            const uint32_t u0n = (uint32_t)fld.lo << 8, u1 = (uint32_t)(fld.lo >> 24), u2 = (uint32_t)(fld.lo >> 56) | ((uint32_t)fld.hi << 8), u3 = (uint32_t)(fld.hi >> 24);
                            // store classes: 1 = two or three bytes, 2 = four to eight (two dwords), 3 = nine to sixteen (two qwords), 4 = longer (slow)
                            const int cls = !keep ? 0 : w <= 3 ? 1 : w <= 8 ? 2 : w <= 16 ? 3 : 4;
                            const uint32_t pA = cls == 1 ? passbits : 0u, pB = cls == 2 ? passbits : 0u, pC = cls == 3 ? passbits : 0u, pD = cls == 4 ? passbits : 0u;
                            const int hasA = __builtin_amdgcn_readfirstlane(__ballot(pA != 0) != 0 ? 1 : 0), hasB = __builtin_amdgcn_readfirstlane(__ballot(pB != 0) != 0 ? 1 : 0),
                                      hasC = __builtin_amdgcn_readfirstlane(__ballot(pC != 0) != 0 ? 1 : 0), hasD = __builtin_amdgcn_readfirstlane(__ballot(pD != 0) != 0 ? 1 : 0);
                            const uint32_t wm4 = w - 4u, sh4 = 8u * (w - 4u);
                            // class 3: bytes [w - 8, w) of the unit (w >= 9: the separator is not among them)
                            uint32_t t_lo = 0, t_hi = 0;
                            if (cls == 3) {
                                const uint32_t sh = w - 8u;         // 1 .. 8
                                const uint32_t a0 = sh < 4 ? u0n : sh < 8 ? u1 : u2, a1 = sh < 4 ? u1 : sh < 8 ? u2 : u3, a2 = sh < 4 ? u2 : sh < 8 ? u3 : 0u;
                                t_lo = __builtin_amdgcn_alignbyte(a1, a0, sh & 3u); t_hi = __builtin_amdgcn_alignbyte(a2, a1, sh & 3u);
                            }
                            const uint64_t tail64 = (uint64_t)t_lo | ((uint64_t)t_hi << 32);
                            xf_for_columns([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
                                const uint32_t pkb = passbits & (1u << k);
                                const unsigned long long m = __ballot(pkb != 0);
                                if (!m) return;
                                const uint32_t wk = pkb ? w : 0u;
                                const uint32_t incl = wave_incl_scan_u32(wk);
                                const int xck = __builtin_amdgcn_readlane(myxc, k);
                                myxc = xf_writelane<k>(xck + __builtin_amdgcn_readlane((int)incl, 63), myxc);
                                const uint32_t p = incl - wk + (uint32_t)xck;
                                const int firstl = ((seen >> k) & 1u) ? 64 : __builtin_ctzll(m);
                                const uint32_t d0 = u0n | (lane == firstl ? (uint32_t)'\t' : sep);
                                if (hasA) if (pA & (1u << k)) {
                                    *reinterpret_cast<text_u16u *>(out0 + p) = (uint16_t)d0;
                                    if (w == 3) out0[p + 2] = (char)(d0 >> 16);
                                }
                                if (hasB) if (pB & (1u << k)) {
                                    *reinterpret_cast<text_u32u *>(out0 + p) = d0;
                                    *reinterpret_cast<text_u32u *>(out0 + p + wm4) = (uint32_t)(((uint64_t)d0 | ((uint64_t)u1 << 32)) >> sh4);
                                }
                                if (hasC) if (pC & (1u << k)) {
                                    *reinterpret_cast<sink_u64u *>(out0 + p) = (uint64_t)d0 | ((uint64_t)u1 << 32);
                                    *reinterpret_cast<sink_u64u *>(out0 + p + w - 8u) = tail64;
                                }
                            });
                            if (hasD) {
                                // fields of more than fifteen bytes (long names, Z tags): the same walk over the columns once more, rolled up, for the
                                // lanes that hold one -- the separator as a byte, the field from where it lies in memory
                                int mx = myxc0;
#pragma unroll 1
                                for (int k = 0; k < DEEP_STRIP; ++k) {
                                    const uint32_t pk = (passbits >> k) & 1u;
                                    const unsigned long long m = __ballot(pk != 0);
                                    if (!m) continue;
                                    const uint32_t wk = pk * w;
                                    const uint32_t incl = wave_incl_scan_u32(wk);
                                    const int xck = __builtin_amdgcn_readlane(mx, k);
                                    const int tot = (int)rl_u(incl, 63);
                                    mx = lane == k ? xck + tot : mx;
                                    const uint32_t p = (uint32_t)xck + incl - wk;
                                    const int firstl = ((seen >> k) & 1u) ? 64 : __builtin_ctzll(m);
                                    if ((pD >> k) & 1u) {
                                        out0[p] = lane == firstl ? '\t' : (char)sep;
                                        xf_store(out0 + p + 1u, fld);
                                    }
                                }
                            }
                        } else {
                            // ---- the field is the decimal of a query position: up to three digits from the table, anything else through the general conversion ----
                            const bool bp5 = kind == STA_MPLP_PRINT_QPOS5 && rev;
                            const int vbase = bp5 ? lq - (p0 - qshift) : p0 - qshift + 1, vstep = bp5 ? -1 : 1;     // plain lanes: the value in column k is vbase + k vstep
                            int somebig = 0;
                            xf_for_columns([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
                                const uint32_t pkb = passbits & (1u << k);
                                const unsigned long long m = __ballot(pkb != 0);
                                if (!m) return;
                                int v = vbase + k * vstep;
                                if (sm) if (slow) { const uint32_t qw = x_qp[k]; const int q = (int)(qw & 0x7fffffffu); v = bp5 ? lq - q + (int)(qw >> 31) : q + 1; }
                                const bool big = pkb && (uint32_t)v >= 1000u;
                                const unsigned long long mbig = __ballot(big);
                                const uint32_t ent = s_dec[big || !pkb ? 0 : v];
                                uint32_t wk = pkb ? (ent >> 24) + 1u : 0u;
                                if (mbig) { somebig = 1; if (big) wk = (uint32_t)xf_digits_ll(v) + 1u; }
                                const uint32_t incl = wave_incl_scan_u32(wk);
                                const int xck = __builtin_amdgcn_readlane(myxc, k);
                                myxc = xf_writelane<k>(xck + __builtin_amdgcn_readlane((int)incl, 63), myxc);
                                const uint32_t p = incl - wk + (uint32_t)xck;
                                const int firstl = ((seen >> k) & 1u) ? 64 : __builtin_ctzll(m);
                                const uint32_t d0 = (ent << 8) | (lane == firstl ? (uint32_t)'\t' : sep);
                                if (pkb && !big) {
                                    if (wk == 4u) *reinterpret_cast<text_u32u *>(out0 + p) = d0;
                                    else { *reinterpret_cast<text_u16u *>(out0 + p) = (uint16_t)d0; if (wk == 3u) out0[p + 2] = (char)(d0 >> 16); }
                                }
                            });
                            if (somebig) {
                                // values beyond the table (a read of a thousand bases and more; negative ones: a reverse read without SEQ): once more, rolled up
                                int mx = myxc0;
#pragma unroll 1
                                for (int k = 0; k < DEEP_STRIP; ++k) {
                                    const uint32_t pk = (passbits >> k) & 1u;
                                    const unsigned long long m = __ballot(pk != 0);
                                    if (!m) continue;
                                    int v = vbase + k * vstep;
                                    if (slow) { const uint32_t qw = x_qp[k]; const int q = (int)(qw & 0x7fffffffu); v = bp5 ? lq - q + (int)(qw >> 31) : q + 1; }
                                    const bool big = pk && (uint32_t)v >= 1000u;
                                    const uint32_t wk = !pk ? 0u : big ? (uint32_t)xf_digits_ll(v) + 1u : (s_dec[v] >> 24) + 1u;
                                    const uint32_t incl = wave_incl_scan_u32(wk);
                                    const int xck = __builtin_amdgcn_readlane(mx, k);
                                    const int tot = (int)rl_u(incl, 63);
                                    mx = lane == k ? xck + tot : mx;
                                    const uint32_t p = (uint32_t)xck + incl - wk;
                                    const int firstl = ((seen >> k) & 1u) ? 64 : __builtin_ctzll(m);
                                    if (big) {
                                        XfField fk; xf_num_field(v, fk);
                                        out0[p] = lane == firstl ? '\t' : (char)sep;
                                        xf_store(out0 + p + 1u, fk);
                                    }
                                }
                            }
                        }
                        if (lane < DEEP_STRIP) s_xcur[wv][x][lane] = (uint32_t)myxc;
                    }
                    seen |= blockm;
                }
            }
            if (sm) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
                for (int j0 = 0; j0 < n_pairs; j0 += 64) {
                    const int j = j0 + lane; const bool valid = j < n_pairs;
                    const int src = valid ? (int)s_list[wv][j >> 4] : lane, k = j & (DEEP_STRIP - 1);
                    const int z_rpos = __shfl(rpos, src), z_rend = __shfl(rend, src), z_lq = __shfl(lq, src);
                    const uint32_t z_info = (uint32_t)__shfl((int)info, src);
                    const uint64_t z_boff = ((uint64_t)(uint32_t)__shfl((int)(boff >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)boff, src);
                    if (!valid) continue;
                    const uint32_t o = s_off[wv][src][k];
                    if (!o) continue;
                    const Entry e = deep_entry(R, b0 + src, z_rpos, z_rend, z_info, z_lq, z_boff, p0 + k);
                    Sink<false> sk; sk.cur = 0; sk.g = out0 + o;
                    token_write<false>(R, W, P, e, p0 + k, sk);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    if (my_ex) fx.put('\n');
}


// ================================================================================================
// Tile kernels (step functions and the layout: plp_tile.h) for windows without --output-extra / -O / -s columns; waves whose rows exceed
// the LDS slice are left to k_mplp_emit_deep.

// wfirst[f][w] = first read of file f that starts at or beyond column col_beg + 64 w (w = 0 .. nwaves): one thread per entry, a
// binary search each.  The tile kernels find their reads from it with one more coalesced load instead of two 64-ary searches
// (eight dependent loads at the head of every wave).
__global__ void __launch_bounds__(256) k_wave_first(StaWinDev W, uint32_t *__restrict__ wfirst, int64_t nwaves, unsigned long long *__restrict__ status, int64_t n_status)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (the measuring kernel's look-back words and ticket, cleared here: this launch runs right before it)
    for (int64_t k = i; k < n_status; k += (int64_t)gridDim.x * blockDim.x) status[k] = 0ull;
    if (i >= (nwaves + 1) * W.nfiles) return;
    const int f = (int)(i / (nwaves + 1)); const int64_t w = i - (int64_t)f * (nwaves + 1);
    const StaReadsDev &R = W.files[f];
    const int64_t key = (int64_t)W.col_beg + 64 * w;
    int64_t lo = 0, hi = R.n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)R.pos[mid] >= key) hi = mid; else lo = mid + 1; }
    wfirst[i] = (uint32_t)lo;
}

// reads that can touch columns [p0, ...) of the 64-column groups [w_lo, w_hi): a superset in file order.  rhi = first read starting
// beyond them; rlo: maxend (prefix maximum of the read ends) is monotone, so the reads before `start` that still reach p0 are a
// suffix of them -- counted on the 64 reads before `start` (one coalesced load), a search only when all 64 do.
__device__ __forceinline__ void wave_range_indexed(const StaReadsDev &R, const uint32_t *__restrict__ wf, int64_t w_lo, int64_t w_hi, int p0, int64_t &rlo, int64_t &rhi)
{
    const int lane = threadIdx.x & 63;
    const int64_t start = wf[w_lo];
    rhi = wf[w_hi];
    const int64_t i = start - 64 + lane;
    const bool gt = i >= 0 && R.maxend[i] > p0;
    const int cnt = __popcll(__ballot(gt));
    if (cnt == 64 && start > 64) rlo = wave_upper_bound(R.maxend, start - 64, p0);
    else rlo = start - cnt;
    if (rlo > rhi) rlo = rhi;
}

// len_step_b (plp_tile.h) for a window with extra columns: an entry that fails -Q also takes its fields out of the extra columns' rows
// of marks -- a point correction, i.e. the opposite mark at the next column
__device__ __forceinline__ void xf_len_step_b(LenLds &L, int t, const StaReadsDev &R, const MplpDevPar &P, int t0, int t1, int *xd, const int *xcode, int nx)
{
    if (P.min_baseQ <= 0) return;
    const uint32_t minq4 = (uint32_t)P.min_baseQ * 0x01010101u;
    const int g = t & 3;
#pragma unroll 1
    for (int sub = 0; sub < LEN_THREADS / 64; ++sub) {
        const int slot = sub * 64 + (t >> 2);
        if (L.m_kind[slot] != 1) continue;
        const int pos = L.m_pos[slot], end = L.m_end[slot];
        const uint64_t boff = (uint64_t)L.m_b8[slot] << 3;
        const int qa = (pos > t0 ? pos : t0) - pos, qe = (end < t1 ? end : t1) - pos;
        for (int j = (qa >> 4) + g; (j << 4) < qe; j += 4) {
            const int q0 = j << 4;
            uint32_t v[4] = { 0, 0, 0, 0 };
            const uint64_t a = boff + (uint64_t)q0;
            if (a + 16 <= R.n_bases_total) __builtin_memcpy(v, R.qual + a, 16);
            else for (int i = 0; i < 16 && a + i < R.n_bases_total; ++i) v[i >> 2] |= (uint32_t)R.qual[a + i] << (8 * (i & 3));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w0 = q0 + 4 * k;
                uint32_t f = ~swar_ge_u8(v[k], minq4) & swar_byte_range(qa - w0, qe - w0) & 0x80808080u;
                while (f) {
                    const int i = (__builtin_ctz(f)) >> 3;
                    const int col = pos + w0 + i - t0;
                    atomicAdd(&L.fail[col], 1);
                    for (int x = 0; x < nx; ++x) {
                        const int l = xf_code_len(xcode[x * LEN_THREADS + slot], w0 + i);
                        atomicAdd(&xd[x * (LEN_TC + 4) + col], -l);
                        atomicAdd(&xd[x * (LEN_TC + 4) + col + 1], l);
                    }
                    f &= f - 1;
                }
            }
        }
    }
}
// len_step_c for such a window: a read with a general CIGAR adds its fields entry by entry
__device__ __forceinline__ void xf_len_step_c(LenLds &L, int gi, int lane, int nlanes, long long b0, const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int t0, int t1,
                                              int *xd, const XfArgs &X)
{
    const int slot = L.glist[gi];
    const long long r = b0 + slot;
    const int pos = L.m_pos[slot], end = L.m_end[slot];
    const int ca = pos > t0 ? pos : t0, cb = end < t1 ? end : t1;
    Entry e;
    e.r = r; e.rpos = pos; e.rend = end; e.info = R.info[r]; e.lq = R.l_qseq[r];
    e.boff = (uint64_t)L.m_b8[slot] << 3;
    const uint32_t *cig = R.cigar + R.cig_off[r];
    const int n = (int)(R.cig_off[r + 1] - R.cig_off[r]);
    for (int p = ca + lane; p < cb; p += nlanes) {
        e.rs = resolve_general(cig, n, pos, p);
        const int c = e.rs.is_del ? placeholder_qual(R, r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? (int)R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
        if (c < P.min_baseQ) atomicAdd(&L.fail[p - t0], 1);
        else {
            const int tl = token_len(R, P, e, p);
            if (tl != 1) atomicAdd(&L.extra[p - t0], tl - 1);
            for (int x = 0; x < X.nx; ++x) {
                const int l = extra_len(R, W, P, X.kinds[x], e);
                atomicAdd(&xd[x * (LEN_TC + 4) + p - t0], l);
                atomicAdd(&xd[x * (LEN_TC + 4) + p - t0 + 1], -l);
            }
        }
    }
}

template <bool XF>
__global__ void __attribute__((amdgpu_flat_work_group_size(LEN_THREADS, LEN_THREADS), amdgpu_waves_per_eu(XF ? 4 : 6, 8))) k_mplp_len_rm(StaWinDev W, MplpDevPar P, uint32_t *line_len, uint2 *colinfo, const uint32_t *__restrict__ wfirst,
                                                                              unsigned long long *__restrict__ status, uint64_t *__restrict__ offs, StaCounters *ctr, int maxcnt, XfArgs X)
{
    __shared__ LenLds L;
    // XF: per extra column a row of difference marks over the tile and, per batch, what every one-op read stores about it (dynamic LDS)
    int *const xd = reinterpret_cast<int *>(lds_text);                  // [nx][LEN_TC + 4]
    int *const xcode = xd + (XF ? X.nx : 0) * (LEN_TC + 4);             // [nx][LEN_THREADS]
    const int nx = XF ? X.nx : 0;
    const int t = threadIdx.x;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    if (t == 0) { L.n_lines = 0; L.n_data = 0; L.wave_max = 0; }
    const unsigned tile = blockIdx.x;
    const int64_t c0 = (int64_t)tile * LEN_TC;
    if (c0 >= ncols) return;
    const int t0 = W.col_beg + (int)c0;
    const int ntile = (int)(ncols - c0 < LEN_TC ? ncols - c0 : LEN_TC);
    const int t1 = t0 + ntile;
    uint32_t total[4] = { 0, 0, 0, 0 };
    bool any[4] = { false, false, false, false };
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        len_clear(L, t);
        if (XF) for (int i = t; i < nx * (LEN_TC + 4); i += LEN_THREADS) xd[i] = 0;
        if (t < 64) {                                       // the first wave finds the tile's reads
            int64_t rlo, rhi;
            const int64_t nwaves = (ncols + 63) >> 6, w_lo = c0 >> 6, w_hi = w_lo + (LEN_TC >> 6) < nwaves ? w_lo + (LEN_TC >> 6) : nwaves;
            wave_range_indexed(R, wfirst + (int64_t)f * (nwaves + 1), w_lo, w_hi, t0, rlo, rhi);
            if (t == 0) { L.rlo = rlo; L.rhi = rhi; }
        } else if (t < 128 && maxcnt > 0) {
            // the second wave, meanwhile: the -d detector (kernels_maxcnt.hip) for the reads that START in this tile (the first tile
            // also takes the reads in front of the window, the last one those behind it).  A read can be dropped only if maxcnt - 1
            // kept reads in front of it still reach its start: maxend is non-decreasing, so that is one look at maxend[i - maxcnt + 1].
            const int64_t nwaves = (ncols + 63) >> 6, w_lo = c0 >> 6, w_hi = w_lo + (LEN_TC >> 6) < nwaves ? w_lo + (LEN_TC >> 6) : nwaves;
            const uint32_t *wf = wfirst + (int64_t)f * (nwaves + 1);
            const int64_t a = tile == 0 ? 0 : (int64_t)wf[w_lo], b = w_hi == nwaves ? R.n : (int64_t)wf[w_hi];
            bool hit = false;
            for (int64_t i = (a > maxcnt - 1 ? a : maxcnt - 1) + (t - 64); i < b; i += 64)
                hit = hit || ((R.info[i] & RI_KEEP) && R.maxend[i - maxcnt + 1] > R.pos[i] - 1);
            if (__ballot(hit) && t == 64) atomicAdd(&ctr->maxcnt_flag, 1ull);
        }
        __syncthreads();
        const long long rlo = L.rlo, rhi = L.rhi;
        for (long long b0 = rlo; b0 < rhi; b0 += LEN_THREADS) {
            len_step_a(L, t, R, P, t0, t1, b0);
            if (XF) {
                // thread t's read, if it is a one-op read in the tile: the marks of every extra column.  (--output-BP-5 on a reverse read whose
                // SEQ is not its CIGAR's length would go negative: such a read takes the per-entry route below.)
                if (L.m_kind[t] == 1) {
                    const long long r = b0 + t;
                    const int pos = L.m_pos[t], end = L.m_end[t], lq = R.l_qseq[r];
                    const uint32_t info = R.info[r];
                    bool general = false;
                    for (int x = 0; x < nx; ++x) general = general || (X.kinds[x] == STA_MPLP_PRINT_QPOS5 && (info & RI_REV) && lq != end - pos);
                    if (general) {
                        // (its '^x' / '$' bytes are counted per entry by step C: take back what step A added)
                        if (!P.no_ends) {
                            const uint64_t boff = (uint64_t)L.m_b8[t] << 3;
                            if (pos >= t0 && (int)R.qual[boff] >= P.min_baseQ) atomicAdd(&L.extra[pos - t0], -2);
                            if (end <= t1 && (int)R.qual[boff + (uint64_t)(end - 1 - pos)] >= P.min_baseQ) atomicAdd(&L.extra[end - 1 - t0], -1);
                        }
                        L.m_kind[t] = 2; L.glist[atomicAdd(&L.gcount, 1)] = t;
                    } else {
                        const int a = (pos > t0 ? pos : t0) - t0, b = (end < t1 ? end : t1) - t0;
                        for (int x = 0; x < nx; ++x) {
                            const int code = xf_read_code(R, W, P, X.kinds[x], r, pos, info, lq);
                            xcode[x * LEN_THREADS + t] = code;
                            int *const D = xd + x * (LEN_TC + 4);
                            atomicAdd(&D[a], xf_code_len(code, a + t0 - pos));
                            atomicAdd(&D[b], -xf_code_len(code, b - 1 + t0 - pos));
                            if (code < 0) {
                                // the columns at which the decimal gains (forward) / loses (reverse) a digit
                                uint32_t th = 10;
                                for (int d = 1; d < 10; ++d, th *= 10u) {
                                    const long long q = code == -1 ? (long long)th - 1 : (long long)(-code - 2) - (long long)th + 1;      // first query index on the far side
                                    const long long c = q + pos - t0;
                                    if (c > a && c < b) atomicAdd(&D[c], code == -1 ? 1 : -1);
                                    if (th >= 1000000000u) break;
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (XF) xf_len_step_b(L, t, R, P, t0, t1, xd, xcode, nx); else
            len_step_b(L, t, R, P, t0, t1);
            const int ng = L.gcount;
            if (XF) { for (int gi = t >> 6; gi < ng; gi += LEN_THREADS / 64) xf_len_step_c(L, gi, t & 63, 64, b0, R, W, P, t0, t1, xd, X); } else
            for (int gi = t >> 6; gi < ng; gi += LEN_THREADS / 64) len_step_c(L, gi, t & 63, 64, b0, R, P, t0, t1);
            if (b0 + LEN_THREADS < rhi) {                   // the batch arrays are reused
                __syncthreads();
                if (t == 0) L.gcount = 0;
                __syncthreads();
            }
        }
        __syncthreads();
        len_scan_1(L, t); __syncthreads();
        len_scan_2(L, t); __syncthreads();
        len_scan_3(L, t); __syncthreads();
        const int before = len_scan_4(L, t);
        uint32_t cnt4[4] = { 0, 0, 0, 0 };
        len_file_result(L, t, before, ntile, colinfo + (int64_t)f * ncols + c0, total, any, P.mq_col != 0, XF ? cnt4 : nullptr);
        __syncthreads();
        if (XF) {
            // every extra column: prefix sum of its marks = bytes of its fields per column; "\t" + fields + separators (or '*') joins the row
            for (int x = 0; x < nx; ++x) {
                const int *const D = xd + x * (LEN_TC + 4);
                L.part[t] = D[4 * t] + D[4 * t + 1] + D[4 * t + 2] + D[4 * t + 3]; __syncthreads();
                len_scan_2(L, t); __syncthreads();
                len_scan_3(L, t); __syncthreads();
                int v = len_scan_4(L, t);
                uint32_t *const xl = X.xlen + ((int64_t)f * nx + x) * ncols + c0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * t + i;
                    v += D[c];
                    if (c >= ntile) continue;
                    xl[c] = (uint32_t)v;
                    total[i] += 1 + (cnt4[i] ? (uint32_t)v + (cnt4[i] - 1) : 1u);
                }
                __syncthreads();
            }
        }
    }
    // row lengths of the thread's four columns and their exclusive scan inside the tile; the tile's bytes / rows / largest wave go to
    // k_tile_scan (one small workgroup), which replaces the whole-window scan and column-statistics launches of the lane-per-column
    // pair.  (A decoupled look-back inside this kernel was measured first: the ticket + wait cost what the two launches had cost.)
    uint32_t len4[4] = { 0, 0, 0, 0 };
    unsigned my_lines = 0, my_data = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * t + i;
        if (c >= ntile) continue;
        const int64_t apos = W.origin + t0 + c;
        const bool in_reg = column_selected(W, apos);
        const bool data = in_reg && any[i];
        bool exists = in_reg && (any[i] || (P.all && apos < P.tlen));
        if (exists && W.has_bed) exists = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
        uint32_t len = 0;
        if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1 + 1 + total[i] + 1;
        line_len[c0 + c] = len | (data ? 0x80000000u : 0u);
        len4[i] = len; my_lines += len != 0; my_data += data ? 1u : 0u;
    }
    L.part[t] = (int)(len4[0] + len4[1] + len4[2] + len4[3]);
    if (my_lines) atomicAdd(&L.n_lines, my_lines);
    if (my_data) atomicAdd(&L.n_data, my_data);
    __syncthreads();
    len_scan_2(L, t);
    if ((t & 15) == 0) {                                   // bytes of one wave's 64 rows (what the emit kernel stages in LDS)
        unsigned long long wb = 0;
        for (int k = 0; k < 16; ++k) wb += (unsigned)L.part[t + k];
        atomicMax(&L.wave_max, wb);
    }
    __syncthreads();
    len_scan_3(L, t); __syncthreads();
    const uint64_t before = (uint64_t)(unsigned)len_scan_4(L, t);
    if (t == LEN_THREADS - 1) L.tile_bytes = before + (unsigned)L.part[t];
    __syncthreads();
    // the tile's aggregates for k_tile_scan: text bytes, rows << 31 | data columns, largest wave; offsets stay tile-relative
    if (t == 0) {
        unsigned long long *agg = status + 3 * (size_t)tile;
        agg[0] = L.tile_bytes; agg[1] = ((unsigned long long)L.n_lines << 31) | L.n_data; agg[2] = L.wave_max;
    }
    uint64_t o = before;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * t + i;
        if (c < ntile) offs[c0 + c] = o;
        o += len4[i];
    }
    // (the entry behind the last column: the end of the last tile, or -- a window of whole tiles -- the start of the tile behind it)
    if (t == LEN_THREADS - 1 && c0 + ntile == ncols) offs[ncols] = ntile == LEN_TC ? 0 : L.tile_bytes;
}

// exclusive scan of the measuring tiles' text bytes -> tbase[0 .. ntiles] (tbase[ntiles] = the window's text bytes), and the window's
// totals.  One workgroup: a window has a few thousand tiles.
__global__ void __launch_bounds__(1024) k_tile_scan(const unsigned long long *__restrict__ agg, uint64_t *__restrict__ tbase, int64_t ntiles, StaCounters *ctr)
{
    __shared__ unsigned long long s_sum[1024], s_rows[1024], s_max[1024];
    const int t = threadIdx.x;
    const int64_t per = (ntiles + 1023) / 1024, a = (int64_t)t * per, b = a + per < ntiles ? a + per : ntiles;
    unsigned long long sum = 0, rows = 0, mx = 0;
    for (int64_t i = a; i < b; ++i) { sum += agg[3 * i]; rows += agg[3 * i + 1]; const unsigned long long m = agg[3 * i + 2]; mx = m > mx ? m : mx; }
    s_sum[t] = sum; s_rows[t] = rows; s_max[t] = mx;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan of the per-thread sums (and totals of the other two)
        unsigned long long v = 0, r = 0, m = s_max[t];
        if (t >= o) { v = s_sum[t - o]; r = s_rows[t - o]; const unsigned long long m2 = s_max[t - o]; m = m2 > m ? m2 : m; }
        __syncthreads();
        s_sum[t] += v; s_rows[t] += r; s_max[t] = m;
        __syncthreads();
    }
    unsigned long long run = s_sum[t] - sum;                // exclusive prefix of this thread's first tile
    for (int64_t i = a; i < b; ++i) { tbase[i] = run; run += agg[3 * i]; }
    if (t == 1023) {
        tbase[ntiles] = s_sum[1023];
        ctr->n_lines = s_rows[1023] >> 31; ctr->n_data_cols = s_rows[1023] & 0x7fffffffull; ctr->max_wave_bytes = s_max[1023];
        ctr->out_bytes = s_sum[1023];                       // (the plan reads the window's text bytes with the counters: one transfer)
    }
}

#define TILE_WAVES 1            // waves per workgroup of k_mplp_emit_tile (they share nothing; measured 1 / 2 / 4 in one box: 0.420 / 0.428 / 0.486 ms)

// DIAG (timing diagnostics only, STA_TILE_DIAG, wrong text; 0 = the product): 1 = no column walk (phase 2), 2 = neither conversion nor walk,
// 3 = no flush of the text slice -- the phase budget of profiles/r06_tile_phase_budget.md
template <int DIAG>
__global__ void __launch_bounds__(64 * TILE_WAVES) k_mplp_emit_tile(StaWinDev W, MplpDevPar P, const uint64_t *__restrict__ offs, const uint2 *__restrict__ colinfo,
                                                                    const uint32_t *__restrict__ wfirst, const uint64_t *__restrict__ tbase, char *out, uint32_t lds_cap, unsigned n8)
{
    const int wid = threadIdx.x >> 6;
    const int64_t wave = xcd_tile(blockIdx.x, n8) * TILE_WAVES + wid;
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = wave * 64;
    if (c0 >= ncols) return;
    const int64_t c1 = c0 + 64 < ncols ? c0 + 64 : ncols;
    const int p0 = W.col_beg + (int)c0;
    const int p = p0 + lane;
    const bool active = p < W.col_end;
    const int plast = W.col_beg + (int)c1 - 1;
    const uint64_t o0 = row_off(offs, tbase, c0), o1 = row_off(offs, tbase, c1);
    const uint64_t my0 = active ? row_off(offs, tbase, c0 + lane) : o1;
    const uint64_t my1 = active ? row_off(offs, tbase, c0 + lane + 1) : o1;
    const bool exists = my1 > my0;
    const uint64_t wbytes = o1 - o0;
    if (wbytes == 0 || wbytes > lds_cap) return;            // rows beyond the slice: k_mplp_emit_deep takes those columns
    const uint32_t slice = (lds_cap + 48 + 15) & ~15u;      // text (+ up to 15 alignment bytes) + dump bytes for predicated writes
    const uint32_t base = (uint32_t)wid * (slice + (uint32_t)TILE_LDS_BYTES);
    TileLds &T = *reinterpret_cast<TileLds *>(lds_text + base + slice);
    const uint32_t mis = (uint32_t)((uintptr_t)(out + o0) & 15);
    const uint32_t dump = base + slice - 8;
    const bool has_ref = W.ref != nullptr;
    const int64_t apos = W.origin + p;

    TileLane st;
    tile_row_head(T, st, lane, W, base + mis + (uint32_t)(my0 - o0), exists, apos);
    if (lane < 2 * TILE_SLOTS) tile_zero_column(T, lane);
    wave_lds_sync();
    if (lane < 4) tile_refpack(T, lane);
    wave_lds_sync();

    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        const int64_t nwaves = (ncols + 63) >> 6;
        wave_range_indexed(R, wfirst + (int64_t)f * (nwaves + 1), wave, wave + 1, p0, rlo, rhi);
        tile_file_head(st, lane, exists ? colinfo[(int64_t)f * ncols + c0 + lane] : make_uint2(0u, 0u), dump, P.mq_col != 0);
        const auto g_info = GPTR(uint32_t, R.info); const auto g_pos = GPTR(int32_t, R.pos); const auto g_end = GPTR(int32_t, R.end);
        const auto g_b8 = GPTR(uint32_t, R.base_off8);
        for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
            const int64_t ri = b0 + lane;
            const bool ok = ri < rhi;
            const uint32_t v_info = ok ? g_info[ri] : 0u;
            const int v_pos = ok ? g_pos[ri] : 0;
            const int v_end = ok ? g_end[ri] : 0;
            const uint32_t v_b8 = ok ? g_b8[ri] : 0u;
            const bool live = ok && tile_read_is_live(v_info, v_pos, v_end, p0, plast);
            const unsigned long long livem = __ballot(live);
            const int nlive = __popcll(livem);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(livem >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)livem, 0u));
            for (int first = 0; first < nlive; first += TILE_SLOTS) {
                const int ns = nlive - first < TILE_SLOTS ? nlive - first : TILE_SLOTS;
                if (live && rank >= first && rank < first + TILE_SLOTS) tile_set_slot(T, rank - first, lane, ri, v_info, v_pos, v_end, v_b8);
                wave_lds_sync();
                const bool slot_simple = DIAG == 2 ? true : tile_phase1(T, lane, ns, R, P, p0, has_ref);
                const unsigned long long sm = __ballot(slot_simple && (lane & 3) == 0);      // bit 4 s: slot s is a one-op read
                wave_lds_sync();
                if (DIAG != 1 && DIAG != 2)
                for (int s = 0; s < ns;) {
                    if (s + 4 <= ns && ((sm >> (4 * s)) & 0x1111ull) == 0x1111ull) { tile_phase2_rows4(T, s, st.col, st.cur_s, st.cur_q, st.mq_d); s += 4; continue; }
                    if ((sm >> (4 * s)) & 1ull) tile_phase2_row(T, s, st.col, st.cur_s, st.cur_q, st.mq_d);
                    else tile_phase2_mixed(T, s, st, R, W, P, b0, p, p0);     // (the read is the same for every lane)
                    ++s;
                }
                wave_lds_sync();                              // the tile rows are rewritten by the next round
            }
        }
        tile_file_tail(st);
    }
    if (exists) lds_text[st.cur] = '\n';
    wave_lds_sync();
    if (DIAG == 3) return;
    // flush: LDS offset == global address (mod 16)
    char *dst = out + o0;
    const uint32_t n = (uint32_t)wbytes;
    uint32_t head = mis ? 16 - mis : 0; if (head > n) head = n;
    if ((uint32_t)lane < head) dst[lane] = lds_text[base + mis + lane];
    const uint32_t body = (n - head) >> 4;
    const uint4 *src4 = reinterpret_cast<const uint4 *>(lds_text + base + mis + head);
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t i = lane; i < body; i += 64) dst4[i] = src4[i];
    const uint32_t done = head + (body << 4);
    if (done + lane < n) dst[done + lane] = lds_text[base + mis + done + lane];
}

// extra columns of a window the generic walkers write: their number if the single-walk emit holds cursors for them, else -1
int sta_mplp_generic_extras(const sta_mplp_params &p)
{
    const int n = __builtin_popcount((unsigned)p.flag & (unsigned)EXTRA_MASK) + (p.n_tags > 0 ? p.n_tags : 0);
    return n <= GEN_NX ? n : -1;
}

static MplpDevPar make_par(const sta_mplp_params &p, int64_t tlen)
{
    MplpDevPar d;
    d.min_baseQ = p.min_baseQ; d.all = p.all; d.rev_del = p.rev_del; d.flag = p.flag;
    d.no_ins = p.no_ins; d.no_del = p.no_del; d.no_ends = p.no_ends; d.tlen = tlen;
    d.n_tags = p.n_tags > 0 ? p.n_tags : 0; d.tag_sep = p.tag_sep ? p.tag_sep : ',';
    d.mods = (p.flag & STA_MPLP_OUTPUT_MODS) ? 1 : 0; d.no_ins_mods = (p.no_ins_mods || p.no_ins) ? 1 : 0;
    d.mq_col = ((sta_mplp_has_fast_path(p) || sta_mplp_has_xfast_path(p)) && (p.flag & STA_MPLP_PRINT_MAPQ_CHAR)) ? 1 : 0;
    return d;
}

size_t sta_mplp_len_status_bytes(int64_t ncols) { return (size_t)(4 * ((ncols + LEN_TC - 1) / LEN_TC) + 2) * 8; }      // 3 aggregates per tile + tbase[ntiles + 1]
const uint64_t *sta_mplp_tile_base(const void *status, int64_t ncols) { return (const uint64_t *)status + 3 * ((ncols + LEN_TC - 1) / LEN_TC); }

void sta_launch_wave_first(hipStream_t s, const StaWinDev &w, uint32_t *wfirst, void *status)
{
    const int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0 || w.nfiles <= 0) return;
    const int64_t nwaves = (ncols + 63) / 64, nt = (nwaves + 1) * w.nfiles;
    (void)status;
    hipLaunchKernelGGL(k_wave_first, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, w, wfirst, nwaves, (unsigned long long *)nullptr, (int64_t)0);
}

bool sta_launch_mplp_len(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, uint32_t *line_len, uint2 *colinfo, StaCounters *ctr, const uint32_t *wfirst,
                         void *status, uint64_t *offs, int detect_maxcnt, uint32_t *gen_xlen)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return false;
    const bool xf = sta_mplp_has_xfast_path(p) && gen_xlen;
    if ((sta_mplp_has_fast_path(p) || xf) && colinfo && wfirst && status && offs && sta_mplp_tile_ok(p)) {
        const int64_t ntiles = (ncols + LEN_TC - 1) / LEN_TC;
        XfArgs X; X.xlen = gen_xlen; X.nx = 0;
        for (int k = 0; k < XF_NX; ++k) X.kinds[k] = 0;
        if (xf) {
            X.nx = xf_kinds(p, X.kinds);
            hipLaunchKernelGGL(k_mplp_len_rm<true>, dim3((unsigned)ntiles), dim3(LEN_THREADS), (size_t)X.nx * (LEN_TC + 4 + LEN_THREADS) * 4, s, w, make_par(p, w.tlen), line_len, colinfo, wfirst,
                               (unsigned long long *)status, offs, ctr, detect_maxcnt, X);
        } else
        hipLaunchKernelGGL(k_mplp_len_rm<false>, dim3((unsigned)ntiles), dim3(LEN_THREADS), 0, s, w, make_par(p, w.tlen), line_len, colinfo, wfirst,
                           (unsigned long long *)status, offs, ctr, detect_maxcnt, X);
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, s, (const unsigned long long *)status, (uint64_t *)status + 3 * ntiles, ntiles, ctr);
        return true;          // offsets (tile-relative + tile bases), totals and the largest wave are done: no scan / column statistics launches
    }
    int64_t nb = (ncols + 255) / 256;
    if (gen_xlen && colinfo && sta_mplp_generic_extras(p) >= 0)
        hipLaunchKernelGGL(k_mplp_len_x, dim3((unsigned)nb), dim3(256), 0, s, w, make_par(p, w.tlen), line_len, colinfo, gen_xlen);
    else
        hipLaunchKernelGGL(k_mplp_len, dim3((unsigned)nb), dim3(256), 0, s, w, make_par(p, w.tlen), line_len, ctr);
    return false;
}

static void launch_deep(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, const uint64_t *offs, const uint2 *colinfo, char *out,
                        int64_t *strip_rng, uint32_t only_above, const uint64_t *tbase, const uint32_t *xlen = nullptr /* the measuring pass's extra-column bytes: the XF form */)
{
    const int64_t ncols = (int64_t)w.col_end - w.col_beg;
    const int64_t nwaves_d = sta_mplp_deep_strips(ncols);
    const int64_t nt = nwaves_d * w.nfiles;
    hipLaunchKernelGGL(k_mplp_strip_ranges, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, w, strip_rng, nwaves_d);
    const unsigned gd = xcd_grid((nwaves_d + 3) / 4);
    XfArgs X; X.xlen = const_cast<uint32_t *>(xlen); X.nx = 0;
    for (int k = 0; k < XF_NX; ++k) X.kinds[k] = 0;
    if (xlen) {
        X.nx = xf_kinds(p, X.kinds);
        if (getenv("STA_DEBUG")) fprintf(stderr, "[sta] extra columns on the read-major kernels: %d\n", X.nx);
        hipLaunchKernelGGL(k_mplp_emit_deep<true>, dim3(gd), dim3(256), 0, s, w, make_par(p, w.tlen), offs, colinfo, (const int64_t *)strip_rng, out, only_above, tbase, xcd_map_on() ? gd : 0u, X);
    } else
        hipLaunchKernelGGL(k_mplp_emit_deep<false>, dim3(gd), dim3(256), 0, s, w, make_par(p, w.tlen), offs, colinfo, (const int64_t *)strip_rng, out, only_above, tbase, xcd_map_on() ? gd : 0u, X);
}

// tile = true (the measuring pass was k_mplp_len_rm: colinfo, wfirst and tbase are valid): deep_mode 1 = every strip through
// k_mplp_emit_deep; otherwise k_mplp_emit_tile, and with deep_mode 2 the 64-column groups whose rows exceed tile_cap through
// k_mplp_emit_deep beside it.  tile = false: the generic walker (absolute offsets, any option set).
void sta_launch_mplp_emit(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, const uint64_t *offs, const uint2 *colinfo,
                          char *out, uint32_t lds_cap, int64_t *strip_rng, uint32_t tile_cap, int deep_mode, const uint32_t *wfirst, const uint64_t *tbase, bool tile,
                          const uint32_t *gen_xlen)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    int64_t nwaves = (ncols + 63) / 64;
    if (tile) {
        if (strip_rng && sta_mplp_has_xfast_path(p)) { launch_deep(s, w, p, offs, colinfo, out, strip_rng, 0u, tbase, gen_xlen); return; }      // extra columns: every strip, read-major
        if (strip_rng && deep_mode == 1) { launch_deep(s, w, p, offs, colinfo, out, strip_rng, 0u, tbase); return; }
        const uint32_t slice = (tile_cap + 48 + 15) & ~15u;       // must match k_mplp_emit_tile
        const size_t lds = (size_t)TILE_WAVES * (slice + TILE_LDS_BYTES);
        const unsigned gt = xcd_grid((nwaves + TILE_WAVES - 1) / TILE_WAVES);
        const char *td = getenv("STA_TILE_DIAG");
        switch (td ? atoi(td) : 0) {
        case 1: hipLaunchKernelGGL(k_mplp_emit_tile<1>, dim3(gt), dim3(64 * TILE_WAVES), lds, s, w, make_par(p, w.tlen), offs, colinfo, wfirst, tbase, out, tile_cap, xcd_map_on() ? gt : 0u); break;
        case 2: hipLaunchKernelGGL(k_mplp_emit_tile<2>, dim3(gt), dim3(64 * TILE_WAVES), lds, s, w, make_par(p, w.tlen), offs, colinfo, wfirst, tbase, out, tile_cap, xcd_map_on() ? gt : 0u); break;
        case 3: hipLaunchKernelGGL(k_mplp_emit_tile<3>, dim3(gt), dim3(64 * TILE_WAVES), lds, s, w, make_par(p, w.tlen), offs, colinfo, wfirst, tbase, out, tile_cap, xcd_map_on() ? gt : 0u); break;
        default: hipLaunchKernelGGL(k_mplp_emit_tile<0>, dim3(gt), dim3(64 * TILE_WAVES), lds, s, w, make_par(p, w.tlen), offs, colinfo, wfirst, tbase, out, tile_cap, xcd_map_on() ? gt : 0u);
        }
        if (deep_mode == 2 && strip_rng) launch_deep(s, w, p, offs, colinfo, out, strip_rng, tile_cap, tbase);
        return;
    }
    // A wave's 64 rows go through an LDS slice only while that slice leaves the CU its waves: with --output-extra columns a wave's rows
    // reach 48 KB (three waves per CU), and the walker -- ~80 instructions per entry and pass, latency-bound -- ran 2.2x slower than
    // with byte stores straight to the text and full occupancy (mpileup30_B_sOx: 37.2 -> 17.1 ms, profiles/r04_generic_walker_lds_cap.md).
    // Waves above 8 KiB of rows therefore take the byte-store branch of k_mplp_emit (STA_GENERIC_LDS_CAP: experiment knob).
    static const uint32_t generic_cap = [] { const char *e = getenv("STA_GENERIC_LDS_CAP"); const int v = e ? atoi(e) : 8192; return (uint32_t)(v < 1024 ? 1024 : v); }();
    if (lds_cap > generic_cap) lds_cap = generic_cap;
    uint32_t slice = (lds_cap + 16 + 15) & ~15u;
    // waves per workgroup so that the workgroup's LDS (one slice per wave) stays within 64 KiB
    int wpb = 4 * slice <= 65536 ? 4 : (2 * slice <= 65536 ? 2 : 1);
    int64_t nb = (nwaves + wpb - 1) / wpb;
    // one walk for all strings of a row (emit_column_1walk) unless the row has more extra columns than it holds cursors for
    const char *pe = getenv("STA_GENERIC_PASSES");
    const char *gd = getenv("STA_GENERIC_DIAG"); const int gdiag = gd ? atoi(gd) : 0;
    const MplpDevPar par = make_par(p, w.tlen);
    if (pe && atoi(pe) == 2) gen_xlen = nullptr;           // STA_GENERIC_PASSES=2: the single-walk emit measuring for itself (A/B of k_mplp_len_x)
    if (sta_mplp_generic_extras(p) >= 0 && !(pe && atoi(pe) == 1))
        hipLaunchKernelGGL(k_mplp_emit<true>, dim3((unsigned)nb), dim3(64 * wpb), (size_t)wpb * slice, s, w, par, offs, out, lds_cap, gdiag, colinfo, gen_xlen);
    else
        hipLaunchKernelGGL(k_mplp_emit<false>, dim3((unsigned)nb), dim3(64 * wpb), (size_t)wpb * slice, s, w, par, offs, out, lds_cap, 0, (const uint2 *)nullptr, (const uint32_t *)nullptr);
}

int64_t sta_mplp_deep_strips(int64_t ncols) { return (ncols + DEEP_STRIP - 1) / DEEP_STRIP; }

// extra columns the read-major kernels cover (XfArgs): -O, --output-BP-5, --output-extra items and tags, up to XF_NX of them beside -s;
// not --output-mods.  STA_XFAST=0: back to the generic walkers (A/B runs, tests of the walkers).
bool sta_mplp_has_xfast_path(const sta_mplp_params &p)
{
    const char *e = getenv("STA_XFAST");
    if (e && atoi(e) == 0) return false;
    if ((uint32_t)p.flag & STA_MPLP_OUTPUT_MODS) return false;
    int kinds[XF_NX];
    const int nx = __builtin_popcount((uint32_t)p.flag & (uint32_t)XF_FLAGS) + (p.n_tags > 0 ? p.n_tags : 0);
    (void)kinds;
    return nx >= 1 && nx <= XF_NX && sta_mplp_tile_ok(p);
}
int sta_mplp_xfast_extras(const sta_mplp_params &p) { return sta_mplp_has_xfast_path(p) ? __builtin_popcount((uint32_t)p.flag & (uint32_t)XF_FLAGS) + (p.n_tags > 0 ? p.n_tags : 0) : 0; }
// no --output-extra / -O / -s columns: the window takes the fast kernel pair
// the tile kernels compare four quality bytes per word against -Q: it has to fit seven bits
bool sta_mplp_tile_ok(const sta_mplp_params &p) { return p.min_baseQ <= 127; }
// ... or -s alone among them: the mapping-quality column is one byte per entry that passed -Q, the quality string's mirror image, and
// rides along in the tile kernels (STA_TILE_NO_MQ=1: back to the generic walkers, for A/B runs)
bool sta_mplp_has_fast_path(const sta_mplp_params &p)
{
    static const bool no_mq = getenv("STA_TILE_NO_MQ") != nullptr;
    const uint32_t extra = (uint32_t)p.flag & (EXTRA_MASK | STA_MPLP_OUTPUT_MODS);
    return (extra == 0 || (extra == STA_MPLP_PRINT_MAPQ_CHAR && !no_mq)) && p.n_tags <= 0;
}
