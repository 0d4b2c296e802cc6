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
//                lines exceed the LDS slice falls back to direct global byte stores.
#include "dev_util.h"
#include <cstdlib>
#include "dev_lookback.h"

struct MplpDevPar {
    int32_t min_baseQ, all, rev_del, flag, no_ins, no_del, no_ends;
    int32_t n_tags, tag_sep;
    int64_t tlen;
};
#define TAGKIND (1 << 28)       // file_pass "kind" of tag column t is TAGKIND + t

__constant__ char c_nt_lc[17] = ",acmgrsvtwyhkdbn";
__constant__ char c_nt_uc[17] = ".ACMGRSVTWYHKDBN";
__constant__ char c_nt16_str[17] = "=ACMGRSVTWYHKDBN";
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

extern __shared__ __attribute__((aligned(16))) char lds_text[];

// ---- byte sink: LDS slice of this wave, or global memory ----
template <bool LDS> struct Sink {
    uint32_t cur;        // LDS: offset into lds_text; global: unused
    char *g;             // global cursor
    __device__ __forceinline__ void put(char c)
    {
        if (LDS) lds_text[cur++] = c;
        else *g++ = c;
    }
    __device__ __forceinline__ void put_dec(long long v)
    {
        if (v < 0) { put('-'); v = -v; }
        unsigned long long u = (unsigned long long)v;
        int n = dec_digits(u);
        if (LDS) {
            uint32_t e = cur + n;
            for (uint32_t q = e; q > cur;) { lds_text[--q] = (char)('0' + u % 10); u /= 10; }
            cur = e;
        } else {
            char *e = g + n;
            for (char *q = e; q > g;) { *--q = (char)('0' + u % 10); u /= 10; }
            g = e;
        }
    }
};

struct Resolved {
    int qpos, indel, k;
    bool is_del, is_refskip;
};

// stateless equivalent of HTSlib resolve_cigar2 for (read, column p) -- SURVEY.md A.2
__device__ __forceinline__ Resolved resolve_general(const uint32_t *cig, int n, int rpos, int p)
{
    Resolved r;
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
    r.k = k; r.indel = 0; r.is_del = false; r.is_refskip = false;
    if (x + l - 1 == p && k + 1 < n) {
        int op2 = cig[k + 1] & 0xf, l2 = (int)(cig[k + 1] >> 4);
        if (op2 == CG_D && op != CG_D) {
            r.indel = -l2;
            for (int j = k + 2; j < n; ++j) {
                if ((cig[j] & 0xf) == CG_D) r.indel -= (int)(cig[j] >> 4); else break;
            }
        } else if (op2 == CG_I) {
            r.indel = l2;
            for (int j = k + 2; j < n; ++j) {
                int o = cig[j] & 0xf;
                if (o == CG_I) r.indel += (int)(cig[j] >> 4);
                else if (o != CG_P) break;
            }
        } else if (op2 == CG_P && k + 2 < n) {
            int l3 = 0;
            for (int j = k + 2; j < n; ++j) {
                int o = cig[j] & 0xf;
                if (o == CG_I) l3 += (int)(cig[j] >> 4);
                else if (cg_is_refop(o)) break;
            }
            if (l3 > 0) r.indel = l3;
        }
    }
    if (cg_is_mop(op)) r.qpos = y + (p - x);
    else { r.is_del = true; r.qpos = y; r.is_refskip = (op == CG_N); }
    return r;
}

// bam_plp_insertion: total length (I+P run after op k) and the D that may follow it
__device__ __forceinline__ void insertion_shape(const uint32_t *cig, int n, int k, int &ins_total, int &del_after)
{
    ins_total = 0; del_after = 0;
    int j = k + 1;
    for (; j < n; ++j) {
        int o = cig[j] & 0xf;
        if (o == CG_I || o == CG_P) ins_total += (int)(cig[j] >> 4); else break;
    }
    if (j < n && (cig[j] & 0xf) == CG_D) del_after = (int)(cig[j] >> 4);
}

// One (read, column) entry after filtering: everything pileup_seq / the extra columns need.
struct Entry {
    int64_t r;           // read index
    int rpos, rend, lq;
    uint32_t info;
    uint64_t boff;       // base offset (bytes into qual; /2 into seq)
    Resolved rs;
};

__device__ __forceinline__ int token_len(const StaReadsDev &R, const MplpDevPar &P, const Entry &e, int p)
{
    int len = 1;
    if (!P.no_ends) len += (p == e.rpos ? 2 : 0) + (p == e.rend - 1 ? 1 : 0);
    if (e.rs.indel != 0) {
        int del_len = -e.rs.indel;
        if (e.rs.indel > 0) {
            const uint32_t *cig = R.cigar + R.cig_off[e.r];
            int n = (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]);
            int ins_total;
            insertion_shape(cig, n, e.rs.k, ins_total, del_len);
            if (P.no_ins < 2) len += 1 + dec_digits_u32((uint32_t)ins_total);
            if (!P.no_ins) len += ins_total;
        }
        if (del_len > 0) {
            if (P.no_del < 2) len += 1 + dec_digits_u32((uint32_t)del_len);
            if (!P.no_del) len += del_len;
        }
    }
    return len;
}

template <bool LDS>
__device__ __forceinline__ void token_write(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, const Entry &e, int p, Sink<LDS> &s)
{
    bool rev = (e.info & RI_REV) != 0;
    int64_t apos = W.origin + p;
    if (!P.no_ends && p == e.rpos) {
        int mq = (int)((e.info >> RI_MAPQ_SHIFT) & 0xff);
        s.put('^');
        s.put((char)(mq > 93 ? 126 : mq + 33));
    }
    if (!e.rs.is_del) {
        int c = e.rs.qpos < e.lq ? seq_nib(R.seq, e.boff >> 1, e.rs.qpos) : 15;
        if (W.ref) {
            int rb = apos < W.ref_len ? nt16_from_char((unsigned char)W.ref[apos]) : 15;
            if (c == rb) c = 0;
        }
        s.put(rev ? c_nt_lc[c] : c_nt_uc[c]);
    } else {
        s.put(e.rs.is_refskip ? (rev ? '<' : '>') : ((rev && P.rev_del) ? '#' : '*'));
    }
    if (e.rs.indel != 0) {
        int del_len = -e.rs.indel;
        if (e.rs.indel > 0) {
            const uint32_t *cig = R.cigar + R.cig_off[e.r];
            int n = (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]);
            int ins_total;
            insertion_shape(cig, n, e.rs.k, ins_total, del_len);
            if (P.no_ins < 2) { s.put('+'); s.put_dec(ins_total); }
            if (!P.no_ins) {
                char pad = (rev && P.rev_del) ? '#' : '*';
                int j = 1;
                for (int kk = e.rs.k + 1; kk < n; ++kk) {
                    int o = cig[kk] & 0xf, l = (int)(cig[kk] >> 4);
                    if (o == CG_P) { for (int t = 0; t < l; ++t) s.put(pad); }
                    else if (o == CG_I) {
                        for (int t = 0; t < l; ++t, ++j) {
                            int qi = e.rs.qpos + j - (e.rs.is_del ? 1 : 0);
                            char ch = qi < e.lq ? c_nt16_str[seq_nib(R.seq, e.boff >> 1, qi)] : 'N';
                            s.put(rev ? lower_c(ch) : upper_c(ch));
                        }
                    } else break;
                }
            }
        }
        if (del_len > 0) {
            if (P.no_del < 2) { s.put('-'); s.put_dec(del_len); }
            if (!P.no_del) {
                for (int j = 1; j <= del_len; ++j) {
                    // reference: (ref && (int)pos+j < ref_len) ? ref[pos+j] : 'N'   (bam_plcmd.c:158)
                    char c = (W.ref && (int64_t)((int)apos + j) < W.ref_len) ? W.ref[apos + j] : 'N';
                    s.put(rev ? lower_c(c) : upper_c(c));
                }
            }
        }
    }
    if (!P.no_ends && p == e.rend - 1) s.put('$');
}

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
template <bool LDS>
__device__ __forceinline__ void extra_write(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int kind, const Entry &e, Sink<LDS> &s)
{
    if (kind == STA_MPLP_PRINT_RNEXT || kind >= TAGKIND) {
        const uint32_t *o = R.xcol_off + (uint64_t)e.r * (uint64_t)R.n_xcols + (uint64_t)xcol_index(P, kind);
        for (uint32_t t = o[0]; t < o[1]; ++t) s.put(R.xcol_text[t]);
    } else if (kind == STA_MPLP_PRINT_MAPQ_CHAR) {
        int c = (int)((e.info >> RI_MAPQ_SHIFT) & 0xff) + 33;
        s.put((char)(c > 126 ? 126 : c));
    } else if (kind == STA_MPLP_PRINT_QNAME) {
        const char *nm = R.names + R.name_off[e.r];
        int l = (int)(R.name_off[e.r + 1] - R.name_off[e.r]) - 1;
        for (int t = 0; t < l; ++t) s.put(nm[t]);
    } else if (kind == STA_MPLP_PRINT_RNAME) {
        for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
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

// The next wave tile's range from the previous one's (tiles are visited left to right, so both bounds only move forward): one
// coalesced 64-entry probe per bound in the common case instead of two 64-ary searches of ~4 dependent loads each.
__device__ __forceinline__ void wave_read_range_next(const StaReadsDev &R, int p0, int p1, int64_t &rlo, int64_t &rhi, bool have_prev)
{
    if (R.n == 0) { rlo = rhi = 0; return; }
    if (!have_prev) { wave_read_range(R, p0, p1, rlo, rhi); return; }
    const int lane = threadIdx.x & 63;
    bool found = false;
    for (int it = 0; it < 3 && !found; ++it) {
        const int64_t idx = rlo + lane;
        const unsigned long long m = __ballot(idx < R.n ? R.maxend[idx] > p0 : true);
        if (m) { rlo += __ffsll((long long)m) - 1; found = true; } else rlo += 64;
    }
    if (!found) rlo = wave_upper_bound(R.maxend, R.n, p0);
    if (rlo > R.n) rlo = R.n;
    found = false;
    for (int it = 0; it < 3 && !found; ++it) {
        const int64_t idx = rhi + lane;
        const unsigned long long m = __ballot(idx < R.n ? R.pos[idx] > p1 : true);
        if (m) { rhi += __ffsll((long long)m) - 1; found = true; } else rhi += 64;
    }
    if (!found) rhi = wave_upper_bound(R.pos, R.n, p1);
    if (rhi > R.n) rhi = R.n;
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

__global__ void __launch_bounds__(256) k_mplp_emit(StaWinDev W, MplpDevPar P, const uint64_t *__restrict__ offs, char *out, uint32_t lds_cap)
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
        emit_column<true>(W, P, p0, plast, p, exists, s);
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
        Sink<false> s; s.cur = 0; s.g = out + my0;
        emit_column<false>(W, P, p0, plast, p, exists, s);
    }
}


// ================================================================================================
// Fast column kernels (no --output-extra / -O / -s columns): same column-per-lane layout, but
//  * read metadata is fetched 64 reads at a time with coalesced vector loads and broadcast with
//    v_readlane, so the walk issues no dependent scalar loads;
//  * only reads that can touch the wave's 64 columns are visited (ballot of a per-lane test);
//  * four reads are in flight at once (their quality / base bytes are loaded before any is used);
//  * the measuring pass stores (count, seq bytes) per column and file, so the emit pass walks the
//    reads ONCE and writes the base string and the quality string at two cursors.
// Non-simple reads (indels, clips, pads, ref skips) take the generic per-entry path inline.

// Pointers that reach a kernel through W.files[] (a struct read from memory) are "generic" to the compiler, which then
// emits flat_load (slower, and it couples vmcnt with lgkmcnt).  They always point to HBM: say so.
#define GPTR(T, p) ((const __attribute__((address_space(1))) T *)(p))

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

template <bool EMIT, bool LDS>
__device__ __forceinline__ void fast_walk(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int p0, int plast,
                                          int p, bool active, int rbcode, int64_t rlo, int64_t rhi,
                                          uint32_t &n_plp, uint32_t &cnt, uint32_t &seq_len, Sink<LDS> &ss, Sink<LDS> &sq)
{
    const int lane = threadIdx.x & 63;
    const bool ends = !P.no_ends;
    const auto g_info = GPTR(uint32_t, R.info); const auto g_pos = GPTR(int32_t, R.pos); const auto g_end = GPTR(int32_t, R.end);
    const auto g_b8 = GPTR(uint32_t, R.base_off8); const auto g_qual = GPTR(uint8_t, R.qual); const auto g_seq = GPTR(uint8_t, R.seq);
    for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
        const int64_t ri = b0 + lane;
        const bool ok = ri < rhi;
        const uint32_t v_info = ok ? g_info[ri] : 0u;
        const int v_pos = ok ? g_pos[ri] : 0;
        const int v_end = ok ? g_end[ri] : 0;
        const uint32_t v_b8 = ok ? g_b8[ri] : 0u;
        unsigned long long live = __ballot(ok && (v_info & RI_KEEP) && v_end > p0 && v_pos <= plast);
        while (live) {
            bool valid[4], cov[4];
            uint32_t info[4], b8[4];
            int rpos[4], rend[4], jx[4], qv[4], sv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                valid[k] = live != 0;
                int j = valid[k] ? __ffsll((long long)live) - 1 : 0;
                if (valid[k]) live &= live - 1;
                jx[k] = j;
                info[k] = rl_u(v_info, j); rpos[k] = rl_i(v_pos, j); rend[k] = rl_i(v_end, j); b8[k] = rl_u(v_b8, j);
                cov[k] = valid[k] && active && p >= rpos[k] && p < rend[k];
                qv[k] = 0; sv[k] = 0;
                if (cov[k] && (info[k] & RI_SIMPLE)) {
                    uint64_t boff = (uint64_t)b8[k] << 3;
                    int qpos = p - rpos[k];
                    qv[k] = g_qual[boff + (uint64_t)qpos];
                    if (EMIT) sv[k] = g_seq[(boff >> 1) + (uint64_t)(qpos >> 1)];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!valid[k]) continue;
                if (info[k] & RI_SIMPLE) {
                    // branch-free: every lane runs the same code; a lane that has nothing to add writes at its cursor
                    // without advancing it (the byte is overwritten by its next real write, or by the separator that
                    // emit_column_fast puts there after the walk)
                    const bool pass = cov[k] && qv[k] >= P.min_baseQ;
                    const bool head = pass && ends && p == rpos[k], tail = pass && ends && p == rend[k] - 1;
                    if (!EMIT) {
                        n_plp += cov[k] ? 1u : 0u;
                        cnt += pass ? 1u : 0u;
                        seq_len += (pass ? 1u : 0u) + (head ? 2u : 0u) + (tail ? 1u : 0u);
                    } else if (LDS) {
                        const bool rev = (info[k] & RI_REV) != 0;
                        const int mq = (int)((info[k] >> RI_MAPQ_SHIFT) & 0xff);
                        const int qpos = p - rpos[k];
                        int c = (sv[k] >> ((~qpos & 1) << 2)) & 0xf;
                        if (c == rbcode) c = 0;
                        uint32_t cur = ss.cur;
                        lds_text[cur] = '^';
                        lds_text[cur + (head ? 1u : 0u)] = (char)(mq > 93 ? 126 : mq + 33);
                        cur += head ? 2u : 0u;
                        lds_text[cur] = base_char_fast(c, rev);
                        cur += pass ? 1u : 0u;
                        lds_text[cur] = '$';
                        cur += tail ? 1u : 0u;
                        ss.cur = cur;
                        lds_text[sq.cur] = (char)(qv[k] + 33 < 126 ? qv[k] + 33 : 126);
                        sq.cur += pass ? 1u : 0u;
                    } else if (pass) {
                        bool rev = (info[k] & RI_REV) != 0;
                        if (head) {
                            int mq = (int)((info[k] >> RI_MAPQ_SHIFT) & 0xff);
                            ss.put('^'); ss.put((char)(mq > 93 ? 126 : mq + 33));
                        }
                        int qpos = p - rpos[k];
                        int c = (sv[k] >> ((~qpos & 1) << 2)) & 0xf;
                        if (c == rbcode) c = 0;
                        ss.put(base_char_fast(c, rev));
                        if (tail) ss.put('$');
                        sq.put((char)(qv[k] + 33 < 126 ? qv[k] + 33 : 126));
                    }
                } else {
                    // generic entry (uniform branch: the read is the same for every lane)
                    if (__ballot(cov[k]) == 0) continue;
                    if (cov[k]) {
                        Entry e;
                        e.r = b0 + jx[k]; e.rpos = rpos[k]; e.rend = rend[k]; e.info = info[k];
                        e.lq = R.l_qseq[e.r];
                        e.boff = (uint64_t)b8[k] << 3;
                        e.rs = resolve_general(R.cigar + R.cig_off[e.r], (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]), e.rpos, p);
                        if (!EMIT) n_plp++;
                        int c = e.rs.is_del ? placeholder_qual(R, e.r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
                        if (c >= P.min_baseQ) {
                            if (!EMIT) { cnt++; seq_len += (uint32_t)token_len(R, P, e, p); }
                            else {
                                token_write<LDS>(R, W, P, e, p, ss);
                                sq.put((char)(c + 33 < 126 ? c + 33 : 126));
                            }
                        }
                    }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_mplp_len_fast(StaWinDev W, MplpDevPar P, uint32_t *line_len, uint2 *colinfo, StaCounters *ctr)
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
    Sink<false> d1, d2; d1.g = d2.g = nullptr; d1.cur = d2.cur = 0;
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        uint32_t n_plp = 0, cnt = 0, seq_len = 0;
        fast_walk<false, false>(R, W, P, p0, plast, p, active, -1, rlo, rhi, n_plp, cnt, seq_len, d1, d2);
        any |= n_plp > 0;
        total += 1 + (uint32_t)dec_digits_u32(cnt) + 1 + (seq_len ? seq_len : 1) + 1 + (cnt ? cnt : 1);
        if (active) colinfo[(int64_t)f * ncols + c0 + lane] = make_uint2(cnt, seq_len);
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
__device__ __forceinline__ void emit_column_fast(const StaWinDev &W, const MplpDevPar &P, const uint2 *colinfo, int64_t ncols, int64_t col,
                                                 int p0, int plast, int p, bool exists, Sink<LDS> &s, uint32_t dump)
{
    int64_t apos = W.origin + p;
    int rbcode = -1;
    if (exists) {
        for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
        s.put('\t');
        s.put_dec(apos + 1);
        s.put('\t');
        char rc = (W.ref && apos < W.ref_len) ? W.ref[apos] : 'N';
        s.put(rc);
        if (W.ref) rbcode = apos < W.ref_len ? (int)c_nt16_of_char[(unsigned char)rc] : 15;
    }
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo, rhi;
        wave_read_range(R, p0, plast, rlo, rhi);
        uint2 ci = exists ? colinfo[(int64_t)f * ncols + col] : make_uint2(0, 0);
        uint32_t cnt = ci.x, seq_len = ci.y;
        Sink<LDS> ss = s, sq = s;
        uint32_t sl = seq_len ? seq_len : 1;
        if (exists) {
            s.put('\t'); s.put_dec(cnt); s.put('\t');
            ss = s;
            sq = s; sq.cur += sl + 1; sq.g += sl + 1;
        }
        const bool walk = exists && cnt;
        if (LDS && !walk) ss.cur = sq.cur = dump;       // lanes without entries: predicated writes land in the wave's dump bytes
        uint32_t d0 = 0, d1 = 0, d2 = 0;
        fast_walk<true, LDS>(R, W, P, p0, plast, p, walk, rbcode, rlo, rhi, d0, d1, d2, ss, sq);
        if (exists) {
            // separators and the '*' placeholders go in AFTER the walk (its last predicated write may sit on them)
            Sink<LDS> st = s;
            if (!cnt) { st.put('*'); st.put('\t'); st.put('*'); }
            else { st.cur += sl; st.g += sl; st.put('\t'); }
            s.cur += sl + 1 + (cnt ? cnt : 1); s.g += sl + 1 + (cnt ? cnt : 1);
        }
    }
    if (exists) s.put('\n');
}

__global__ void __launch_bounds__(256) k_mplp_emit_fast(StaWinDev W, MplpDevPar P, const uint64_t *__restrict__ offs, const uint2 *__restrict__ colinfo,
                                                        char *out, uint32_t lds_cap)
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
        uint32_t slice = (lds_cap + 48 + 15) & ~15u;       // text (+ up to 15 alignment bytes) + 16 dump bytes for predicated writes
        uint32_t base = (uint32_t)wid * slice;
        uint32_t mis = (uint32_t)((uintptr_t)(out + o0) & 15);
        Sink<true> s; s.g = nullptr; s.cur = base + mis + (uint32_t)(my0 - o0);
        emit_column_fast<true>(W, P, colinfo, ncols, c0 + lane, p0, plast, p, exists, s, base + slice - 8);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
        Sink<false> s; s.cur = 0; s.g = out + my0;
        emit_column_fast<false>(W, P, colinfo, ncols, c0 + lane, p0, plast, p, exists, s, 0);
    }
}


// ================================================================================================
// Single-pass column kernel: measuring pass, offsets and text in ONE launch (no k_mplp_len / scan / k_col_stats launches,
// no host round trip between "how long are the lines" and "write them").
//
//  * A workgroup takes a ticket (tile = 256 consecutive columns, one wave per 64) -- tiles are handed out in start order,
//    so a tile only ever waits for tiles that are already running.
//  * COUNT pass: every wave walks its candidate reads and counts, per column, the entries, the entries that pass -Q and
//    the base-string bytes; line lengths -> wave / workgroup totals.
//  * Decoupled look-back over the tiles' status words ({flag, bytes} and {flag, rows, data columns}, each ONE 8-byte word
//    written with a device-scope store, so the value IS the flag: no fence pairs) gives the tile's byte offset in the
//    output; the last tile leaves the window totals in StaCounters.
//  * EMIT pass: the wave walks the same reads again and builds its lines in its LDS line buffer at their final relative
//    positions, then flushes with coalesced 16-byte stores.  A wave whose text exceeds the line buffer emits it in rounds
//    of consecutive columns (deep columns: 300x needs ~40 KB per 64 columns); a single line longer than the buffer is
//    written straight to global memory.  LDS per wave is therefore a constant, whatever the depth.
//
// The walk itself is restructured for throughput: the candidate reads' metadata is loaded 64 at a time, the live ones are
// compacted, and their bases are converted READ-MAJOR -- 16 lanes per read, 4 columns per lane, one 4-byte load of
// qualities and one of packed bases per lane -- into a small LDS tile (one byte of quality-character + pass flag and one
// byte of base code per (read, column)).  The column-major pass then costs one LDS byte read per (read, lane) instead of
// per-lane global byte loads and the per-read address arithmetic.  Reads that are not a single M run take the generic
// per-entry path (resolve_general / token_len / token_write), as in k_mplp_emit_fast.

#define FT_ROWS 32                       // reads per LDS tile
#define FT_META 1024                     // 4 arrays x 64 x 4 bytes of compacted read metadata
#define FT_TILE (2 * 64 * FT_ROWS)

struct FusedArgs {
    unsigned long long *status;          // 2 words per tile, zeroed before the launch
    unsigned int *ticket;                // zeroed before the launch
    char *out; unsigned long long capacity;
    uint2 *colinfo;                      // [nfiles][ncols] (count, seq bytes) when the window has more than one input file
    StaCounters *ctr;
    uint32_t lbuf;                       // line-buffer bytes per wave
    uint32_t per_wave;                   // LDS bytes per wave (metadata + tile + line buffer slice)
    uint32_t n_tiles;
    uint32_t tiles_per_batch;            // consecutive tiles one workgroup (one ticket) works through
    // waves holding a line longer than the line buffer: their wave index goes to giant[ctr->n_giant++], the per-column
    // offsets to offs[] and (one input file) the per-column counts to colinfo_one[]; k_mplp_emit_listed writes them
    uint32_t *giant; unsigned long long *offs; uint2 *colinfo_one;
};

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// MODE 0: count (n_plp, cnt, seq_len)   MODE 1: write base string at ss and quality string at sq
//
// Tile layout: tq32[(row / 4) * 64 + column] packs the quality bytes of FOUR consecutive reads for one column into one dword
// (byte row & 3), tc32 likewise the base codes -- the column pass fetches four reads with one LDS load per array and lane.
// (one instance serves both passes -- `emit` is wave-uniform -- so that the kernel's code stays small: the conversion and the
// generic-entry path appear once)
__device__ __forceinline__ void fused_walk(const bool emit, const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int p0, int plast, int p, bool active,
                                           int rbcode, int64_t rlo, int64_t rhi, char *wl, uint32_t &n_plp, uint32_t &cnt, uint32_t &seq_len,
                                           Sink<true> &ss, Sink<true> &sq)
{
    const int lane = threadIdx.x & 63;
    const bool ends = !P.no_ends;
    const auto g_info = GPTR(uint32_t, R.info); const auto g_pos = GPTR(int32_t, R.pos); const auto g_end = GPTR(int32_t, R.end);
    const auto g_b8 = GPTR(uint32_t, R.base_off8); const auto g_qual = GPTR(uint8_t, R.qual); const auto g_seq = GPTR(uint8_t, R.seq);
    uint32_t *m_info = reinterpret_cast<uint32_t *>(wl); int32_t *m_pos = reinterpret_cast<int32_t *>(wl + 256);
    int32_t *m_end = reinterpret_cast<int32_t *>(wl + 512); uint32_t *m_b8 = reinterpret_cast<uint32_t *>(wl + 768);
    uint8_t *tq = reinterpret_cast<uint8_t *>(wl + FT_META), *tc = tq + 64 * FT_ROWS;
    const uint32_t *tq32 = reinterpret_cast<const uint32_t *>(tq), *tc32 = reinterpret_cast<const uint32_t *>(tc);
    const int grp = lane >> 4, t4 = (lane & 15) << 2;
    const uint64_t qual_bytes = R.n_bases_total, seq_bytes = R.n_bases_total >> 1;
    constexpr int NSTEP = FT_ROWS / 4;
    for (int64_t b0 = rlo; b0 < rhi; b0 += 64) {
        const int64_t ri = b0 + lane;
        const bool ok = ri < rhi;
        const uint32_t v_info = ok ? g_info[ri] : 0u;
        const int v_pos = ok ? g_pos[ri] : 0;
        const int v_end = ok ? g_end[ri] : 0;
        const uint32_t v_b8 = ok ? g_b8[ri] : 0u;
        const bool lv = ok && (v_info & RI_KEEP) && v_end > p0 && v_pos <= plast;
        const unsigned long long live = __ballot(lv);
        if (!live) continue;
        const int nlive = __popcll(live);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        wave_lds_sync();                                 // the previous batch's tile / metadata readers are done
        if (lv) { m_info[rank] = (v_info & 0x00ffffffu) | ((uint32_t)lane << 24); m_pos[rank] = v_pos; m_end[rank] = v_end; m_b8[rank] = v_b8; }
        wave_lds_sync();
        // compacted copies: lane k holds live read k (lanes >= nlive hold stale words that are never selected)
        const uint32_t c_info = m_info[lane]; const int c_pos = m_pos[lane], c_end = m_end[lane]; const uint32_t c_b8 = m_b8[lane];
        for (int t0 = 0; t0 < nlive; t0 += FT_ROWS) {
            const int nt = nlive - t0 < FT_ROWS ? nlive - t0 : FT_ROWS;
            if (t0) wave_lds_sync();                     // the previous tile's column pass is done
            // ---- read-major conversion: 4 reads per step, 16 lanes per read, 4 columns per lane.  All loads of the tile are
            //      issued before the first one is used (one exposed memory latency per tile, not per step) ----
            uint32_t qd[NSTEP], sd[NSTEP];
#pragma unroll
            for (int u = 0; u < NSTEP; ++u) {
                qd[u] = 0; sd[u] = 0;
                const int row = 4 * u + grp;
                if (row < nt) {
                    const int src = t0 + row;
                    if (m_info[src] & RI_SIMPLE) {
                        const int rpos = m_pos[src], rl = m_end[src] - rpos;
                        const int qpos0 = p0 + t4 - rpos;
                        if (qpos0 > -4 && qpos0 < rl) {
                            const uint64_t boff = (uint64_t)m_b8[src] << 3;
                            const int ld = qpos0 < 0 ? 0 : qpos0;
                            const uint64_t qa = boff + (uint64_t)ld, sa = (boff >> 1) + (uint64_t)(ld >> 1);
                            if (qa + 4 <= qual_bytes) qd[u] = *reinterpret_cast<const __attribute__((address_space(1))) u32_unaligned *>(g_qual + qa);
                            else { for (int k = 0; k < 4; ++k) if (qa + (uint64_t)k < qual_bytes) qd[u] |= (uint32_t)g_qual[qa + k] << (8 * k); }
                            if (sa + 4 <= seq_bytes) sd[u] = *reinterpret_cast<const __attribute__((address_space(1))) u32_unaligned *>(g_seq + sa);
                            else { for (int k = 0; k < 4; ++k) if (sa + (uint64_t)k < seq_bytes) sd[u] |= (uint32_t)g_seq[sa + k] << (8 * k); }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NSTEP; ++u) {
                const int row = 4 * u + grp;
                if (row < nt) {
                    const int src = t0 + row;
                    if (m_info[src] & RI_SIMPLE) {
                        const int rpos = m_pos[src], rl = m_end[src] - rpos;
                        const int qpos0 = p0 + t4 - rpos;
                        const int ld = qpos0 < 0 ? 0 : qpos0;
                        const bool any_valid = qpos0 > -4 && qpos0 < rl;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int qpos = qpos0 + k;
                            const bool valid = any_valid && qpos >= 0 && qpos < rl;
                            const int q = (int)((qd[u] >> (8 * ((qpos - ld) & 3))) & 0xffu);
                            const int n = qpos - (ld & ~1);                    // nibble index inside sd (high nibble first)
                            const uint32_t c = (sd[u] >> (8 * ((n >> 1) & 3) + ((n & 1) ? 0 : 4))) & 0xfu;
                            int qc = q + 33; qc = qc > 126 ? 126 : qc;
                            qc |= q >= P.min_baseQ ? 0x80 : 0;
                            const int at = ((u * 64 + t4 + k) << 2) + grp;
                            tq[at] = (uint8_t)(valid ? qc : 0);
                            tc[at] = (uint8_t)c;
                        }
                    }
                }
            }
            wave_lds_sync();
            // ---- column-major pass: one lane per column, reads in file order, four reads per LDS load ----
            uint32_t nq4 = tq32[lane], nc4 = tc32[lane];
            for (int u = 0; 4 * u < nt; ++u) {
                const uint32_t q4 = nq4, c4 = nc4;
                if (4 * (u + 1) < nt) { nq4 = tq32[(u + 1) * 64 + lane]; nc4 = tc32[(u + 1) * 64 + lane]; }   // next group: in flight during this one
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 4 * u + k;
                    if (row >= nt) break;
                    const int j = t0 + row;
                    const uint32_t info = rl_u(c_info, j);
                    const int rpos = rl_i(c_pos, j), rend = rl_i(c_end, j);
                    if (info & RI_SIMPLE) {
                        const uint32_t qb = (q4 >> (8 * k)) & 0xffu;
                        const bool pass = (qb & 0x80u) && active;
                        if (!emit) {
                            const bool cov = qb != 0 && active;
                            n_plp += cov ? 1u : 0u;
                            cnt += pass ? 1u : 0u;
                            seq_len += pass ? 1u : 0u;
                            // head / tail marks: only when the read starts / ends inside this wave's 64 columns (uniform branches)
                            if (ends && rpos >= p0 && rpos <= plast) seq_len += (pass && p == rpos) ? 2u : 0u;
                            if (ends && rend - 1 >= p0 && rend - 1 <= plast) seq_len += (pass && p == rend - 1) ? 1u : 0u;
                        } else {
                            const bool rev = (info & RI_REV) != 0;
                            int c = (int)((c4 >> (8 * k)) & 0xffu);
                            if (c == rbcode) c = 0;
                            const char ch = base_char_fast(c, rev);
                            const char qch = (char)(qb & 0x7fu);
                            const bool head_here = ends && rpos >= p0 && rpos <= plast, tail_here = ends && rend - 1 >= p0 && rend - 1 <= plast;
                            const int mq = (int)((info >> RI_MAPQ_SHIFT) & 0xff);
                            // branch-free appends: a lane with nothing to add writes at its cursor without advancing it (the byte is
                            // overwritten by its next real write or by the separator fused_emit_lines puts there after the walk)
                            uint32_t cur = ss.cur;
                            if (head_here) {
                                const bool head = pass && p == rpos;
                                lds_text[cur] = '^';
                                lds_text[cur + (head ? 1u : 0u)] = (char)(mq > 93 ? 126 : mq + 33);
                                cur += head ? 2u : 0u;
                            }
                            lds_text[cur] = ch;
                            cur += pass ? 1u : 0u;
                            if (tail_here) {
                                const bool tail = pass && p == rend - 1;
                                lds_text[cur] = '$';
                                cur += tail ? 1u : 0u;
                            }
                            ss.cur = cur;
                            lds_text[sq.cur] = qch;
                            sq.cur += pass ? 1u : 0u;
                        }
                    } else {
                        // generic entry (uniform branch: the read is the same for every lane)
                        const bool cov = active && p >= rpos && p < rend;
                        if (__ballot(cov) == 0) continue;
                        if (cov) {
                            Entry e;
                            e.r = b0 + (int64_t)(info >> 24); e.rpos = rpos; e.rend = rend; e.info = info & 0x00ffffffu;
                            e.lq = R.l_qseq[e.r];
                            e.boff = (uint64_t)rl_u(c_b8, j) << 3;
                            e.rs = resolve_general(R.cigar + R.cig_off[e.r], (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]), e.rpos, p);
                            if (!emit) n_plp++;
                            int c = e.rs.is_del ? placeholder_qual(R, e.r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
                            if (c >= P.min_baseQ) {
                                if (!emit) { cnt++; seq_len += (uint32_t)token_len(R, P, e, p); }
                                else {
                                    token_write<true>(R, W, P, e, p, ss);
                                    sq.put((char)(c + 33 < 126 ? c + 33 : 126));
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// the lines of the wave's columns (lanes with `mine`), each at s (LDS: final position inside the wave's line buffer; global: final
// address).  Mirrors emit_column_fast; (count, base-string bytes) per file come from the COUNT pass: registers for one input file,
// colinfo otherwise.
__device__ __forceinline__ void fused_emit_lines(const StaWinDev &W, const MplpDevPar &P, const uint2 *colinfo, int64_t ncols, int64_t col,
                                                 int p0, int plast, int p, bool mine, uint32_t cnt0, uint32_t sl0, char *wl, Sink<true> &s, uint32_t dump,
                                                 int64_t rlo1, int64_t rhi1)
{
    const int64_t apos = W.origin + p;
    int rbcode = -1;
    if (mine) {
        for (int t = 0; t < W.tname_len; ++t) s.put(W.tname[t]);
        s.put('\t');
        s.put_dec(apos + 1);
        s.put('\t');
        const char rc = (W.ref && apos < W.ref_len) ? W.ref[apos] : 'N';
        s.put(rc);
        if (W.ref) rbcode = apos < W.ref_len ? (int)c_nt16_of_char[(unsigned char)rc] : 15;
    }
    for (int f = 0; f < W.nfiles; ++f) {
        const StaReadsDev &R = W.files[f];
        int64_t rlo = rlo1, rhi = rhi1;                      // the COUNT pass's range when the window has one input file
        if (W.nfiles > 1) wave_read_range(R, p0, plast, rlo, rhi);
        uint32_t cnt = cnt0, seq_len = sl0;
        if (colinfo) { const uint2 ci = mine ? colinfo[(int64_t)f * ncols + col] : make_uint2(0, 0); cnt = ci.x; seq_len = ci.y; }
        if (!mine) { cnt = 0; seq_len = 0; }
        Sink<true> ss = s, sq = s;
        const uint32_t sl = seq_len ? seq_len : 1;
        if (mine) {
            s.put('\t'); s.put_dec(cnt); s.put('\t');
            ss = s;
            sq = s; sq.cur += sl + 1; sq.g += sl + 1;
        }
        const bool walk = mine && cnt;
        if (!walk) ss.cur = sq.cur = dump;     // lanes without entries: predicated writes land in the wave's dump bytes
        uint32_t d0 = 0, d1 = 0, d2 = 0;
        fused_walk(true, R, W, P, p0, plast, p, walk, rbcode, rlo, rhi, wl, d0, d1, d2, ss, sq);
        if (mine) {
            // separators and the '*' placeholders go in AFTER the walk (its last predicated write may sit on them)
            Sink<true> st = s;
            if (!cnt) { st.put('*'); st.put('\t'); st.put('*'); }
            else { st.cur += sl; st.g += sl; st.put('\t'); }
            s.cur += sl + 1 + (cnt ? cnt : 1); s.g += sl + 1 + (cnt ? cnt : 1);
        }
    }
    if (mine) s.put('\n');
}

__global__ void __launch_bounds__(256) k_mplp_fused(StaWinDev W, MplpDevPar P, FusedArgs A)
{
    __shared__ unsigned int s_batch;
    __shared__ unsigned long long s_wtot[4][2];
    __shared__ unsigned long long s_base[2];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // one ticket per BATCH of consecutive tiles (a ticket word sustains ~90 atomics/us: one per tile would cost more than the
    // tiles); batches are handed out in start order, so a tile only ever waits for tiles that are already running
    if (threadIdx.x == 0) s_batch = atomicAdd(A.ticket, 1u);
    __syncthreads();
    const unsigned int batch = s_batch;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    char *wl = lds_text + (size_t)wid * A.per_wave;
    const uint32_t lb = (uint32_t)wid * A.per_wave + FT_META + FT_TILE;          // line buffer: index into lds_text
    const uint32_t dump = lb + ((A.lbuf + 16 + 15) & ~15u);                      // 16 bytes behind the text area (+ alignment slack)
    const uint2 *ci = A.colinfo;
    int64_t rlo1 = 0, rhi1 = 0; bool have_range = false;                         // single input file: the wave's read range moves forward tile by tile

    for (unsigned ti = 0; ti < A.tiles_per_batch; ++ti) {
        const unsigned int tile = batch * A.tiles_per_batch + ti;
        if (tile >= A.n_tiles) break;
        __syncthreads();                                     // every wave is done with the previous tile's shared words
        const int64_t c0 = ((int64_t)tile * 4 + wid) * 64;
        const bool wave_on = c0 < ncols;
        const int p0 = W.col_beg + (int)(wave_on ? c0 : 0);
        const int p = p0 + lane;
        const bool active = wave_on && p < W.col_end;
        const int plast = p0 + 63 < W.col_end ? p0 + 63 : W.col_end - 1;
        const int64_t apos = W.origin + p;

        // ---- COUNT ----
        uint32_t total = 0, cnt0 = 0, sl0 = 0; bool any = false;
        if (wave_on) {
            Sink<true> d1, d2; d1.g = d2.g = nullptr; d1.cur = d2.cur = 0;
            for (int f = 0; f < W.nfiles; ++f) {
                const StaReadsDev &R = W.files[f];
                int64_t rlo, rhi;
                if (W.nfiles == 1) { wave_read_range_next(R, p0, plast, rlo1, rhi1, have_range); have_range = true; rlo = rlo1; rhi = rhi1; }
                else wave_read_range(R, p0, plast, rlo, rhi);
                uint32_t n_plp = 0, cnt = 0, seq_len = 0;
                fused_walk(false, R, W, P, p0, plast, p, active, -1, rlo, rhi, wl, n_plp, cnt, seq_len, d1, d2);
                any |= n_plp > 0;
                total += 1 + (uint32_t)dec_digits_u32(cnt) + 1 + (seq_len ? seq_len : 1) + 1 + (cnt ? cnt : 1);
                if (A.colinfo) { if (active) A.colinfo[(int64_t)f * ncols + c0 + lane] = make_uint2(cnt, seq_len); }
                else { cnt0 = cnt; sl0 = seq_len; }
            }
        }
        const bool in_reg = active && column_selected(W, apos);
        const bool data = in_reg && any;
        bool exists = in_reg && (any || (P.all && apos < P.tlen));
        if (exists && W.has_bed) exists = bed_overlap_dev(W.bed_beg, W.bed_end, W.n_bed, apos, apos + 1);
        uint32_t len = 0;
        if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1 + 1 + total + 1;
        // offsets inside the wave
        uint32_t incl = len;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
        const uint32_t excl = incl - len;
        const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const unsigned long long n_rows = (unsigned long long)__popcll(__ballot(exists)), n_data = (unsigned long long)__popcll(__ballot(data));
        if (lane == 0) { s_wtot[wid][0] = wave_total; s_wtot[wid][1] = (n_rows << 31) | n_data; }
        __syncthreads();

        // ---- decoupled look-back (wave 0) ----
        if (wid == 0) {
            const unsigned long long agg0 = s_wtot[0][0] + s_wtot[1][0] + s_wtot[2][0] + s_wtot[3][0];
            const unsigned long long agg1 = s_wtot[0][1] + s_wtot[1][1] + s_wtot[2][1] + s_wtot[3][1];
            unsigned long long ex0, ex1;
            tile_lookback(A.status, tile, agg0, agg1, ex0, ex1);
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
        if (wg_off + wg_bytes > A.capacity) { if (threadIdx.x == 0) A.ctr->overflow = 1; continue; }    // counted, not written: the host retries with room
        if (!wave_on || wave_total == 0) continue;
        unsigned long long wave_off = wg_off;
        for (int w = 0; w < wid; ++w) wave_off += s_wtot[w][0];

        // ---- EMIT, in rounds of consecutive columns that fit the line buffer ----
        int a = 0; bool listed = false;
        while (a < 64) {
            const uint32_t start = (uint32_t)__shfl((int)excl, a);
            const bool fits = lane >= a && incl - start <= A.lbuf;
            const int nb = __popcll(__ballot(fits));                             // line ends are monotone: the fitting lanes are a .. a+nb-1
            if (nb == 0) {
                // One line longer than the whole buffer (thousands of reads deep): this wave's columns go on the list of the
                // follow-up kernel, which writes such lines straight to global memory (k_mplp_emit_listed) from the offsets and
                // per-column counts left here.  Rare; the rest of the wave is still written below.
                if (!listed) {
                    listed = true;
                    if (active) {
                        A.offs[c0 + lane] = wave_off + excl;
                        if (!ci) A.colinfo_one[c0 + lane] = make_uint2(cnt0, sl0);
                    }
                    if (lane == 0) {
                        A.offs[c0 + 64 < ncols ? c0 + 64 : ncols] = wave_off + wave_total;
                        const unsigned long long k = atomicAdd(&A.ctr->n_giant, 1ull);
                        A.giant[k] = (uint32_t)(c0 >> 6);
                    }
                }
                a += 1;
                continue;
            }
            const int b = a + nb;
            const uint32_t rbytes = (uint32_t)__shfl((int)incl, b - 1) - start;
            if (rbytes) {
                char *dst = A.out + wave_off + start;
                const uint32_t mis = (uint32_t)((uintptr_t)dst & 15);
                const bool mine = lane >= a && lane < b && exists;
                Sink<true> s; s.g = nullptr; s.cur = lb + mis + (excl - start);
                wave_lds_sync();                                                 // the previous round's flush has read the buffer
                fused_emit_lines(W, P, ci, ncols, c0 + lane, p0, plast, p, mine, cnt0, sl0, wl, s, dump, rlo1, rhi1);
                wave_lds_sync();
                wave_flush_text(lds_text + lb + mis, dst, rbytes);              // LDS offset == global address (mod 16)
            }
            a = b;
        }
    }
}

// The waves k_mplp_fused put on its list (a line longer than its LDS line buffer): every line of such a wave is written
// straight to global memory at the offsets the fused kernel left in offs[] (lines the fused kernel already wrote are simply
// written again with the same bytes).
__global__ void __launch_bounds__(256) k_mplp_emit_listed(StaWinDev W, MplpDevPar P, const unsigned long long *__restrict__ offs, const uint2 *__restrict__ colinfo,
                                                          char *out, const uint32_t *__restrict__ list, unsigned long long n_list)
{
    const unsigned long long li = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (li >= n_list) return;
    const int lane = threadIdx.x & 63;
    const int64_t ncols = (int64_t)W.col_end - W.col_beg;
    const int64_t c0 = (int64_t)list[li] * 64;
    const int64_t c1 = c0 + 64 < ncols ? c0 + 64 : ncols;
    const int p0 = W.col_beg + (int)c0;
    const int p = p0 + lane;
    const bool active = p < W.col_end;
    const int plast = W.col_beg + (int)c1 - 1;
    const unsigned long long o1 = offs[c1];
    const unsigned long long my0 = active ? offs[c0 + lane] : o1;
    const unsigned long long my1 = active ? (lane == 63 || c0 + lane + 1 == c1 ? o1 : offs[c0 + lane + 1]) : o1;
    const bool exists = my1 > my0;
    Sink<false> s; s.cur = 0; s.g = out + my0;
    emit_column_fast<false>(W, P, colinfo, ncols, c0 + lane, p0, plast, p, exists, s, 0);
}

static MplpDevPar make_par(const sta_mplp_params &p, int64_t tlen)
{
    MplpDevPar d;
    d.min_baseQ = p.min_baseQ; d.all = p.all; d.rev_del = p.rev_del; d.flag = p.flag;
    d.no_ins = p.no_ins; d.no_del = p.no_del; d.no_ends = p.no_ends; d.tlen = tlen;
    d.n_tags = p.n_tags > 0 ? p.n_tags : 0; d.tag_sep = p.tag_sep ? p.tag_sep : ',';
    return d;
}

void sta_launch_mplp_len(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, uint32_t *line_len, uint2 *colinfo, StaCounters *ctr)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    int64_t nb = (ncols + 255) / 256;
    if (!((uint32_t)p.flag & EXTRA_MASK) && p.n_tags <= 0 && colinfo) {
        hipLaunchKernelGGL(k_mplp_len_fast, dim3((unsigned)nb), dim3(256), 0, s, w, make_par(p, w.tlen), line_len, colinfo, ctr);
        return;
    }
    hipLaunchKernelGGL(k_mplp_len, dim3((unsigned)nb), dim3(256), 0, s, w, make_par(p, w.tlen), line_len, ctr);
}

void sta_launch_mplp_emit(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, const uint64_t *offs, const uint2 *colinfo,
                          char *out, uint32_t lds_cap)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    uint32_t slice = (lds_cap + 16 + 15) & ~15u;
    // waves per workgroup so that the workgroup's LDS (one slice per wave) stays within 64 KiB
    int wpb = 4 * slice <= 65536 ? 4 : (2 * slice <= 65536 ? 2 : 1);
    int64_t nwaves = (ncols + 63) / 64;
    int64_t nb = (nwaves + wpb - 1) / wpb;
    if (!((uint32_t)p.flag & EXTRA_MASK) && p.n_tags <= 0 && colinfo) {
        uint32_t fslice = (lds_cap + 48 + 15) & ~15u;      // must match k_mplp_emit_fast
        int fw = 4 * fslice <= 65536 ? 4 : (2 * fslice <= 65536 ? 2 : 1);
        int64_t fnb = (nwaves + fw - 1) / fw;
        hipLaunchKernelGGL(k_mplp_emit_fast, dim3((unsigned)fnb), dim3(64 * fw), (size_t)fw * fslice, s, w, make_par(p, w.tlen), offs, colinfo, out, lds_cap);
        return;
    }
    hipLaunchKernelGGL(k_mplp_emit, dim3((unsigned)nb), dim3(64 * wpb), (size_t)wpb * slice, s, w, make_par(p, w.tlen), offs, out, lds_cap);
}

// LDS per wave of k_mplp_fused for a line buffer of `lbuf` bytes
static uint32_t fused_per_wave(uint32_t lbuf) { return FT_META + FT_TILE + ((lbuf + 16 + 15) & ~15u) + 16; }

size_t sta_mplp_fused_status_bytes(int64_t ncols)
{
    int64_t n_tiles = (ncols + 255) / 256;
    return (size_t)n_tiles * 16 + 16;          // 2 words per tile + the ticket
}

void sta_launch_mplp_fused(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, void *status, uint2 *colinfo, char *out,
                           uint64_t capacity, StaCounters *ctr, uint32_t lbuf, uint32_t *giant, unsigned long long *offs)
{
    int64_t ncols = (int64_t)w.col_end - w.col_beg;
    if (ncols <= 0) return;
    const int64_t n_tiles = (ncols + 255) / 256;
    hipMemsetAsync(status, 0, sta_mplp_fused_status_bytes(ncols), s);
    FusedArgs a;
    a.status = (unsigned long long *)status;
    a.ticket = (unsigned int *)((char *)status + (size_t)n_tiles * 16);
    a.out = out; a.capacity = capacity; a.colinfo = w.nfiles > 1 ? colinfo : nullptr; a.ctr = ctr;
    a.lbuf = lbuf; a.per_wave = fused_per_wave(lbuf); a.n_tiles = (uint32_t)n_tiles;
    a.giant = giant; a.offs = offs; a.colinfo_one = colinfo;
    // enough batches to keep every CU busy through the tail (~768 workgroups are resident), as few tickets as that allows
    int64_t tpb = 1;      // consecutive tiles per workgroup serialise the look-back chain (measured: 400x slower); kept as an experiment knob
    { static const char *ev = getenv("STA_FUSED_TPB"); if (ev && atoi(ev) > 0) tpb = atoi(ev); }
    a.tiles_per_batch = (uint32_t)tpb;
    const int64_t n_batches = (n_tiles + tpb - 1) / tpb;
    hipLaunchKernelGGL(k_mplp_fused, dim3((unsigned)n_batches), dim3(256), (size_t)4 * a.per_wave, s, w, make_par(p, w.tlen), a);
}

void sta_launch_mplp_emit_listed(hipStream_t s, const StaWinDev &w, const sta_mplp_params &p, const unsigned long long *offs, const uint2 *colinfo,
                                 char *out, const uint32_t *list, uint64_t n_list)
{
    if (!n_list) return;
    hipLaunchKernelGGL(k_mplp_emit_listed, dim3((unsigned)((n_list + 3) / 4)), dim3(256), 0, s, w, make_par(p, w.tlen), offs, colinfo, out, list, (unsigned long long)n_list);
}

// no --output-extra / -O / -s columns: the window can take the single-pass kernel
bool sta_mplp_has_fast_path(const sta_mplp_params &p) { return !((uint32_t)p.flag & EXTRA_MASK) && p.n_tags <= 0; }
