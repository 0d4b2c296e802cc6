// plp_entry.h -- one (read, column) entry of the pileup: CIGAR resolution, the token mpileup prints for it, the byte sinks.
//
// HTSlib resolve_cigar2 and samtools pileup_seq (bam_plcmd.c:54-169) as plain functions over the staged arrays,
// shared by the column kernels (kernels_plp.hip) and by the CPU harness of the tile kernels (tests/cpu/plp_emul.cpp: test
// infrastructure, not linked into the library).  On the device every function is __forceinline__ device code, as before.
#pragma once
#include "sta_dev.h"

#if defined(__HIPCC__)
#define PLP_HD __host__ __device__ __forceinline__
#else
#define PLP_HD inline
#endif

// the wave's text slice: LDS on the device; the harness points it at a plain array
#if defined(__HIPCC__)
extern __shared__ __attribute__((aligned(16))) char lds_text[];
#define PLP_LDS lds_text
#else
extern thread_local char *plp_host_lds;
#define PLP_LDS plp_host_lds
#endif

#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16

enum { CG_M = 0, CG_I, CG_D, CG_N, CG_S, CG_H, CG_P, CG_EQ, CG_X, CG_B };

PLP_HD bool cg_is_refop(int op) { return (0x18Du >> op) & 1; }   // M D N = X  -> bits 0,2,3,7,8
PLP_HD bool cg_is_mop(int op) { return (0x181u >> op) & 1; }     // M = X
PLP_HD bool cg_is_qop(int op) { return op == CG_I || op == CG_S; }

PLP_HD int dec_digits_u32(uint32_t v)
{
    return 1 + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u)
             + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}
PLP_HD int dec_digits(unsigned long long v)
{
    int n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}

// merged, sorted, disjoint intervals: does [beg,end) overlap any?  (bedidx.c:159-197 semantics)
PLP_HD bool bed_overlap_dev(const int64_t *bbeg, const int64_t *bend, int64_t n, int64_t beg, int64_t end)
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
PLP_HD int nt16_from_char(unsigned char c)
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

PLP_HD int seq_nib(const uint8_t *seq, uint64_t seq_byte0, int i)
{
    return (seq[seq_byte0 + (uint64_t)(i >> 1)] >> ((~i & 1) << 2)) & 0xf;
}

PLP_HD char lower_c(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }
PLP_HD char upper_c(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// a read reached bam_plp_push and was not dropped by the -d cap (it moved the iterator's max_pos)
PLP_HD bool read_advances_iterator(const StaReadsDev &R, int64_t j)
{
    uint32_t info = R.info[j];
    bool dropped = (info & RI_PUSHED) && !(info & RI_KEEP) && R.end[j] > R.pos[j];
    return (info & RI_PUSHED) && !dropped;
}

// Quality a deletion / ref-skip placeholder of read r shows at column p (bam_plcmd.c:676-679 reads qual[qpos] of the NEXT base).
// HTSlib resolves a mate pair when the second mate is pushed, and a column is handed out as soon as some read starting
// beyond it has been pushed -- so a column before the mate's start sees the resolved quality only if the mate itself is
// that first read.  Everything else about the overlap pass is order independent; this is the one place where it is not.
PLP_HD int placeholder_qual(const StaReadsDev &R, int64_t r, int qpos, int lq, uint64_t boff, int p)
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

struct MplpDevPar {
    int32_t min_baseQ, all, rev_del, flag, no_ins, no_del, no_ends;
    int32_t n_tags, tag_sep;
    int32_t mods, no_ins_mods;       // --output-mods: append StaReadsDev.mod_* text to modified bases (and to inserted ones unless no_ins_mods)
    int32_t mq_col;                  // -s on the tile path: a third string per file, the mapping-quality character of every entry that passed -Q (bam_plcmd.c:727-737)
    int64_t tlen;
};
#define TAGKIND (1 << 28)       // file_pass "kind" of tag column t is TAGKIND + t


// seq_nt16_table (hts.c) by arithmetic: character -> 4-bit code, 15 for anything that is not a nucleotide code letter, '=' or '0'..'3'
PLP_HD unsigned nt16_arith(unsigned char c)
{
    const unsigned li = (unsigned)(c | 32) - 'a';
    if (li < 16u) return (unsigned)(0xfff3fcffb4ffd2e1ull >> (4 * li)) & 15u;          // a..p
    if (li < 26u) return (unsigned)(0xfaf97f865full >> (4 * (li - 16u))) & 15u;        // q..z
    if (c == '=') return 0u;
    const unsigned di = (unsigned)c - '0';
    return di < 4u ? 1u << di : 15u;
}

// code -> character tables of bam_plcmd.c:75-84 / hts.c seq_nt16_str
PLP_HD char nt16_lc(int c) { return ",acmgrsvtwyhkdbn"[c]; }
PLP_HD char nt16_uc(int c) { return ".ACMGRSVTWYHKDBN"[c]; }
PLP_HD char nt16_chr(int c) { return "=ACMGRSVTWYHKDBN"[c]; }


// ---- byte sink: KIND 1 the LDS slice of this wave, 0 global memory byte by byte, 2 global memory eight bytes per store ----
// KIND 2 (round 5, the generic walker's rows that exceed the LDS slice): a lane collects the bytes of ONE string in a register pair and
// stores them eight at a time at the string's own (unaligned) addresses; flush() writes the last 0..7 with byte stores, so nothing outside
// the string is touched.  The lanes of a wave write different rows: a byte store is one L2 request per lane and byte, and at 1.8 GB of
// --output-extra text per window the request rate, not the bytes, was what the emit kernel waited for (profiles/r05_generic_walker.md).
typedef uint64_t __attribute__((aligned(1))) sink_u64u;
template <int KIND> struct Sink {
    uint32_t cur;        // LDS: offset into lds_text; global: unused
    char *g;             // global cursor (KIND 2: where the next eight bytes go)
    uint64_t acc;        // KIND 2: the bytes not stored yet, first byte lowest
    uint32_t nb;         // KIND 2: how many
    uint32_t dry;        // KIND 2, timing diagnostics only (STA_GENERIC_DIAG=4): the eight-byte stores are not executed
    PLP_HD void open(char *at) { g = at; cur = 0; acc = 0; nb = 0; dry = 0; }
    PLP_HD void put_n(uint64_t v, uint32_t n)         // KIND 2: n <= 8 bytes, first byte lowest, the bytes above n zero
    {
        acc |= v << (8 * nb);
        nb += n;
        if (nb >= 8) {
            if (!dry) *reinterpret_cast<sink_u64u *>(g) = acc;
            g += 8; nb -= 8;
            acc = nb ? v >> (8 * (n - nb)) : 0;
        }
    }
    PLP_HD void put_digits(uint32_t w, uint32_t n)      // KIND 2: the n <= 8 low decimal digits of w < 10^8 (32-bit divisions by a constant)
    {
        uint64_t t = 0;
        for (uint32_t i = 0; i < n; ++i) { t = (t << 8) | (uint64_t)('0' + w % 10u); w /= 10u; }
        put_n(t, n);
    }
    PLP_HD void flush()
    {
        if (KIND == 2) { for (uint32_t i = 0; i < nb; ++i) g[i] = (char)(acc >> (8 * i)); g += nb; nb = 0; acc = 0; }
    }
    PLP_HD void put(char c)
    {
        if (KIND == 1) PLP_LDS[cur++] = c;
        else if (KIND == 0) *g++ = c;
        else put_n((uint64_t)(unsigned char)c, 1);
    }
    PLP_HD void put_dec(long long v)
    {
        if (v < 0) { put('-'); v = -v; }
        unsigned long long u = (unsigned long long)v;
        if (KIND != 2 && u < 0x100000000ull) {
            // (a position or a count: 32-bit arithmetic -- a 64-bit division by ten is a dozen vector instructions per digit, twice
            // per digit with the counting loop, and the read-major kernel's row heads paid ~200 of them per strip)
            uint32_t w = (uint32_t)u;
            const int n = dec_digits_u32(w);
            if (KIND == 1) { uint32_t e = cur + n; for (uint32_t q = e; q > cur;) { const uint32_t d = w / 10u; PLP_LDS[--q] = (char)('0' + (w - d * 10u)); w = d; } cur = e; }
            else { char *e = g + n; for (char *q = e; q > g;) { const uint32_t d = w / 10u; *--q = (char)('0' + (w - d * 10u)); w = d; } g = e; }
            return;
        }
        int n = dec_digits(u);
        if (KIND == 1) {
            uint32_t e = cur + n;
            for (uint32_t q = e; q > cur;) { PLP_LDS[--q] = (char)('0' + u % 10); u /= 10; }
            cur = e;
        } else if (KIND == 0) {
            char *e = g + n;
            for (char *q = e; q > g;) { *--q = (char)('0' + u % 10); u /= 10; }
            g = e;
        } else {
            // groups of at most eight digits from the top (a group below the first one is zero-padded: its loop runs its full width)
            if (n > 16) { const unsigned long long h = u / 10000000000000000ull; u -= h * 10000000000000000ull; put_digits((uint32_t)h, (uint32_t)(n - 16)); n = 16; }
            if (n > 8) { const unsigned long long h = u / 100000000ull; u -= h * 100000000ull; put_digits((uint32_t)h, (uint32_t)(n - 8)); n = 8; }
            put_digits((uint32_t)u, (uint32_t)n);
        }
    }
};

struct Resolved {
    int qpos, indel, k;
    bool is_del, is_refskip;
};

// stateless equivalent of HTSlib resolve_cigar2 for (read, column p) -- SURVEY.md A.2
PLP_HD Resolved resolve_general(const uint32_t *cig, int n, int rpos, int p)
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

// --output-mods: the text HTSlib's bam_mods_at_qpos yields for query position qpos of read r ("[+m128]"), staged by the host
// (host_mods.cpp); returns its length (0: the base is not modified) and where it starts
PLP_HD int mod_text_at(const StaReadsDev &R, int64_t r, int qpos, uint32_t &t0)
{
    if (!R.mod_off) return 0;
    uint32_t lo = R.mod_off[r], hi = R.mod_off[r + 1];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t q = R.mod_qpos[mid];
        if (q == (uint32_t)qpos) { t0 = R.mod_toff[mid]; return (int)(R.mod_toff[mid + 1] - t0); }
        if (q < (uint32_t)qpos) lo = mid + 1; else hi = mid;
    }
    return 0;
}

// bam_plp_insertion: total length (I+P run after op k) and the D that may follow it
PLP_HD void insertion_shape(const uint32_t *cig, int n, int k, int &ins_total, int &del_after)
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

PLP_HD int token_len(const StaReadsDev &R, const MplpDevPar &P, const Entry &e, int p)
{
    int len = 1;
    if (!P.no_ends) len += (p == e.rpos ? 2 : 0) + (p == e.rend - 1 ? 1 : 0);
    uint32_t mt0;
    if (P.mods && !e.rs.is_del) len += mod_text_at(R, e.r, e.rs.qpos, mt0);
    if (e.rs.indel != 0) {
        int del_len = -e.rs.indel;
        if (e.rs.indel > 0) {
            const uint32_t *cig = R.cigar + R.cig_off[e.r];
            int n = (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]);
            int ins_total;
            insertion_shape(cig, n, e.rs.k, ins_total, del_len);
            if (P.no_ins < 2) len += 1 + dec_digits_u32((uint32_t)ins_total);
            if (!P.no_ins) {
                len += ins_total;
                if (P.mods && !P.no_ins_mods) {
                    // modification text of the inserted bases (bam_plp_insertion_mod)
                    int j = 1;
                    for (int kk = e.rs.k + 1; kk < n; ++kk) {
                        int o = cig[kk] & 0xf, l = (int)(cig[kk] >> 4);
                        if (o == CG_I) { for (int t = 0; t < l; ++t, ++j) len += mod_text_at(R, e.r, e.rs.qpos + j - (e.rs.is_del ? 1 : 0), mt0); }
                        else if (o != CG_P) break;
                    }
                }
            }
        }
        if (del_len > 0) {
            if (P.no_del < 2) len += 1 + dec_digits_u32((uint32_t)del_len);
            if (!P.no_del) len += del_len;
        }
    }
    return len;
}

template <int LDS>
PLP_HD void token_write(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, const Entry &e, int p, Sink<LDS> &s)
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
        s.put(rev ? nt16_lc(c) : nt16_uc(c));
        if (P.mods) { uint32_t t0; const int ml = mod_text_at(R, e.r, e.rs.qpos, t0); for (int t = 0; t < ml; ++t) s.put(R.mod_text[t0 + t]); }
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
                            char ch = qi < e.lq ? nt16_chr(seq_nib(R.seq, e.boff >> 1, qi)) : 'N';
                            s.put(rev ? lower_c(ch) : upper_c(ch));
                            if (P.mods && !P.no_ins_mods) { uint32_t t0; const int ml = mod_text_at(R, e.r, qi, t0); for (int t2 = 0; t2 < ml; ++t2) s.put(R.mod_text[t0 + t2]); }
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

