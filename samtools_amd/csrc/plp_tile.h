// plp_tile.h -- step functions of the two tile kernels of the mpileup text path (kernels_plp.hip):
//
//   k_mplp_len_rm    measuring pass, READ-major.  A workgroup owns LEN_TC consecutive columns.  The depth of a column is a
//                    prefix sum of +1 / -1 marks at read starts / ends, the reads whose base fails -Q are found by scanning
//                    the quality bytes sixteen at a time (SWAR compare) and subtracted per column, '^x' / '$' bytes are added
//                    at a read's first / last column: O(bases / 16) vector steps instead of one step per (read, 64 columns).
//                    (what bam_plcmd.c:669-725 counts per column: n_plp, the post -Q count, the base string length)
//   k_mplp_emit_tile emit pass.  A wave owns 64 columns, as before, but the per-base work (quality test, base character,
//                    strand case, reference match, quality character) is done READ-major: sixteen reads at a time are converted
//                    into a 16 x 64 byte tile in LDS, sixteen columns per lane with packed byte arithmetic; then every lane
//                    (= column) walks down its tile column and appends the non-zero bytes at its two cursors.
//                    (bam_plcmd.c:54-169 pileup_seq for reads that are one M op; everything else takes token_write)
//
// Every function here is plain per-thread code over explicit "LDS" pointers, so that tests/cpu/plp_emul.cpp (test
// infrastructure, not linked into the library) runs the very same functions in loops over the lanes and diffs the text against
// the oracle without a GPU.  Cross-lane steps (ballots, the 64-ary searches, the flush) stay in kernels_plp.hip.
#pragma once
#include "plp_entry.h"
#include "deep_strip.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define PLP_LDS_ADD(p, v) atomicAdd((p), (v))
#define PLP_PERM(hi, lo, sel) __builtin_amdgcn_perm((hi), (lo), (sel))
#else
static inline int plp_add_host(int *p, int v) { const int old = *p; *p = old + v; return old; }      // atomicAdd's return value
#define PLP_LDS_ADD(p, v) plp_add_host((p), (v))
static inline uint32_t plp_perm_host(uint32_t hi, uint32_t lo, uint32_t sel)
{
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = (sel >> (8 * i)) & 0xff;           // only selectors 0..7 are used here
        r |= (uint32_t)((src >> (8 * (s & 7))) & 0xff) << (8 * i);
    }
    return r;
}
#define PLP_PERM(hi, lo, sel) plp_perm_host((hi), (lo), (sel))
#endif
// a branch every lane of the wave takes together: the device passes a ballot, the harness runs it both ways (always / per lane)
#ifndef PLP_WAVE_ANY
#define PLP_WAVE_ANY(x) (__ballot(x) != 0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// packed byte arithmetic (four bytes per 32-bit word)

// 0x80 in every byte of x that is >= m (m4 = m replicated, 0 <= m <= 127)
PLP_HD uint32_t swar_ge_u8(uint32_t x, uint32_t m4) { return ((((x | 0x80808080u) - m4)) | x) & 0x80808080u; }
// 0x80 flags -> 0xff bytes
PLP_HD uint32_t swar_expand80(uint32_t f) { return (f - (f >> 7)) | f; }
// bytes [lo, hi) of a word as 0xff (any lo, hi: clamped to the word)
PLP_HD uint32_t swar_byte_range(int lo, int hi)
{
    lo = lo < 0 ? 0 : lo; hi = hi > 4 ? 4 : hi;
    if (hi <= lo) return 0u;
    const uint32_t upto_hi = hi >= 4 ? 0xffffffffu : ((1u << (8 * hi)) - 1u);
    const uint32_t upto_lo = (1u << (8 * lo)) - 1u;          // lo <= 3 here
    return upto_hi & ~upto_lo;
}
// min(q + 33, 126) per byte (bam_plcmd.c:687), any byte value
PLP_HD uint32_t swar_qual_chars(uint32_t q)
{
    const uint32_t big = swar_expand80(swar_ge_u8(q, 0x5e5e5e5eu));        // q >= 94
    return ((q & ~big) + (0x21212121u & ~big)) | (big & 0x7e7e7e7eu);      // (bytes of 94 and more are taken out before the add: no carries)
}
// four 4-bit codes (one per byte) -> characters of bam_plcmd.c:75-84: ".ACMGRSVTWYHKDBN" forward, ",acmgrsvtwyhkdbn" reverse
PLP_HD uint32_t swar_base_chars(uint32_t codes, bool rev)
{
    const uint32_t t0 = rev ? 0x6d63612cu : 0x4d43412eu;      // , a c m   /  . A C M
    const uint32_t t1 = rev ? 0x76737267u : 0x56535247u;      // g r s v   /  G R S V
    const uint32_t t2 = rev ? 0x68797774u : 0x48595754u;      // t w y h   /  T W Y H
    const uint32_t t3 = rev ? 0x6e62646bu : 0x4e42444bu;      // k d b n   /  K D B N
    const uint32_t idx = codes & 0x07070707u;
    const uint32_t lo = PLP_PERM(t1, t0, idx), hi = PLP_PERM(t3, t2, idx);
    const uint32_t m = swar_expand80((codes << 4) & 0x80808080u);           // bit 3 of the code picks the upper half of the table
    return (hi & m) | (lo & ~m);
}
// nibbles 4j .. 4j+3 of a 16-nibble stream, one per byte
PLP_HD uint32_t swar_spread_nibbles(uint64_t nib, int j)
{
    uint32_t x = (uint32_t)(nib >> (16 * j)) & 0xffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    return (x | (x << 4)) & 0x0f0f0f0fu;
}

// One read (a single M op) in one chunk of sixteen consecutive tile columns.
//   q4 / s4: the 16 quality bytes / 12 packed-base bytes from query index qb on (qb = query index shown in chunk column d0)
//   d0: first covered chunk column, ncov: covered columns (>= 1), rbpack / has_ref: reference codes of the chunk (deep_strip.h)
//   minq4: -Q replicated into four bytes (0..127); head_col / tail_col: chunk column of the read's first / last base, or -1
// Result: byte k of tb / tq = what column k appends to its base / quality string, 0 where the read shows nothing there (not
// covered, or quality below -Q).  Bit 7 of a tq byte: the token starts with '^' + mapping quality; bit 7 of a tb byte: it ends with '$'.
PLP_HD void tile_convert16(const uint32_t q4[4], const uint32_t s4[3], int qb, int d0, int ncov, uint64_t rbpack, bool has_ref,
                           uint32_t minq4, bool rev, int head_col, int tail_col, uint32_t tb[4], uint32_t tq[4])
{
    uint32_t qs[4];
    deep_shift_quals(q4, d0, qs);
    const uint64_t nib = deep_shift_bases(s4, qb, d0, rbpack, has_ref);
    const uint32_t vm16 = ((1u << (d0 + ncov)) - 1u) & ~((1u << d0) - 1u);          // bit k: chunk column k is covered (d0 + ncov <= 16)
    // quality characters: qualities below 64 (everything a sequencer writes) take one packed add; anything else the general form
    const bool plain_q = ((qs[0] | qs[1] | qs[2] | qs[3]) & 0xc0c0c0c0u) == 0u;
    const bool all_plain = !PLP_WAVE_ANY(!plain_q);
    const int hsel = head_col >> 2, tsel = tail_col >> 2;                          // (-1 >> 2 = -1: no word)
    const uint32_t hbit = 0x80u << (8 * (head_col & 3)), tbit = 0x80u << (8 * (tail_col & 3));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // bit i of the nibble -> bit 7 of byte i: the products of the three low bits with 2^7, 2^14, 2^21 fall on distinct positions
        const uint32_t x = (vm16 >> (4 * j)) & 15u;
        const uint32_t valid = ((x * 0x204080u) | (x << 28)) & 0x80808080u;
        const uint32_t pass = swar_expand80(swar_ge_u8(qs[j], minq4) & valid);
        uint32_t qc = all_plain ? qs[j] + 0x21212121u : swar_qual_chars(qs[j]);
        uint32_t bc = swar_base_chars(swar_spread_nibbles(nib, j), rev);
        qc |= hsel == j ? hbit : 0u;
        bc |= tsel == j ? tbit : 0u;
        tq[j] = qc & pass;
        tb[j] = bc & pass;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_mplp_len_rm

#define LEN_TC 1024             // columns per workgroup
#define LEN_THREADS 256         // = reads per batch

struct LenLds {
    int diff[LEN_TC + 4];       // +1 at the first covered tile column of a read, -1 behind its last
    int fail[LEN_TC];           // entries of the column whose quality is below -Q
    int extra[LEN_TC];          // base-string bytes beyond one per passing entry ('^x', '$', indel text)
    int m_pos[LEN_THREADS], m_end[LEN_THREADS];
    uint32_t m_b8[LEN_THREADS];
    int m_kind[LEN_THREADS];    // 0: nothing, 1: one M op (quality scan), 2: general CIGAR (listed)
    int glist[LEN_THREADS];
    int gcount;
    int part[LEN_THREADS], part2[LEN_THREADS / 16 + 1];
    long long rlo, rhi;
    // the tile's place in the window's text (the kernel's own look-back over the tiles, kernels_plp.hip)
    unsigned tile; unsigned n_lines, n_data; unsigned long long tile_bytes, wave_max, ex_bytes;
};

PLP_HD void len_clear(LenLds &L, int t)
{
    for (int i = t; i < LEN_TC + 4; i += LEN_THREADS) L.diff[i] = 0;
    for (int i = t; i < LEN_TC; i += LEN_THREADS) { L.fail[i] = 0; L.extra[i] = 0; }
    if (t == 0) L.gcount = 0;
}

// step A: thread t looks at read b0 + t of the tile's range: depth marks, '^x' / '$' bytes, what kind of walk it needs
PLP_HD void len_step_a(LenLds &L, int t, const StaReadsDev &R, const MplpDevPar &P, int t0, int t1, long long b0)
{
    const long long r = b0 + t;
    int kind = 0;
    if (r < L.rhi) {
        const uint32_t info = R.info[r];
        const int pos = R.pos[r], end = R.end[r];
        if ((info & RI_KEEP) && end > t0 && pos < t1) {
            const int a = (pos > t0 ? pos : t0) - t0, b = (end < t1 ? end : t1) - t0;
            PLP_LDS_ADD(&L.diff[a], 1);
            PLP_LDS_ADD(&L.diff[b], -1);
            const uint32_t b8 = R.base_off8[r];
            if (info & RI_SIMPLE) {
                kind = 1;
                if (!P.no_ends) {
                    const uint64_t boff = (uint64_t)b8 << 3;
                    if (pos >= t0 && (int)R.qual[boff] >= P.min_baseQ) PLP_LDS_ADD(&L.extra[pos - t0], 2);
                    if (end <= t1 && (int)R.qual[boff + (uint64_t)(end - 1 - pos)] >= P.min_baseQ) PLP_LDS_ADD(&L.extra[end - 1 - t0], 1);
                }
            } else {
                kind = 2;
                L.glist[PLP_LDS_ADD(&L.gcount, 1)] = t;
            }
            L.m_pos[t] = pos; L.m_end[t] = end; L.m_b8[t] = b8;
        }
    }
    L.m_kind[t] = kind;
}

// step B: the quality bytes of the batch's one-op reads, sixteen per step; four lanes share a read
PLP_HD void len_step_b(LenLds &L, int t, const StaReadsDev &R, const MplpDevPar &P, int t0, int t1)
{
    if (P.min_baseQ <= 0) return;                         // nothing can fail
    const uint32_t minq4 = (uint32_t)P.min_baseQ * 0x01010101u;
    const int g = t & 3;
#pragma unroll 1
    for (int sub = 0; sub < LEN_THREADS / 64; ++sub) {
        const int slot = sub * 64 + (t >> 2);
        if (L.m_kind[slot] != 1) continue;
        const int pos = L.m_pos[slot], end = L.m_end[slot];
        const uint64_t boff = (uint64_t)L.m_b8[slot] << 3;
        const int qa = (pos > t0 ? pos : t0) - pos, qe = (end < t1 ? end : t1) - pos;       // query indices [qa, qe) lie in the tile
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
                    PLP_LDS_ADD(&L.fail[pos + w0 + i - t0], 1);
                    f &= f - 1;
                }
            }
        }
    }
}

// step C: a read with a general CIGAR, one (read, column) pair per lane and step (lane of nlanes cooperating on the list entry gi)
PLP_HD void len_step_c(LenLds &L, int gi, int lane, int nlanes, long long b0, const StaReadsDev &R, const MplpDevPar &P, int t0, int t1)
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
        if (c < P.min_baseQ) PLP_LDS_ADD(&L.fail[p - t0], 1);
        else {
            const int tl = token_len(R, P, e, p);
            if (tl != 1) PLP_LDS_ADD(&L.extra[p - t0], tl - 1);
        }
    }
}

// prefix sum of the depth marks over the tile: four phases with a barrier between them (thread t owns columns 4t .. 4t+3)
PLP_HD void len_scan_1(LenLds &L, int t) { L.part[t] = L.diff[4 * t] + L.diff[4 * t + 1] + L.diff[4 * t + 2] + L.diff[4 * t + 3]; }
PLP_HD void len_scan_2(LenLds &L, int t)
{
    if (t < LEN_THREADS / 16) { int s = 0; for (int i = 0; i < 16; ++i) s += L.part[16 * t + i]; L.part2[t] = s; }
}
PLP_HD void len_scan_3(LenLds &L, int t)
{
    if (t == 0) { int run = 0; for (int i = 0; i < LEN_THREADS / 16; ++i) { const int s = L.part2[i]; L.part2[i] = run; run += s; } }
}
// depth before column 4t
PLP_HD int len_scan_4(const LenLds &L, int t)
{
    int s = L.part2[t >> 4];
    for (int i = t & ~15; i < t; ++i) s += L.part[i];
    return s;
}

// per-file result of thread t's four columns: (count after -Q, base string bytes) -> colinfo; adds the file's text bytes to total[]
// (bytes of "\t cnt \t seq \t qual" as bam_plcmd.c:699-725 prints them), any[] |= the column has entries before -Q
PLP_HD void len_file_result(const LenLds &L, int t, int depth_before, int ncols_tile, uint2 *colinfo_tile, uint32_t total[4], bool any[4], bool mq_col = false,
                            uint32_t *cnt_out = nullptr /* [4]: the columns' counts after -Q, for the caller that adds extra columns */)
{
    int d = depth_before;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * t + i;
        d += L.diff[c];
        if (c >= ncols_tile) continue;
        const uint32_t cnt = (uint32_t)(d - L.fail[c]);
        const uint32_t seq_len = cnt + (uint32_t)L.extra[c];
        any[i] |= d > 0;
        total[i] += 1 + (uint32_t)dec_digits_u32(cnt) + 1 + (seq_len ? seq_len : 1) + 1 + (cnt ? cnt : 1);
        if (mq_col) total[i] += 1 + (cnt ? cnt : 1);          // -s: "\t" + one mapping-quality character per entry, or '*'

        colinfo_tile[c] = make_uint2(cnt, seq_len);
        if (cnt_out) cnt_out[i] = cnt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_mplp_emit_tile

#define TILE_SLOTS 16           // reads converted per round
#define TILE_STRIDE 80          // bytes per tile row: 64 columns + 16 zero bytes (the column lanes without entries look at)
#define TILE_ZERO_COL 64
// the wave's LDS behind its text slice
struct TileLds {
    uint8_t tb[TILE_SLOTS * TILE_STRIDE];       // base characters (bit 7: '$' follows)
    uint8_t tq[TILE_SLOTS * TILE_STRIDE];       // quality characters (bit 7: '^' + mapping quality first)
    int s_pos[TILE_SLOTS], s_end[TILE_SLOTS];   // the round's reads: the next sixteen live reads of the block of 64, in file order
    uint32_t s_b8[TILE_SLOTS], s_info[TILE_SLOTS];
    uint8_t s_lane[TILE_SLOTS];                 // their lane (= index in the block)
    long long s_ridx[TILE_SLOTS];               // their read index
    uint8_t s_mq[TILE_SLOTS];                   // their '^' character
    uint16_t s_plain[TILE_SLOTS * 4];           // reads with a general CIGAR: bit c of word (slot, chunk) = column 16 chunk + c was converted as plain
    uint8_t s_ref[64];                          // 4-bit reference code per column (0xff: none)
    uint64_t s_refpack[4];                      // the same, sixteen nibbles per chunk
};
#define TILE_LDS_BYTES ((sizeof(TileLds) + 15) & ~(size_t)15)

// decimal digits of a value below 2^32 at PLP_LDS[cur ...]; returns the cursor behind them
PLP_HD uint32_t lds_put_u32(uint32_t cur, uint32_t v)
{
    const int n = dec_digits_u32(v);
    uint32_t e = cur + (uint32_t)n;
    for (uint32_t q = e; q > cur;) { const uint32_t d = v / 10u; PLP_LDS[--q] = (char)('0' + (v - d * 10u)); v = d; }
    return e;
}

// what a lane (= column) carries through the kernel
struct TileLane {
    uint32_t cur;               // where the row's next fixed field goes
    uint32_t cur_s, cur_q;      // cursors of the current file's base string / quality string
    uint32_t cnt, sl;           // the current file's count and base-string bytes (1 for the '*' placeholder)
    uint32_t mq_d;              // -s: distance from the quality cursor to the mapping-quality string's cursor (count + 1); 0 = no such column
    int col;                    // tile column the lane reads in phase 2 (TILE_ZERO_COL: it has no entries to append)
    bool exists, walk;
};

// "name \t position \t reference base" of the lane's row (mpileup(), bam_plcmd.c:663-667); its reference code for the tile conversion
PLP_HD void tile_row_head(TileLds &T, TileLane &st, int lane, const StaWinDev &W, uint32_t row_start, bool exists, int64_t apos)
{
    const bool has_ref = W.ref != nullptr;
    uint32_t cur = row_start;
    int rbcode = 0xff;
    if (exists) {
        for (int t = 0; t < W.tname_len; ++t) PLP_LDS[cur++] = W.tname[t];
        PLP_LDS[cur++] = '\t';
        if ((uint64_t)(apos + 1) < 0x100000000ull) cur = lds_put_u32(cur, (uint32_t)(apos + 1));
        else { Sink<true> s; s.g = nullptr; s.cur = cur; s.put_dec(apos + 1); cur = s.cur; }
        PLP_LDS[cur++] = '\t';
        const char rc = (has_ref && apos < W.ref_len) ? W.ref[apos] : 'N';
        PLP_LDS[cur++] = rc;
        if (has_ref) rbcode = apos < W.ref_len ? (int)nt16_arith((unsigned char)rc) : 15;
    }
    T.s_ref[lane] = (uint8_t)rbcode;
    st.cur = cur; st.exists = exists; st.cur_s = st.cur_q = 0; st.cnt = 0; st.sl = 1; st.col = TILE_ZERO_COL; st.walk = false; st.mq_d = 0;
}
// the sixteen zero bytes behind every tile row (lanes 0 .. 2 TILE_SLOTS - 1, once)
PLP_HD void tile_zero_column(TileLds &T, int lane)
{
    uint8_t *row = (lane < TILE_SLOTS ? T.tb : T.tq) + (lane & (TILE_SLOTS - 1)) * TILE_STRIDE + 64;
    const uint32_t z[4] = { 0, 0, 0, 0 };
    __builtin_memcpy(row, z, 16);
}
// "\t count \t" of one file and where its two strings start (bam_plcmd.c:699-725)
PLP_HD void tile_file_head(TileLane &st, int lane, uint2 ci, uint32_t dump, bool mq_col = false)
{
    st.cnt = ci.x; st.sl = ci.y ? ci.y : 1u;
    if (st.exists) {
        PLP_LDS[st.cur++] = '\t';
        st.cur = lds_put_u32(st.cur, st.cnt);
        PLP_LDS[st.cur++] = '\t';
    }
    st.walk = st.exists && st.cnt;
    st.cur_s = st.walk ? st.cur : dump;
    st.cur_q = st.walk ? st.cur + st.sl + 1 : dump;
    st.col = st.walk ? lane : TILE_ZERO_COL;
    // (the mapping-quality string lies behind the quality string and its tab; a lane without entries writes next to the dump byte)
    st.mq_d = mq_col ? (st.walk ? st.cnt + 1u : 1u) : 0u;
}
// separators and the '*' placeholders go in AFTER the walk (its last predicated write may sit on them)
PLP_HD void tile_file_tail(TileLane &st)
{
    if (!st.exists) return;
    const bool mq = st.mq_d != 0;
    if (!st.cnt) {
        PLP_LDS[st.cur] = '*'; PLP_LDS[st.cur + 1] = '\t'; PLP_LDS[st.cur + 2] = '*';
        if (mq) { PLP_LDS[st.cur + 3] = '\t'; PLP_LDS[st.cur + 4] = '*'; }
    } else {
        PLP_LDS[st.cur + st.sl] = '\t';
        if (mq) PLP_LDS[st.cur + st.sl + 1 + st.cnt] = '\t';
    }
    st.cur += st.sl + 1 + (st.cnt ? st.cnt : 1u);
    if (mq) st.cur += 1 + (st.cnt ? st.cnt : 1u);
}
PLP_HD bool tile_read_is_live(uint32_t info, int pos, int end, int p0, int plast) { return (info & RI_KEEP) && end > p0 && pos <= plast; }
PLP_HD void tile_set_slot(TileLds &T, int i, int lane, long long ridx, uint32_t info, int pos, int end, uint32_t b8)
{
    T.s_pos[i] = pos; T.s_end[i] = end; T.s_b8[i] = b8; T.s_info[i] = info; T.s_lane[i] = (uint8_t)lane; T.s_ridx[i] = ridx;
}

// 16 reference codes (bytes, 0xff = no reference) -> nibble k = code of column k; lanes 0..3 run this for their chunk
PLP_HD void tile_refpack(TileLds &T, int k)
{
    uint64_t p = 0;
    for (int i = 0; i < 16; ++i) p |= (uint64_t)(T.s_ref[16 * k + i] & 15u) << (4 * i);
    T.s_refpack[k] = p;
}

// phase 1: lane = (slot, chunk): the round's read `slot` (of nslots) in tile columns [16 chunk, 16 chunk + 16)
// Returns whether the lane's slot holds a one-op read (phase 2 walks its tile row) -- the kernel ballots it into a slot mask.
PLP_HD bool tile_phase1(TileLds &T, int lane, int nslots, const StaReadsDev &R, const MplpDevPar &P, int p0, bool has_ref)
{
    const int slot = lane >> 2, k = lane & 3;
    uint32_t tb[4] = { 0, 0, 0, 0 }, tq[4] = { 0, 0, 0, 0 };
    const int idx = slot;
    bool simple = false;
    if (idx < nslots) {
        const uint32_t info = T.s_info[idx];
        const int pos = T.s_pos[idx], end = T.s_end[idx];
        const int c_lo = 16 * k;
        const int a = pos - p0 > c_lo ? pos - p0 : c_lo, b = end - p0 < c_lo + 16 ? end - p0 : c_lo + 16;
        // query index shown in the chunk's first covered column, or -1: the chunk is not plain for this read
        int qb = -1;
        if (info & RI_SIMPLE) { simple = true; if (b > a) qb = p0 + a - pos; }
        else if (b > a) {
            // A read with indels / clips / skips is still PLAIN INSIDE THIS CHUNK when its covered columns all fall in one M/=/X op and
            // the last of them carries no indel token (resolve_general: only the op's last column can, and only before D / I / P):
            // then qpos = column - (x - y) exactly as for a one-op read, '^' / '$' included.  Everything else stays with token_write.
            const long long r = T.s_ridx[idx];
            const uint32_t *cig = R.cigar + R.cig_off[r];
            const int n = (int)(R.cig_off[r + 1] - R.cig_off[r]), lq = R.l_qseq[r];
            const int ca = p0 + a, cb = p0 + b - 1;
            int x = pos, y = 0;
            for (int kk = 0; kk < n; ++kk) {
                const uint32_t c = cig[kk];
                const int op = (int)(c & 0xf), l = (int)(c >> 4);
                if (cg_is_refop(op)) {
                    if (ca < x + l) {
                        if (cg_is_mop(op) && cb <= x + l - 1 && y + l <= lq) {
                            bool quiet = cb < x + l - 1 || kk + 1 >= n;
                            if (!quiet) { const int op2 = (int)(cig[kk + 1] & 0xf); quiet = op2 != CG_D && op2 != CG_I && op2 != CG_P; }
                            if (quiet) qb = y + (ca - x);
                        }
                        break;
                    }
                    if (cg_is_mop(op)) y += l;
                    x += l;
                } else if (cg_is_qop(op)) y += l;
            }
        }
        uint32_t plain16 = 0;
        if (qb >= 0) {
            {
                const int d0 = a - c_lo;
                if (!simple) plain16 = ((1u << (d0 + (b - a))) - 1u) & ~((1u << d0) - 1u);
                const uint64_t boff = (uint64_t)T.s_b8[idx] << 3;
                uint32_t q4[4] = { 0, 0, 0, 0 }, s4[3] = { 0, 0, 0 };
                const uint64_t qa = boff + (uint64_t)qb, sa = (boff >> 1) + (uint64_t)(qb >> 1);
                if (qa + 16 <= R.n_bases_total) __builtin_memcpy(q4, R.qual + qa, 16);
                else for (int t = 0; t < 16 && qa + t < R.n_bases_total; ++t) q4[t >> 2] |= (uint32_t)R.qual[qa + t] << (8 * (t & 3));
                if (sa + 12 <= (R.n_bases_total >> 1)) __builtin_memcpy(s4, R.seq + sa, 12);
                else for (int t = 0; t < 12 && sa + t < (R.n_bases_total >> 1); ++t) s4[t >> 2] |= (uint32_t)R.seq[sa + t] << (8 * (t & 3));
                int hc = -1, tc = -1;
                if (!P.no_ends) {
                    const int h = pos - p0 - c_lo, tl = end - 1 - p0 - c_lo;
                    if (h >= 0 && h < 16) hc = h;
                    if (tl >= 0 && tl < 16) tc = tl;
                }
                tile_convert16(q4, s4, qb, d0, b - a, T.s_refpack[k], has_ref, (uint32_t)P.min_baseQ * 0x01010101u, (info & RI_REV) != 0, hc, tc, tb, tq);
            }
        }
        T.s_plain[slot * 4 + k] = (uint16_t)plain16;
        if (k == 0) { const int mq = (int)((info >> RI_MAPQ_SHIFT) & 0xff); T.s_mq[slot] = (uint8_t)(mq > 93 ? 126 : mq + 33); }
    }
    __builtin_memcpy(&T.tb[slot * TILE_STRIDE + 16 * k], tb, 16);
    __builtin_memcpy(&T.tq[slot * TILE_STRIDE + 16 * k], tq, 16);
    return simple;
}

// phase 2: the lane of column `col` (TILE_ZERO_COL for a lane without entries) appends what a tile row shows there.
// cur_s / cur_q: the lane's cursors into the wave's text (base string, quality string); a lane that appends nothing still
// writes at its cursors without advancing them -- the bytes are overwritten by its next real write or by the separators.
// b, q: the row's tile bytes of the column; mq: the row's '^' character.
PLP_HD void tile_phase2_apply(uint32_t b, uint32_t q, uint32_t mq, uint32_t &cur_s, uint32_t &cur_q, uint32_t mq_d = 0)
{
    const uint32_t pass = q != 0 ? 1u : 0u, hd = q >> 7, tl = b >> 7;
    if (PLP_WAVE_ANY(hd)) {
        PLP_LDS[cur_s] = '^';
        PLP_LDS[cur_s + hd] = (char)mq;
        cur_s += 2 * hd;
    }
    PLP_LDS[cur_s] = (char)(b & 0x7f);
    cur_s += pass;
    if (PLP_WAVE_ANY(tl)) {
        PLP_LDS[cur_s] = '$';
        cur_s += tl;
    }
    PLP_LDS[cur_q] = (char)(q & 0x7f);
    if (mq_d) PLP_LDS[cur_q + mq_d] = (char)mq;                // (-s is a window's option: every lane or none)
    cur_q += pass;
}
PLP_HD void tile_phase2_row(const TileLds &T, int slot, int col, uint32_t &cur_s, uint32_t &cur_q, uint32_t mq_d = 0)
{
    tile_phase2_apply(T.tb[slot * TILE_STRIDE + col], T.tq[slot * TILE_STRIDE + col], T.s_mq[slot], cur_s, cur_q, mq_d);
}
// four consecutive rows (slot a multiple of four): every tile byte is asked for before the first is used
PLP_HD void tile_phase2_rows4(const TileLds &T, int slot, int col, uint32_t &cur_s, uint32_t &cur_q, uint32_t mq_d = 0)
{
    uint32_t b[4], q[4], mq4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = T.tb[(slot + i) * TILE_STRIDE + col]; q[i] = T.tq[(slot + i) * TILE_STRIDE + col]; }
    __builtin_memcpy(&mq4, &T.s_mq[slot], 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) tile_phase2_apply(b[i], q[i], (mq4 >> (8 * i)) & 0xffu, cur_s, cur_q, mq_d);
}

// phase 2 for a read with a general CIGAR (indels, clips, pads, skips): per-entry resolution, as k_mplp_emit does it
PLP_HD void tile_phase2_general(const TileLds &T, int slot, TileLane &st, const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int64_t b0, int p, int p0)
{
    const int rpos = T.s_pos[slot], rend = T.s_end[slot];
    const int c = p - p0;                                       // the lane's tile column: plain columns were converted by phase 1
    const bool cov = st.walk && p >= rpos && p < rend && !((T.s_plain[slot * 4 + (c >> 4)] >> (c & 15)) & 1u);
    if (!PLP_WAVE_ANY(cov)) return;
    if (!cov) return;
    Entry e;
    e.r = b0 + (int64_t)T.s_lane[slot]; e.rpos = rpos; e.rend = rend; e.info = T.s_info[slot];
    e.lq = R.l_qseq[e.r];
    e.boff = (uint64_t)T.s_b8[slot] << 3;
    e.rs = resolve_general(R.cigar + R.cig_off[e.r], (int)(R.cig_off[e.r + 1] - R.cig_off[e.r]), rpos, p);
    const int qc = e.rs.is_del ? placeholder_qual(R, e.r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? (int)R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
    if (qc < P.min_baseQ) return;
    Sink<true> ss; ss.g = nullptr; ss.cur = st.cur_s;
    token_write<true>(R, W, P, e, p, ss);
    st.cur_s = ss.cur;
    if (st.mq_d) PLP_LDS[st.cur_q + st.mq_d] = (char)T.s_mq[slot];
    PLP_LDS[st.cur_q++] = (char)(qc + 33 < 126 ? qc + 33 : 126);
}
// a read with a general CIGAR: its plain chunks through the tile row, the remaining columns through token_write (a column is one or the other)
PLP_HD void tile_phase2_mixed(const TileLds &T, int slot, TileLane &st, const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int64_t b0, int p, int p0)
{
    tile_phase2_row(T, slot, st.col, st.cur_s, st.cur_q, st.mq_d);
    tile_phase2_general(T, slot, st, R, W, P, b0, p, p0);
}
