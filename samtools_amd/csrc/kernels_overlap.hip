// kernels_overlap.hip -- read-name pairing on the device (gfx950).
//
// mpileup: HTSlib overlap_push / overlap_remove / tweak_overlap_quality (sam.c, absent from the
//          reference tree; SURVEY.md A.3 + A.3.1; enabled at bam_plcmd.c:586).
// depth -s: the name->end hash of fastdepth_core (bam2depth.c:598-623).
//
// The reference keeps a qname hash that is mutated in arrival order.  Here every read that can
// touch the hash is inserted into an open-addressing table keyed by a 64-bit name hash; reads
// with the same key are linked through `chain_next`.  The member with the smallest read index of
// each exact-name group (the "leader" thread) then replays the reference's state machine over its
// group in file order -- groups are independent, so this is race free -- and applies the mate
// quality rewrite in place on the working quality pool.
#include "dev_util.h"

struct NameSlot { unsigned long long key; unsigned int head; unsigned int pad; };

size_t sta_overlap_table_slots(int64_t n_reads)
{
    size_t s = 1024;
    while (s < (size_t)n_reads * 2) s <<= 1;
    return s;
}

// ---- read names by aligned words ----
// A byte loop over a name is one dependent memory round trip per byte (hashing a seven-byte name twice, comparing it with its mate's and taking the
// keeper's hash were ~35 of the ~55 round trips of a k_name_groups wave: 0.28 ms at bench size with three waves per SIMD).  name_words16() brings
// sixteen bytes at a time: five ALIGNED word loads issued together (a word that holds at least one byte of the name lies inside the pool's
// last page whatever follows it), funnel-shifted to the name's first byte, zero behind its last.
__device__ __forceinline__ void name_words16(const char *names, uint32_t n0, int l, int off, uint32_t w[4])
{
    const uintptr_t p = (uintptr_t)(names + n0) + (uintptr_t)off;
    const uint32_t *a = (const uint32_t *)(p & ~(uintptr_t)3);
    const int sh = (int)(p & 3), rem = l - off;
    uint32_t d[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = (4 * k - sh < rem) ? a[k] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t v = sh ? (d[k] >> (8 * sh)) | (d[k + 1] << (32 - 8 * sh)) : d[k];
        const int left = rem - 4 * k;
        w[k] = left >= 4 ? v : left <= 0 ? 0u : v & ((1u << (8 * left)) - 1u);
    }
}
// table key: any 64-bit mix of the name's bytes and its length will do (equal names are told apart from equal keys by name_eq)
__device__ __forceinline__ unsigned long long name_hash64(const char *names, uint32_t n0, int l)
{
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)(unsigned)l;
    for (int off = 0; off < l; off += 16) {
        uint32_t w[4];
        name_words16(names, n0, l, off, w);
        h = (h ^ (((unsigned long long)w[1] << 32) | w[0])) * 1099511628211ull;
        h ^= h >> 29;
        h = (h ^ (((unsigned long long)w[3] << 32) | w[2])) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32;
    }
    return h ? h : 1;
}
// khash.h __ac_X31_hash_string + __ac_Wang_hash (keeper selection, A.3)
__device__ __forceinline__ uint32_t x31_wang(const char *names, uint32_t n0, int l)
{
    uint32_t h = 0;
    bool more = true;                              // (the string ends at its first NUL)
    for (int off = 0; off < l; off += 16) {
        uint32_t w[4];
        name_words16(names, n0, l, off, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
            if (off + k == 0) { h = c; more = c != 0; }
            else if (off + k < l && more) { if (c == 0) more = false; else h = (h << 5) - h + c; }
        }
    }
    uint32_t key = h;
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}

__device__ __forceinline__ bool name_eq(const StaReadsDev &R, int64_t a, int64_t b)
{
    const uint32_t a0 = R.name_off[a], a1 = R.name_off[a + 1], b0 = R.name_off[b], b1 = R.name_off[b + 1];
    if (a1 - a0 != b1 - b0) return false;
    const int l = (int)(a1 - a0) - 1;
    for (int off = 0; off < l; off += 16) {
        uint32_t x[4], y[4];
        name_words16(R.names, a0, l, off, x);
        name_words16(R.names, b0, l, off, y);
        if (((x[0] ^ y[0]) | (x[1] ^ y[1]) | (x[2] ^ y[2]) | (x[3] ^ y[3])) != 0) return false;
    }
    return true;
}

// which reads take part in name matching
#define SEL_MPLP 0   // every read that reaches bam_plp_push: the eligible ones (RI_OLAP_EL) put and find entries; the others still REMOVE the
                     // entry of their name -- dropped by the -d cap at the push, or when they leave the buffer (overlap_remove is by name)
#define SEL_DEPTH 1  // kept && PAIRED && !MUNMAP
__device__ __forceinline__ bool in_set(const StaReadsDev &R, int64_t i, int sel)
{
    uint32_t info = R.info[i];
    if (sel == SEL_MPLP) return (info & RI_PUSHED) && R.end[i] > R.pos[i];
    uint32_t flag = R.flag[i];
    return (info & RI_KEEP) && (flag & BAM_FPAIRED) && !(flag & BAM_FMUNMAP);
}

__global__ void __launch_bounds__(256) k_name_insert(StaReadsDev R, NameSlot *tab, size_t mask, int32_t *chain_next, int sel, const unsigned long long *gate)
{
    if (gate && *gate == 0) return;       // mpileup: no read of the window is eligible (counted by k_prep_reads)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.n || !in_set(R, i, sel)) return;
    uint32_t n0 = R.name_off[i];
    int l = (int)(R.name_off[i + 1] - n0) - 1;
    unsigned long long h = name_hash64(R.names, n0, l);
    size_t s = (size_t)h & mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&tab[s].key, 0ull, h);
        if (prev == 0ull || prev == h) break;
        s = (s + 1) & mask;
    }
    chain_next[i] = (int32_t)atomicExch(&tab[s].head, (unsigned int)(i + 1));
}

// ---- tweak_overlap_quality(a, b): a = read already in the hash, b = arriving mate (A.3.1) ----
// The reference walks two CIGAR cursors base by base and rewrites qualities as it goes.  Every query base is touched by at most one action of
// the walk (both cursors only move forward), so the walk is split into WHAT it does -- a sequence of actions (resolve a_qual[x] against
// b_qual[y] | lower a_qual[x] across a deletion of b | lower b_qual[y]) found from the CIGARs alone (pair_event: the reference's loop cut at its
// actions, the current CIGAR word and the operation in front of it in registers, no quality or base touched) -- and DOING it, which is
// independent work per base (apply_action).  One thread on its own was ~150 instructions and several dependent memory round trips per
// overlapping column, the lanes of a wave waiting for the longest overlap among them (k_name_groups: 0.50 ms of the 0.66 ms overlap pass of
// `mpileup -E -A` at bench size); resolve_pairs_wave() has every lane produce the next RUN of its pair (consecutive actions of one kind on
// consecutive bases), lays the runs of the wave end to end (prefix sum of their lengths) and lets the lanes take one base each, 64 at a time,
// until no lane has a run left.  When both CIGARs are [H][S] one M/=/X [S][H] the walk is closed form -- with d = bpos - apos (0 <= d < the M
// run of a) cigar_iref2iseq_set leaves a at query clipA + d and b at clipB, every step advances both by one and the loop ends with the shorter
// M run: ONE run of min(lenA - d, lenB) resolutions and no walk at all.  Pairs are disjoint (a record puts one entry or finds one), so the
// order among them does not matter.  Positions inside a read and window-relative columns are 32-bit here (R.pos / R.end are).
struct CCur { uint32_t cig, cig_max, cur; int prev_op, icig, iseq, iref; };
__device__ __forceinline__ void ccur_step(const uint32_t *C, CCur &w) { w.prev_op = (int)(w.cur & 0xf); w.cig++; w.cur = w.cig < w.cig_max ? C[w.cig] : 0xffffffffu; }
__device__ __forceinline__ void ccur_open(const uint32_t *C, CCur &w, uint32_t c, uint32_t c_end)
{
    w.cig = c; w.cig_max = c_end; w.prev_op = -1; w.cur = c < c_end ? C[c] : 0xffffffffu; w.icig = w.iseq = w.iref = 0;
}
// cigar_iref2iseq_set / _next
__device__ int ccur_set(const uint32_t *C, CCur &w, int pos)
{
    if (pos < 0) return -1;
    w.icig = 0; w.iseq = 0; w.iref = 0;
    while (w.cig < w.cig_max) {
        const int op = (int)(w.cur & 0xf), n = (int)(w.cur >> 4);
        if (op == CG_S) { ccur_step(C, w); w.iseq += n; w.icig = 0; continue; }
        if (op == CG_H || op == CG_P) { ccur_step(C, w); w.icig = 0; continue; }
        if (cg_is_mop(op)) {
            pos -= n;
            if (pos < 0) { w.icig = n + pos; w.iseq += w.icig; w.iref += w.icig; return 0; }
            ccur_step(C, w); w.iseq += n; w.icig = 0; w.iref += n;
            continue;
        }
        if (op == CG_I) { ccur_step(C, w); w.iseq += n; w.icig = 0; continue; }
        if (op == CG_D || op == CG_N) { pos -= n; if (pos < 0) pos = 0; ccur_step(C, w); w.icig = 0; w.iref += n; continue; }
        return -2;
    }
    w.iseq = -1;
    return -1;
}
__device__ int ccur_next(const uint32_t *C, CCur &w)
{
    while (w.cig < w.cig_max) {
        const int op = (int)(w.cur & 0xf), n = (int)(w.cur >> 4);
        if (cg_is_mop(op)) {
            if (w.icig >= n - 1) { w.icig = -1; ccur_step(C, w); continue; }
            w.iseq++; w.icig++; w.iref++;
            return 0;
        }
        if (op == CG_D || op == CG_N) { ccur_step(C, w); w.iref += n; w.icig = -1; continue; }
        if (op == CG_I || op == CG_S) { ccur_step(C, w); w.iseq += n; w.icig = -1; continue; }
        if (op == CG_H || op == CG_P) { ccur_step(C, w); w.icig = -1; continue; }
        return -2;
    }
    w.iseq = -1; w.iref = -1;
    return -1;
}

#define EV_PAIR 0      // resolve a_qual[x] against b_qual[y]
#define EV_LOW_A 1     // a_qual[x] lowered (or zeroed) across a deletion of b
#define EV_LOW_B 2
struct PairWalk { CCur wa, wb; int iref, apos, bpos, a_ret, b_ret, mode, alq, blq; };

// a's bases under b's deletion: the do-while steps a by one and lowers the next base while a stays inside its M run and below b's column
__device__ __forceinline__ int low_a_more(PairWalk &p)
{
    int k = (int)(p.wa.cur >> 4) - 1 - p.wa.icig;
    const int g = (p.wb.iref + p.bpos) - (p.wa.iref + p.apos) - 1;
    const int q = p.alq - 1 - p.wa.iseq;             // (never past the SEQ: memory safety on malformed records only)
    k = k < g ? k : g; k = k < q ? k : q;
    if (k <= 0) return 0;
    p.wa.icig += k; p.wa.iseq += k; p.wa.iref += k;
    return k;
}
// the next action of the walk; false when it is over (len consecutive actions of its kind on consecutive bases)
__device__ bool pair_event(const uint32_t *C, PairWalk &p, int &kind, int &x, int &y, int &len)
{
    len = 1;
    for (;;) {
        if (p.mode == 0) {
            while (p.a_ret >= 0 && p.wa.iref >= 0 && p.wa.iref < p.iref - p.apos) p.a_ret = ccur_next(C, p.wa);
            if (p.a_ret < 0) return false;
            if (p.iref < p.wa.iref + p.apos) p.iref = p.wa.iref + p.apos;
            while (p.b_ret >= 0 && p.wb.iref >= 0 && p.wb.iref < p.iref - p.bpos) p.b_ret = ccur_next(C, p.wb);
            if (p.b_ret < 0) return false;
            if (p.iref < p.wb.iref + p.bpos) p.iref = p.wb.iref + p.bpos;
            p.iref++;
            if (p.wa.iref + p.apos != p.wb.iref + p.bpos) {
                if (p.wa.iref + p.apos < p.wb.iref + p.bpos && p.wb.prev_op == CG_D) { if (p.wa.iseq >= p.alq) return false; p.mode = 1; kind = EV_LOW_A; x = p.wa.iseq; y = p.wb.iseq; len = 1 + low_a_more(p); return true; }
                if (p.wa.prev_op == CG_D) { if (p.wb.iseq >= p.blq) return false; p.mode = 2; kind = EV_LOW_B; x = p.wa.iseq; y = p.wb.iseq; return true; }
                continue;
            }
        } else if (p.mode == 1) {                 // inside the do-while over a's bases under b's deletion
            p.a_ret = ccur_next(C, p.wa);
            if (p.a_ret < 0) return false;
            if (p.wa.iref + p.apos < p.wb.iref + p.bpos) { if (p.wa.iseq >= p.alq) return false; kind = EV_LOW_A; x = p.wa.iseq; y = p.wb.iseq; len = 1 + low_a_more(p); return true; }
        } else {
            p.b_ret = ccur_next(C, p.wb);
            if (p.b_ret < 0) return false;
            if (p.wb.iref + p.bpos < p.wa.iref + p.apos) { if (p.wb.iseq >= p.blq) return false; kind = EV_LOW_B; x = p.wa.iseq; y = p.wb.iseq; return true; }
        }
        p.mode = 0;
        if (p.wa.iseq >= p.alq || p.wb.iseq >= p.blq) return false;      // (the reference tests `>` and reads one byte behind a record whose CIGAR outruns its SEQ)
        kind = EV_PAIR; x = p.wa.iseq; y = p.wb.iseq;
        if (p.wa.iref + p.apos == p.wb.iref + p.bpos) {
            // Both cursors stand on the same column inside M runs: the following iterations step both by one and resolve the next column until
            // either run ends (or a cursor passes its SEQ: the iteration after that makes the walk's bound check fail) -- taken in one go.
            int k = (int)(p.wa.cur >> 4) - 1 - p.wa.icig;
            const int kb = (int)(p.wb.cur >> 4) - 1 - p.wb.icig, qa = p.alq - 1 - p.wa.iseq, qb = p.blq - 1 - p.wb.iseq;
            k = k < kb ? k : kb; k = k < qa ? k : qa; k = k < qb ? k : qb;
            if (k > 0) {
                p.wa.icig += k; p.wa.iseq += k; p.wa.iref += k;
                p.wb.icig += k; p.wb.iseq += k; p.wb.iref += k;
                p.iref = p.wa.iref + p.apos + 1;
                len += k;
            }
        }
        return true;
    }
}
// both cursors onto the mate's first column (and tweak_overlap's look at a deletion of a in front of it: its placeholders print the quality of
// the next query base, which the resolution may rewrite -- placeholder_qual in dev_util.h); false: the pair has nothing to resolve
__device__ bool pair_open(const StaReadsDev &R, int64_t ia, int64_t ib, PairWalk &p)
{
    const uint32_t ca = R.cig_off[ia], ca_end = R.cig_off[ia + 1], cb = R.cig_off[ib], cb_end = R.cig_off[ib + 1];
    p.apos = R.pos[ia]; p.bpos = R.pos[ib]; p.alq = R.l_qseq[ia]; p.blq = R.l_qseq[ib]; p.mode = 0;
    if (R.fix_y && p.bpos > p.apos) {
        int x = p.apos, y = 0;
        for (uint32_t c = ca; c < ca_end; ++c) {
            const int op = (int)(R.cigar[c] & 0xf), l = (int)(R.cigar[c] >> 4);
            if (cg_is_refop(op)) {
                if (p.bpos - 1 < x + l) {
                    if ((op == CG_D || op == CG_N) && y < p.alq) { R.fix_y[ia] = y; R.fix_q[ia] = R.qual[((uint64_t)R.base_off8[ia] << 3) + (uint64_t)y]; R.fix_mate[ia] = (int32_t)ib; }
                    break;
                }
                if (cg_is_mop(op)) y += l;
                x += l;
            } else if (cg_is_qop(op)) y += l;
        }
    }
    if (p.alq <= 0 || p.blq <= 0) return false;      // a mate without SEQ: the reference reads qual[] of the empty record and then fails the run
    ccur_open(R.cigar, p.wa, ca, ca_end); ccur_open(R.cigar, p.wb, cb, cb_end);
    p.iref = p.bpos;
    p.a_ret = ccur_set(R.cigar, p.wa, p.iref - p.apos);
    if (p.a_ret < 0) return false;
    p.b_ret = ccur_set(R.cigar, p.wb, p.iref - p.bpos);
    return p.b_ret >= 0;
}
// one action on the bases ga / gb of the pools (km = kind | keeper-is-a << 2)
__device__ __forceinline__ void apply_action(const StaReadsDev &R, int km, uint64_t ga, uint64_t gb)
{
    const int kind = km & 3, am = km >> 2, bm = 1 - am;
    if (kind == EV_PAIR) {
        const int qa = R.qual[ga], qb = R.qual[gb];
        const int na = (R.seq[ga >> 1] >> ((~ga & 1) << 2)) & 0xf, nb = (R.seq[gb >> 1] >> ((~gb & 1) << 2)) & 0xf;
        uint8_t ra, rb;
        if (na == nb) {
            int q = qa + qb; if (q > 200) q = 200;
            ra = (uint8_t)(am * q); rb = (uint8_t)(bm * q);
        } else if (qa > qb) { ra = (uint8_t)(0.8 * qa); rb = 0; }
        else if (qa < qb) { rb = (uint8_t)(0.8 * qb); ra = 0; }
        else { ra = (uint8_t)(am * 0.8 * qa); rb = (uint8_t)(bm * 0.8 * qb); }
        R.qual[ga] = ra; R.qual[gb] = rb;
    } else if (kind == EV_LOW_A) R.qual[ga] = am ? (uint8_t)(R.qual[ga] * 0.8) : (uint8_t)0;
    else R.qual[gb] = bm ? (uint8_t)(R.qual[gb] * 0.8) : (uint8_t)0;
}
__device__ __forceinline__ int keeper_is_a(const StaReadsDev &R, int64_t ia)
{
    const uint32_t n0 = R.name_off[ia];
    return (x31_wang(R.names, n0, (int)(R.name_off[ia + 1] - n0) - 1) & 1) ? 1 : 0;
}

// one pair by one thread (the templates with more than two records: their pairs come out of the leader's replay one after the other)
__device__ void tweak_overlap(const StaReadsDev &R, int64_t ia, int64_t ib)
{
    PairWalk p;
    if (!pair_open(R, ia, ib, p)) return;
    const uint64_t aoff = (uint64_t)R.base_off8[ia] << 3, boff = (uint64_t)R.base_off8[ib] << 3;
    const int am = keeper_is_a(R, ia);
    int k, x, y, len;
    while (pair_event(R.cigar, p, k, x, y, len))
        for (int j = 0; j < len; ++j)
            apply_action(R, k | (am << 2), aoff + (uint64_t)(int64_t)(k == EV_LOW_B ? x : x + j), boff + (uint64_t)(int64_t)(k == EV_LOW_A ? y : y + j));
}

__device__ __forceinline__ bool plain_cigar(const uint32_t *c, int n, int lq, int &clip, int &mlen)
{
    int k = 0;
    clip = 0; mlen = 0;
    if (k < n && (c[k] & 0xf) == CG_H) ++k;
    if (k < n && (c[k] & 0xf) == CG_S) { clip = (int)(c[k] >> 4); ++k; }
    if (k >= n || !cg_is_mop((int)(c[k] & 0xf))) return false;
    mlen = (int)(c[k] >> 4); ++k;
    if (k < n && (c[k] & 0xf) == CG_S) ++k;
    if (k < n && (c[k] & 0xf) == CG_H) ++k;
    return k == n && mlen > 0 && clip + mlen <= lq;       // (a record without SEQ: the walk's own bound check decides)
}

__device__ void resolve_pairs_wave(const StaReadsDev &R, int64_t ia, int64_t ib)
{
    const int lane = (int)(threadIdx.x & 63);
    uint64_t aoff = 0, boff = 0, run_a = 0, run_b = 0;      // the reads' first bases in the pools; the run's first bases
    int run_len = 0, run_kind = EV_PAIR, amul = 0;
    bool live = false, pending = false;
    int pend_kind = 0, pend_x = 0, pend_y = 0, pend_len = 1;
    PairWalk p;
    p.mode = 0; p.a_ret = p.b_ret = -1; p.iref = p.apos = p.bpos = 0; p.alq = p.blq = 0;
    p.wa.cig = p.wa.cig_max = p.wb.cig = p.wb.cig_max = 0; p.wa.cur = p.wb.cur = 0xffffffffu;
    p.wa.prev_op = p.wb.prev_op = -1; p.wa.icig = p.wa.iseq = p.wa.iref = p.wb.icig = p.wb.iseq = p.wb.iref = 0;
    if (ia >= 0) {
        const uint32_t ca = R.cig_off[ia], ca_end = R.cig_off[ia + 1], cb = R.cig_off[ib], cb_end = R.cig_off[ib + 1];
        aoff = (uint64_t)R.base_off8[ia] << 3; boff = (uint64_t)R.base_off8[ib] << 3;
        int clip_a, len_a, clip_b, len_b;
        if (plain_cigar(R.cigar + ca, (int)(ca_end - ca), R.l_qseq[ia], clip_a, len_a) && plain_cigar(R.cigar + cb, (int)(cb_end - cb), R.l_qseq[ib], clip_b, len_b)) {
            const int d = R.pos[ib] - R.pos[ia];
            if (d >= 0 && d < len_a) {
                run_len = len_a - d < len_b ? len_a - d : len_b; run_kind = EV_PAIR;
                run_a = aoff + (uint64_t)clip_a + (uint64_t)d;
                run_b = boff + (uint64_t)clip_b;
            }
        } else live = pair_open(R, ia, ib, p);
        if (run_len > 0 || live) amul = keeper_is_a(R, ia);
    }
    for (;;) {
        if (live && run_len == 0) {
            int k = 0, x = 0, y = 0, len = 1;
            bool have = true;
            if (pending) { k = pend_kind; x = pend_x; y = pend_y; len = pend_len; pending = false; }
            else have = pair_event(R.cigar, p, k, x, y, len);
            if (!have) live = false;
            else {
                run_kind = k; run_a = aoff + (uint64_t)(int64_t)x; run_b = boff + (uint64_t)(int64_t)y; run_len = len;
                for (;;) {
                    int k2, x2, y2, len2;
                    if (!pair_event(R.cigar, p, k2, x2, y2, len2)) { live = false; break; }
                    if (k2 == run_kind && (k2 == EV_LOW_B || x2 == x + run_len) && (k2 == EV_LOW_A || y2 == y + run_len)) { run_len += len2; continue; }
                    pending = true; pend_kind = k2; pend_x = x2; pend_y = y2; pend_len = len2;
                    break;
                }
            }
        }
        if (__ballot(run_len > 0) == 0) break;
        int incl = run_len;
        for (int dd = 1; dd < 64; dd <<= 1) { const int v = __shfl_up(incl, dd); if (lane >= dd) incl += v; }
        const int total = __shfl(incl, 63), excl = incl - run_len;
        const uint32_t a_lo = (uint32_t)run_a, a_hi = (uint32_t)(run_a >> 32), b_lo = (uint32_t)run_b, b_hi = (uint32_t)(run_b >> 32);
        const int km = run_kind | (amul << 2);
        for (int base = 0; base < total; base += 64) {
            const int w = base + lane;
            int own = 0;                                       // the lanes whose running total is <= w come in front of w's run
            for (int step = 32; step; step >>= 1) { const int v = __shfl(incl, own + step - 1); if (v <= w) own += step; }
            const int t = w - __shfl(excl, own);
            const int kmo = __shfl(km, own);
            const uint64_t ga = (((uint64_t)__shfl(a_hi, own) << 32) | __shfl(a_lo, own)) + (uint64_t)((kmo & 3) == EV_LOW_B ? 0 : t);
            const uint64_t gb = (((uint64_t)__shfl(b_hi, own) << 32) | __shfl(b_lo, own)) + (uint64_t)((kmo & 3) == EV_LOW_A ? 0 : t);
            if (w < total) apply_action(R, kmo, ga, gb);
        }
        run_len = 0;
    }
}

// position of the previous read that reached bam_plp_push and was not dropped (max_pos at x's push)
__device__ __forceinline__ int prev_pushed_pos(const StaReadsDev &R, int64_t x)
{
    for (int64_t j = x - 1; j >= 0; --j) {
        uint32_t info = R.info[j];
        bool dropped = (info & RI_PUSHED) && !(info & RI_KEEP) && R.end[j] > R.pos[j];
        if ((info & RI_PUSHED) && !dropped) return R.pos[j];
    }
    return INT32_MIN;
}

// the replay of read i's name group by its leader; a group of exactly two records leaves its one possible pair in (pa, pb) for the wave
__device__ void name_group_replay(const StaReadsDev &R, int64_t i, int64_t origin, int32_t tid, const NameSlot *tab, size_t mask,
                                  const int32_t *chain_next, int sel, StaCounters *ctr, int64_t &pa, int64_t &pb)
{
    uint32_t n0 = R.name_off[i];
    int l = (int)(R.name_off[i + 1] - n0) - 1;
    unsigned long long h = name_hash64(R.names, n0, l);
    size_t s = (size_t)h & mask;
    while (tab[s].key != h) s = (s + 1) & mask;
    // am I the first (smallest index) member with exactly my name?
    int members = 0;
    int64_t other = -1;                   // a group of two: the leader's one companion (the replay below needs no second look at the chain)
    for (unsigned int j = tab[s].head; j; j = (unsigned int)chain_next[j - 1]) {
        int64_t m = (int64_t)j - 1;
        if (m == i) { members++; continue; }
        if (!name_eq(R, i, m)) continue;
        if (m < i) return;
        members++;
        other = m;
    }
    if (members < 2) return;
    if (members > 2) atomicAdd(&ctr->n_anom, 1ull);
    // replay in file order
    int64_t last = -1, holder = -1;
    long long holder_end = 0;
    for (int step = 0; step < members; ++step) {
        int64_t x = INT64_MAX;
        if (members == 2) x = step == 0 ? i : other;
        else for (unsigned int j = tab[s].head; j; j = (unsigned int)chain_next[j - 1]) {
            int64_t m = (int64_t)j - 1;
            if (m > last && m < x && (m == i || name_eq(R, i, m))) x = m;
        }
        last = x;
        if (sel == SEL_MPLP) {
            uint32_t info = R.info[x];
            // A member that leaves the buffer takes the entry of its NAME with it (bam_plp_next frees a node whose end the iterator has
            // passed and calls overlap_remove): the holder itself, or any other record of the template -- a supplementary alignment, a
            // primary that overlap_push turned away.  A member e is freed once a read beyond its end has been pushed; the entry it finds
            // is the holder's if the holder was put before that: max_pos at the holder's push <= end[e] < max_pos at x's push.
            if (holder >= 0 && members > 2) {
                const int before_x = prev_pushed_pos(R, x), before_h = prev_pushed_pos(R, holder);
                for (unsigned int j = tab[s].head; j && holder >= 0; j = (unsigned int)chain_next[j - 1]) {
                    const int64_t m = (int64_t)j - 1;
                    if (m >= x || !(R.info[m] & RI_KEEP) || !(m == i || name_eq(R, i, m))) continue;
                    if (before_h <= R.end[m] && R.end[m] < before_x) holder = -1;
                }
            }
            if (!(info & RI_KEEP)) { holder = -1; continue; }              // dropped by -d at its push: overlap_remove
            if (!(info & RI_OLAP_EL)) continue;                            // in the buffer, but overlap_push returned before the hash
            if (holder < 0) {
                long long mpos = R.mpos[x];
                if (mpos >= origin + R.pos[x] || ((R.flag[x] & BAM_FPAIRED) && mpos == -1)) holder = x;
            } else {
                if (members == 2) { pa = holder; pb = x; }
                else tweak_overlap(R, holder, x);
                holder = -1;
            }
        } else {
            // bam_endpos (R.end is the CIGAR's reach; an unmapped-flagged record counts as one column, see k_prep_reads_depth)
            long long endpos = origin + ((R.info[x] & RI_UNMAP_SPAN) ? R.pos[x] + 1 : R.end[x]);
            if (holder < 0) {
                long long mpos = R.mpos[x];
                if (mpos == -1 || (R.mtid[x] == tid && mpos <= endpos)) { holder = x; holder_end = endpos; }
            } else {
                long long c = holder_end - origin;
                R.clip[x] = (int32_t)(c > INT32_MAX ? INT32_MAX : (c < INT32_MIN + 1 ? INT32_MIN + 1 : c));
                holder = -1;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_name_groups(StaReadsDev R, int64_t origin, int32_t tid, const NameSlot *tab, size_t mask,
                                                    const int32_t *chain_next, int sel, StaCounters *ctr, const unsigned long long *gate)
{
    if (gate && *gate == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t pa = -1, pb = -1;
    if (i < R.n && in_set(R, i, sel)) name_group_replay(R, i, origin, tid, tab, mask, chain_next, sel, ctr, pa, pb);
    if (sel == SEL_MPLP) resolve_pairs_wave(R, pa, pb);
}

// mpileup with the caller's overlap hash (sta_reads.olap_mate, host_names.h): read i found the entry of read mate[i] at its push -- the
// pair tweak_overlap_quality(mate[i], i) resolves.  Pairs are disjoint (a record puts an entry or finds one, once), so one thread per
// finder is race free.
__global__ void __launch_bounds__(256) k_olap_pairs(StaReadsDev R, const unsigned long long *gate)
{
    if (gate && *gate == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t m = i < R.n ? (int64_t)R.mate[i] : -1;
    if (m >= R.n || m == i || m < 0 || !(R.info[i] & RI_KEEP) || !(R.info[m] & RI_KEEP)) m = -1;
    resolve_pairs_wave(R, m, i);
}

// mpileup, before the name matching: everything the mate-overlap pass needs is set up by ONE launch that looks at the window's
// number of eligible reads first (StaCounters.n_olap_el, counted by k_prep_reads).  None -- single-end data, the usual case of a
// long-read or amplicon run: the file's device descriptor is pointed back at the input quality pool and loses its fix-up arrays, and
// neither the pool copy (one byte in, one out per staged base) nor the table clear nor the matching happens.  Otherwise: working
// pool = input pool (unless k_qual_prep already built it: qout == NULL), name table cleared, fix_y = -1.
__global__ void __launch_bounds__(256) k_olap_setup(const unsigned long long *gate, StaReadsDev *dev_file, const uint8_t *__restrict__ qin, uint8_t *__restrict__ qout,
                                                    uint64_t nbytes, uint4 *__restrict__ table16, uint64_t table_n16, int32_t *__restrict__ fix_y, int64_t n)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    if (*gate == 0) {
        if (tid == 0) { if (qout) dev_file->qual = const_cast<uint8_t *>(qin); dev_file->fix_y = nullptr; dev_file->fix_mate = nullptr; dev_file->fix_q = nullptr; }
        return;
    }
    if (qout) {
        const uint64_t n16 = nbytes >> 4;
        for (uint64_t i = tid; i < n16; i += stride) reinterpret_cast<uint4 *>(qout)[i] = reinterpret_cast<const uint4 *>(qin)[i];
        if (tid < (nbytes & 15)) qout[(n16 << 4) + tid] = qin[(n16 << 4) + tid];
    }
    for (uint64_t i = tid; i < table_n16; i += stride) table16[i] = make_uint4(0, 0, 0, 0);
    for (uint64_t i = tid; i < (uint64_t)n; i += stride) fix_y[i] = -1;
}

void sta_launch_overlap_setup(hipStream_t s, const StaReadsDev &r, StaReadsDev *dev_file, bool copy_qual, void *table, size_t slots, const StaCounters *ctr)
{
    if (r.n == 0) return;
    uint64_t work = (r.n_bases_total >> 4) > slots ? (r.n_bases_total >> 4) : slots;
    uint64_t nb = (work + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(k_olap_setup, dim3((unsigned)nb), dim3(256), 0, s, &ctr->n_olap_el, dev_file, r.qual_in, copy_qual ? r.qual : (uint8_t *)nullptr,
                       (uint64_t)r.n_bases_total, (uint4 *)table, (uint64_t)(slots * sizeof(NameSlot) / 16), r.fix_y, r.n);
}

static void run_names(hipStream_t s, const StaReadsDev &r, int64_t origin, int32_t tid, void *table, size_t slots,
                      int32_t *chain_next, int sel, StaCounters *ctr)
{
    if (r.n == 0) return;
    // (mpileup: sta_launch_overlap_setup cleared the table, and the two kernels return at once when no read is eligible)
    const unsigned long long *gate = sel == SEL_MPLP ? &ctr->n_olap_el : nullptr;
    if (sel != SEL_MPLP) hipMemsetAsync(table, 0, slots * sizeof(NameSlot), s);
    unsigned nb = (unsigned)((r.n + 255) / 256);
    hipLaunchKernelGGL(k_name_insert, dim3(nb), dim3(256), 0, s, r, (NameSlot *)table, slots - 1, chain_next, sel, gate);
    hipLaunchKernelGGL(k_name_groups, dim3(nb), dim3(256), 0, s, r, origin, tid, (const NameSlot *)table, slots - 1,
                       (const int32_t *)chain_next, sel, ctr, gate);
}

size_t sta_overlap_table_bytes(size_t slots) { return slots * sizeof(NameSlot); }

void sta_launch_overlap(hipStream_t s, const StaReadsDev &r, int64_t origin, int32_t tid, void *table, size_t slots,
                        int32_t *chain_next, StaCounters *ctr)
{
    if (r.mate) {
        if (r.n) hipLaunchKernelGGL(k_olap_pairs, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, &ctr->n_olap_el);
        return;
    }
    run_names(s, r, origin, tid, table, slots, chain_next, SEL_MPLP, ctr);
}

void sta_launch_depth_pair(hipStream_t s, const StaReadsDev &r, int64_t origin, int32_t tid, void *table, size_t slots,
                           int32_t *chain_next, StaCounters *ctr)
{
    run_names(s, r, origin, tid, table, slots, chain_next, SEL_DEPTH, ctr);
}
