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

__device__ __forceinline__ unsigned long long name_hash64(const char *s, int l)
{
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < l; ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    return h ? h : 1;
}
// khash.h __ac_X31_hash_string + __ac_Wang_hash (keeper selection, A.3)
__device__ __forceinline__ uint32_t x31_wang(const char *s, int l)
{
    uint32_t h = l > 0 ? (uint32_t)(unsigned char)s[0] : 0;
    if (h) for (int i = 1; i < l; ++i) h = (h << 5) - h + (uint32_t)(unsigned char)s[i];
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
    uint32_t a0 = R.name_off[a], a1 = R.name_off[a + 1], b0 = R.name_off[b], b1 = R.name_off[b + 1];
    if (a1 - a0 != b1 - b0) return false;
    for (uint32_t i = 0; i < a1 - a0; ++i) if (R.names[a0 + i] != R.names[b0 + i]) return false;
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
    unsigned long long h = name_hash64(R.names + n0, l);
    size_t s = (size_t)h & mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&tab[s].key, 0ull, h);
        if (prev == 0ull || prev == h) break;
        s = (s + 1) & mask;
    }
    chain_next[i] = (int32_t)atomicExch(&tab[s].head, (unsigned int)(i + 1));
}

// ---- CIGAR cursor of cigar_iref2iseq_set/next (A.3.1) ----
struct CWalk { const uint32_t *cig, *cig0, *cig_max; long long icig, iseq, iref; };

__device__ int iref2iseq_set(CWalk &w, long long pos)
{
    if (pos < 0) return -1;
    w.icig = 0; w.iseq = 0; w.iref = 0;
    while (w.cig < w.cig_max) {
        int op = *w.cig & 0xf; long long n = *w.cig >> 4;
        if (op == CG_S) { w.cig++; w.iseq += n; w.icig = 0; continue; }
        if (op == CG_H || op == CG_P) { w.cig++; w.icig = 0; continue; }
        if (cg_is_mop(op)) {
            pos -= n;
            if (pos < 0) { w.icig = n + pos; w.iseq += w.icig; w.iref += w.icig; return 0; }
            w.cig++; w.iseq += n; w.icig = 0; w.iref += n;
            continue;
        }
        if (op == CG_I) { w.cig++; w.iseq += n; w.icig = 0; continue; }
        if (op == CG_D || op == CG_N) { pos -= n; if (pos < 0) pos = 0; w.cig++; w.icig = 0; w.iref += n; continue; }
        return -2;
    }
    w.iseq = -1;
    return -1;
}
__device__ int iref2iseq_next(CWalk &w)
{
    while (w.cig < w.cig_max) {
        int op = *w.cig & 0xf; long long n = *w.cig >> 4;
        if (cg_is_mop(op)) {
            if (w.icig >= n - 1) { w.icig = -1; w.cig++; continue; }
            w.iseq++; w.icig++; w.iref++;
            return 0;
        }
        if (op == CG_D || op == CG_N) { w.cig++; w.iref += n; w.icig = -1; continue; }
        if (op == CG_I || op == CG_S) { w.cig++; w.iseq += n; w.icig = -1; continue; }
        if (op == CG_H || op == CG_P) { w.cig++; w.icig = -1; continue; }
        return -2;
    }
    w.iseq = -1; w.iref = -1;
    return -1;
}

// tweak_overlap_quality(a, b): a = read already in the hash, b = arriving mate
__device__ void tweak_overlap(const StaReadsDev &R, int64_t ia, int64_t ib)
{
    long long apos = R.pos[ia], bpos = R.pos[ib];
    int alq = R.l_qseq[ia], blq = R.l_qseq[ib];
    uint64_t aoff = (uint64_t)R.base_off8[ia] << 3, boff = (uint64_t)R.base_off8[ib] << 3;
    uint8_t *a_qual = R.qual + aoff, *b_qual = R.qual + boff;
    CWalk wa, wb;
    wa.cig = wa.cig0 = R.cigar + R.cig_off[ia]; wa.cig_max = R.cigar + R.cig_off[ia + 1];
    wb.cig = wb.cig0 = R.cigar + R.cig_off[ib]; wb.cig_max = R.cigar + R.cig_off[ib + 1];
    long long iref = bpos;
    if (R.fix_y && bpos > apos) {
        // does a deletion / ref-skip run of a cover the column just before the mate starts?  Its placeholders look at the
        // quality of the next query base, which the resolution below may rewrite (placeholder_qual in dev_util.h)
        long long x = apos; int y = 0;
        for (const uint32_t *c = wa.cig; c < wa.cig_max; ++c) {
            int op = *c & 0xf; long long l = *c >> 4;
            if (cg_is_refop(op)) {
                if (bpos - 1 < x + l) {
                    if ((op == CG_D || op == CG_N) && y < alq) { R.fix_y[ia] = y; R.fix_q[ia] = a_qual[y]; R.fix_mate[ia] = (int32_t)ib; }
                    break;
                }
                if (cg_is_mop(op)) y += (int)l;
                x += l;
            } else if (cg_is_qop(op)) y += (int)l;
        }
    }
    int a_ret = iref2iseq_set(wa, iref - apos);
    if (a_ret < 0) return;
    int b_ret = iref2iseq_set(wb, iref - bpos);
    if (b_ret < 0) return;
    uint32_t n0 = R.name_off[ia];
    int nl = (int)(R.name_off[ia + 1] - n0) - 1;
    int amul = (x31_wang(R.names + n0, nl) & 1) ? 1 : 0, bmul = 1 - amul;

    for (;;) {
        while (a_ret >= 0 && wa.iref >= 0 && wa.iref < iref - apos) a_ret = iref2iseq_next(wa);
        if (a_ret < 0) break;
        if (iref < wa.iref + apos) iref = wa.iref + apos;
        while (b_ret >= 0 && wb.iref >= 0 && wb.iref < iref - bpos) b_ret = iref2iseq_next(wb);
        if (b_ret < 0) break;
        if (iref < wb.iref + bpos) iref = wb.iref + bpos;
        iref++;
        if (wa.iref + apos != wb.iref + bpos) {
            if (wa.iref + apos < wb.iref + bpos && wb.cig > wb.cig0 && (*(wb.cig - 1) & 0xf) == CG_D) {
                do {
                    a_qual[wa.iseq] = amul ? (uint8_t)(a_qual[wa.iseq] * 0.8) : 0;
                    a_ret = iref2iseq_next(wa);
                    if (a_ret < 0) return;
                } while (wa.iref + apos < wb.iref + bpos);
            } else if (wa.cig > wa.cig0 && (*(wa.cig - 1) & 0xf) == CG_D) {
                do {
                    b_qual[wb.iseq] = bmul ? (uint8_t)(b_qual[wb.iseq] * 0.8) : 0;
                    b_ret = iref2iseq_next(wb);
                    if (b_ret < 0) return;
                } while (wb.iref + bpos < wa.iref + apos);
            } else continue;
        }
        if (wa.iseq > alq || wb.iseq > blq) return;
        int qa = a_qual[wa.iseq], qb = b_qual[wb.iseq];
        if (seq_nib(R.seq, aoff >> 1, (int)wa.iseq) == seq_nib(R.seq, boff >> 1, (int)wb.iseq)) {
            int q = qa + qb; if (q > 200) q = 200;
            a_qual[wa.iseq] = (uint8_t)(amul * q);
            b_qual[wb.iseq] = (uint8_t)(bmul * q);
        } else if (qa > qb) {
            a_qual[wa.iseq] = (uint8_t)(0.8 * qa); b_qual[wb.iseq] = 0;
        } else if (qa < qb) {
            b_qual[wb.iseq] = (uint8_t)(0.8 * qb); a_qual[wa.iseq] = 0;
        } else {
            a_qual[wa.iseq] = (uint8_t)(amul * 0.8 * qa);
            b_qual[wb.iseq] = (uint8_t)(bmul * 0.8 * qb);
        }
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

__global__ void __launch_bounds__(256) k_name_groups(StaReadsDev R, int64_t origin, int32_t tid, const NameSlot *tab, size_t mask,
                                                    const int32_t *chain_next, int sel, StaCounters *ctr, const unsigned long long *gate)
{
    if (gate && *gate == 0) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.n || !in_set(R, i, sel)) return;
    uint32_t n0 = R.name_off[i];
    int l = (int)(R.name_off[i + 1] - n0) - 1;
    unsigned long long h = name_hash64(R.names + n0, l);
    size_t s = (size_t)h & mask;
    while (tab[s].key != h) s = (s + 1) & mask;
    // am I the first (smallest index) member with exactly my name?
    int members = 0;
    for (unsigned int j = tab[s].head; j; j = (unsigned int)chain_next[j - 1]) {
        int64_t m = (int64_t)j - 1;
        if (m == i) { members++; continue; }
        if (!name_eq(R, i, m)) continue;
        if (m < i) return;
        members++;
    }
    if (members < 2) return;
    if (members > 2) atomicAdd(&ctr->n_anom, 1ull);
    // replay in file order
    int64_t last = -1, holder = -1;
    long long holder_end = 0;
    for (int step = 0; step < members; ++step) {
        int64_t x = INT64_MAX;
        for (unsigned int j = tab[s].head; j; j = (unsigned int)chain_next[j - 1]) {
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
                tweak_overlap(R, holder, x);
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

// mpileup with the caller's overlap hash (sta_reads.olap_mate, host_names.h): read i found the entry of read mate[i] at its push -- the
// pair tweak_overlap_quality(mate[i], i) resolves.  Pairs are disjoint (a record puts an entry or finds one, once), so one thread per
// finder is race free.
__global__ void __launch_bounds__(256) k_olap_pairs(StaReadsDev R, const unsigned long long *gate)
{
    if (gate && *gate == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.n) return;
    const int64_t m = R.mate[i];
    if (m < 0 || m >= R.n || m == i) return;
    if (!(R.info[i] & RI_KEEP) || !(R.info[m] & RI_KEEP)) return;
    tweak_overlap(R, m, i);
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
