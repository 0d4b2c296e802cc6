// kernels_md.hip -- calmd's per-record arithmetic: MD / NM recomputation and the BAQ tag.
//
// Replaces bam_fillmd1_core (bam_md.c:64-224) and the tag-writing tail of HTSlib's sam_prob_realn (call site
// bam_md.c:474-479); the BAQ itself is the engine's own kernels (kernels_baq.hip), run in plain or extended mode.
// Records are independent, so this is one thread per record over the staged SoA (same layout as the pileup path):
//   k_calmd_tag : for records whose BAQ was computed, tag[i] = 64 + (quality as read - quality after BAQ) -- the string
//                 realn.c stores as BQ:Z (qualities restored, no -A) or ZQ:Z (qualities kept, -A)
//   k_md_len    : NM and the length of the MD string (decimal run lengths + mismatch / deletion characters)
//   k_md_emit   : writes the MD string at its scanned offset, then applies -e (matches -> '='), -n (records at or over the
//                 edit-distance bound: matches -> N with quality 0) and -q (quality binning) to the working copies
// HBM-bound byte work: ~1.5 B read per aligned base (4-bit base, reference character), a few bytes of text out per record.
#include "dev_util.h"

struct MdPar { int32_t use_equal, bin_qual, max_nm, apply; };

// one wave per 64 consecutive records, the lanes stride over the bases of one record at a time (coalesced byte streams)
__global__ void __launch_bounds__(256) k_calmd_tag(StaReadsDev R, MdPar P, uint8_t *tag_pool, uint8_t *state, const uint8_t *bq_pool)
{
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;
    for (int j = 0; j < 64; ++j) {
        const int64_t r = r0 + j;
        if (r >= R.n) break;
        const uint32_t info = R.info[r], aux = R.aux[r];
        const uint64_t boff = (uint64_t)R.base_off8[r] << 3;
        const int lq = R.l_qseq[r];
        uint8_t st = 0;
        if (info & RI_BAQ) {
            st |= 2;
            for (int i = lane; i < lq; i += 64) {
                const uint8_t q0 = R.qual_in[boff + i], q1 = R.qual[boff + i];
                tag_pool[boff + i] = (uint8_t)(64 + (q0 - q1));
                if (!P.apply) R.qual[boff + i] = q0;
            }
        } else if (P.apply && (aux & STA_AUX_HAS_BQ) && R.bq && !(R.flag[r] & BAM_FUNMAP) && lq > 0 && R.qual_in[boff] != 0xff) {
            st |= 4;                   // an existing BQ:Z was applied by k_qual_prep (realn.c renames it ZQ:Z)
        } else if (!P.apply && (aux & STA_AUX_ZQ_RESTORE) && bq_pool && !(R.flag[r] & BAM_FUNMAP) && lq > 0 && R.qual_in[boff] != 0xff) {
            st |= 8;                   // ZQ:Z back to BQ:Z: the qualities get the stored difference back (realn.c: qual[i] += zq[i] - 64)
            for (int i = lane; i < lq; i += 64) R.qual[boff + i] = (uint8_t)(R.qual_in[boff + i] + ((int)bq_pool[boff + i] - 64));
        }
        if (lane == 0) state[r] = st;
    }
}

// shared walk of bam_md.c:89-124; EMIT = write the string, otherwise only measure
template <bool EMIT>
__device__ __forceinline__ void md_walk(const StaReadsDev &R, const StaWinDev &W, int64_t r, const uint8_t *seq, int &nm_out, uint32_t &len_out, char *dst)
{
    const uint32_t c0 = R.cig_off[r], c1 = R.cig_off[r + 1];
    const int lq = R.l_qseq[r];
    const uint64_t boff = (uint64_t)R.base_off8[r] << 3;
    const uint8_t *sq = seq + (boff >> 1);
    int64_t rpos = W.origin + R.pos[r];
    int qpos = 0, matched = 0, nm = 0;
    uint32_t len = 0;
    auto put_num = [&](int v) {
        char tmp[12]; int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        if (EMIT) for (int k = 0; k < n; ++k) dst[len + (uint32_t)k] = tmp[n - 1 - k];
        len += (uint32_t)n;
    };
    auto put_chr = [&](char c) { if (EMIT) dst[len] = c; ++len; };
    auto up = [](unsigned char c) -> char { return (char)((c >= 'a' && c <= 'z') ? c - 32 : c); };
    bool stop = false;
    for (uint32_t k = c0; k < c1 && !stop; ++k) {
        const int op = (int)(R.cigar[k] & 0xf), oplen = (int)(R.cigar[k] >> 4);
        if (cg_is_mop(op)) {
            int j;
            for (j = 0; j < oplen; ++j) {
                const int z = qpos + j;
                if (rpos + j >= W.ref_len || z >= lq) break;
                const int q1 = (sq[z >> 1] >> ((~z & 1) << 2)) & 0xf;
                const int q2 = nt16_from_char((unsigned char)W.ref[rpos + j]);
                if ((q1 == q2 && q1 != 15 && q2 != 15) || q1 == 0) ++matched;
                else { put_num(matched); put_chr(up((unsigned char)W.ref[rpos + j])); matched = 0; ++nm; }
            }
            if (j < oplen) { stop = true; break; }
            rpos += oplen; qpos += oplen;
        } else if (op == CG_D) {
            put_num(matched); put_chr('^');
            int j;
            for (j = 0; j < oplen; ++j) {
                if (rpos + j >= W.ref_len) break;
                put_chr(up((unsigned char)W.ref[rpos + j]));
            }
            matched = 0; rpos += j; nm += j;
            if (j < oplen) { stop = true; break; }
        } else if (op == CG_I || op == CG_S) {
            qpos += oplen;
            if (op == CG_I) nm += oplen;
        } else if (op == CG_N) rpos += oplen;
    }
    put_num(matched);
    nm_out = nm; len_out = len;
}

__global__ void __launch_bounds__(256) k_md_len(StaReadsDev R, StaWinDev W, int32_t *nm, uint32_t *md_len, uint8_t *state)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n) return;
    int n = -1; uint32_t l = 0;
    if (W.ref && R.l_qseq[r] > 0) {
        md_walk<false>(R, W, r, R.seq, n, l, nullptr);
        if (!(R.flag[r] & BAM_FUNMAP)) state[r] |= 1;      // NM / MD are written (UPDATE_NM|UPDATE_MD, mapped record)
        else l = 0;
    }
    nm[r] = n; md_len[r] = l;
}

__global__ void __launch_bounds__(256) k_md_emit(StaReadsDev R, StaWinDev W, MdPar P, const int32_t *nm, const uint64_t *md_off, char *md_text,
                                                uint8_t *seq_work)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R.n) return;
    const int lq = R.l_qseq[r];
    if (!W.ref || lq <= 0) return;
    if (!(R.flag[r] & BAM_FUNMAP)) { int n; uint32_t l; md_walk<true>(R, W, r, R.seq, n, l, md_text + md_off[r]); }
    const uint64_t boff = (uint64_t)R.base_off8[r] << 3;
    if (P.use_equal || (P.max_nm > 0 && nm[r] >= P.max_nm)) {
        // second walk (bam_md.c:131-152 and the USE_EQUAL branch of the first): rewrite the matching bases
        const bool to_n = P.max_nm > 0 && nm[r] >= P.max_nm;
        const uint8_t *sq = R.seq + (boff >> 1);
        uint8_t *sw = seq_work + (boff >> 1);
        int64_t rpos = W.origin + R.pos[r];
        int qpos = 0;
        for (uint32_t k = R.cig_off[r]; k < R.cig_off[r + 1]; ++k) {
            const int op = (int)(R.cigar[k] & 0xf), oplen = (int)(R.cigar[k] >> 4);
            if (cg_is_mop(op)) {
                int j;
                for (j = 0; j < oplen; ++j) {
                    const int z = qpos + j;
                    if (rpos + j >= W.ref_len || z >= lq) break;
                    const int q1 = (sq[z >> 1] >> ((~z & 1) << 2)) & 0xf;
                    const int q2 = nt16_from_char((unsigned char)W.ref[rpos + j]);
                    if ((q1 == q2 && q1 != 15 && q2 != 15) || q1 == 0) {
                        // -e first clears the nibble ('='), -n then sets it to 15 (N) and zeroes the quality
                        uint8_t b = sw[z >> 1];
                        if (P.use_equal) b &= (z & 1) ? 0xf0 : 0x0f;
                        if (to_n) { b |= (z & 1) ? 0x0f : 0xf0; R.qual[boff + (uint64_t)z] = 0; }
                        sw[z >> 1] = b;
                    }
                }
                if (j < oplen) break;
                rpos += oplen; qpos += oplen;
            } else if (op == CG_D) {
                // the first walk stops inside a deletion that runs off the reference; the second one does not look (it adds oplen)
                rpos += oplen;
            } else if (op == CG_N) rpos += oplen;
            else if (op == CG_I || op == CG_S) qpos += oplen;
        }
    }
    if (P.bin_qual)
        for (int i = 0; i < lq; ++i) { uint8_t q = R.qual[boff + i]; if (q >= 3) R.qual[boff + i] = (uint8_t)(q / 10 * 10 + 7); }
}

void sta_launch_calmd_tag(hipStream_t s, const StaReadsDev &r, int apply, uint8_t *tag_pool, uint8_t *state, const uint8_t *bq_pool)
{
    if (!r.n) return;
    MdPar p{ 0, 0, 0, apply };
    hipLaunchKernelGGL(k_calmd_tag, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, p, tag_pool, state, bq_pool);
}

void sta_launch_md_len(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int32_t *nm, uint32_t *md_len, uint8_t *state)
{
    if (!r.n) return;
    hipLaunchKernelGGL(k_md_len, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, w, nm, md_len, state);
}

void sta_launch_md_emit(hipStream_t s, const StaReadsDev &r, const StaWinDev &w, int use_equal, int bin_qual, int max_nm,
                        const int32_t *nm, const uint64_t *md_off, char *md_text, uint8_t *seq_work)
{
    if (!r.n) return;
    MdPar p{ use_equal, bin_qual, max_nm, 0 };
    hipLaunchKernelGGL(k_md_emit, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, s, r, w, p, nm, md_off, md_text, seq_work);
}
