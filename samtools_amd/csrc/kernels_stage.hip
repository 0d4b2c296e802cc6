// kernels_stage.hip -- staging on the device: the pools of a window's reads are cut out of raw BAM alignment records (gfx950).
//
// What it replaces on the host: the per-record feed of the reference's drivers (sam_read1 in mplp_func, bam_plcmd.c:409;
// sam_itr_next / sam_read1 in fastdepth_core, bam2depth.c:541-543) as far as the variable-length parts of a record go -- and, in
// this engine, the serial "copy every read's CIGAR / bases / qualities / name into the staging pools" phase of the window
// producer (host_stage.cpp add_ranges), which bounded every file -> text run.  The host still walks the records for what the
// window logic needs (position, span, flags: 42 bytes per read); the 260 bytes per read of variable-length data never pass
// through a host copy: the inflated BAM bytes are uploaded as they are and this kernel addresses them record by record.
//
// Record layout (SAM spec 4.2, after the 4-byte block_size): refID i32 | pos i32 | l_read_name u8 | mapq u8 | bin u16 |
// n_cigar_op u16 | flag u16 | l_seq i32 | next_refID i32 | next_pos i32 | tlen i32 | read_name | cigar u32[] | seq (4-bit, high
// nibble first) | qual | aux.  Records are not aligned: every source access is by bytes.
//
// One wave per read: the lanes walk the destination bytes of the read's four slices (coalesced stores, byte gathers from the
// record).  Bases are padded to a multiple of 8 with zero bytes exactly as StagedFile::add does (qual offset = 8 * base_off8,
// seq offset = 4 * base_off8), so the device-built pools are byte-identical to the host-built ones (k_stage_compare checks
// that in the tests).
#include "dev_util.h"

__global__ void __launch_bounds__(256) k_bam_pools(const uint8_t *raw, const uint32_t *rec_off, uint64_t raw_bytes, int64_t raw_first, int64_t n_raw,
                                                   const int32_t *l_qseq, const uint32_t *cig_off, const uint32_t *base_off8, const uint32_t *name_off,
                                                   uint32_t *cigar, uint8_t *seq, uint8_t *qual, char *names, unsigned long long *bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_raw) return;
    const int64_t i = raw_first + t;
    // (rec_off comes through the public sta_reads fields: a record must lie inside the uploaded bytes, head and all four slices)
    const uint64_t ro = rec_off[t];
    if (ro + 32 > raw_bytes) { if (lane == 0) atomicAdd(bad, 1ull); return; }
    const uint8_t *rec = raw + ro;
    const int l_name = rec[8];
    const int n_cig = rec[12] | (rec[13] << 8);
    const int l_seq = (int)((uint32_t)rec[16] | ((uint32_t)rec[17] << 8) | ((uint32_t)rec[18] << 16) | ((uint32_t)rec[19] << 24));
    const uint32_t c0 = cig_off[i], c1 = cig_off[i + 1], n0 = name_off[i], n1 = name_off[i + 1];
    // the host's offsets were computed from ITS parse of the same record: a disagreement means the two views of the input differ
    if (l_seq != l_qseq[i] || (int)(c1 - c0) != n_cig || (int)(n1 - n0) != l_name || l_seq < 0
        || ro + 32 + (uint64_t)l_name + 4ull * (uint64_t)n_cig + (uint64_t)((l_seq + 1) >> 1) + (uint64_t)l_seq > raw_bytes) { if (lane == 0) atomicAdd(bad, 1ull); return; }
    const uint8_t *s_name = rec + 32, *s_cig = s_name + l_name, *s_seq = s_cig + 4 * n_cig, *s_qual = s_seq + ((l_seq + 1) >> 1);
    for (int k = lane; k < l_name; k += 64) names[n0 + k] = (char)s_name[k];
    for (int k = lane; k < n_cig; k += 64) {
        const uint8_t *p = s_cig + 4 * k;
        cigar[c0 + k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    const uint64_t b0 = (uint64_t)base_off8[i] << 3;
    const int padded = (l_seq + 7) & ~7, seq_bytes = (l_seq + 1) >> 1;
    for (int k = lane; k < padded; k += 64) qual[b0 + k] = k < l_seq ? s_qual[k] : (uint8_t)0;
    for (int k = lane; k < (padded >> 1); k += 64) seq[(b0 >> 1) + k] = k < seq_bytes ? s_seq[k] : (uint8_t)0;
}

// tests (sta_reads.raw_verify): the device-built pool slices against the host-built ones, byte for byte
__global__ void __launch_bounds__(256) k_stage_compare(const uint8_t *a, const uint8_t *b, uint64_t n, unsigned long long *bad)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long d = 0;
    for (; i < n; i += st) d += a[i] != b[i];
    if (d) atomicAdd(bad, d);
}

void sta_launch_bam_pools(hipStream_t s, const uint8_t *raw, const uint32_t *rec_off, uint64_t raw_bytes, int64_t raw_first, int64_t n_raw, const StaReadsDev &d,
                          uint32_t *cigar, uint8_t *seq, uint8_t *qual, char *names, unsigned long long *bad)
{
    if (n_raw <= 0) return;
    hipLaunchKernelGGL(k_bam_pools, dim3((unsigned)((n_raw + 3) / 4)), dim3(256), 0, s, raw, rec_off, raw_bytes, raw_first, n_raw, d.l_qseq, d.cig_off, d.base_off8, d.name_off,
                       cigar, seq, qual, names, bad);
}

void sta_launch_stage_compare(hipStream_t s, const void *a, const void *b, uint64_t n, unsigned long long *bad)
{
    if (!n) return;
    uint64_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_stage_compare, dim3((unsigned)nb), dim3(256), 0, s, (const uint8_t *)a, (const uint8_t *)b, n, bad);
}
