// kernels_inflate.hip -- BGZF blocks inflated on the device, one block per lane (inflate_core.h has the
// decoder and says why it looks the way it does).  A wave = 64 blocks of the file; the two Huffman codes of a lane (352
// 16-bit entries) live in LDS with the lane index fastest, the 320 code lengths of a dynamic header in the lane's private
// memory, the CRC-32 table once per workgroup in LDS.  Bound: serial bit-twiddling per lane (DEFLATE is inherently serial
// within a block) -- the parallelism is the number of blocks, so what matters is blocks in flight: 45 KB of LDS per wave,
// three waves per CU, 49 152 blocks = 3 GiB of inflated data per pass over the chip.
#include "sta_dev.h"
#include "inflate_core.h"
#include "../../include/samtools_amd.h"

#define INF_WAVE 64

__global__ void __launch_bounds__(INF_WAVE) k_bgzf_inflate(const uint8_t *__restrict__ comp, uint64_t comp_bytes, const sta_bgzf_block *__restrict__ blocks,
                                                           uint64_t n_blocks, uint8_t *out, uint64_t out_cap, uint32_t *status, unsigned long long *bad /* [0] count, [1] first index */)
{
    __shared__ uint32_t s_crc[256];
    __shared__ uint16_t s_tab[352 * INF_WAVE];
    for (uint32_t i = threadIdx.x; i < 256; i += INF_WAVE) sta_inflate::crc_table_entry(i, &s_crc[i]);
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * INF_WAVE + threadIdx.x;
    if (i >= n_blocks) return;
    const sta_bgzf_block b = blocks[i];
    int err = sta_inflate::OK;
    uint32_t got = 0, crc = 0;
    if (b.in_off > comp_bytes || b.in_len > comp_bytes - b.in_off || b.out_off > out_cap || b.out_len > out_cap - b.out_off) err = sta_inflate::ERR_SIZE;
    else {
        uint8_t lens[320];
        sta_inflate::Work w;
        uint16_t *base = s_tab + threadIdx.x;
        w.lit.count = base; w.lit.symbol = base + 16 * INF_WAVE; w.lit.stride = INF_WAVE;
        w.dist.count = base + 304 * INF_WAVE; w.dist.symbol = base + 320 * INF_WAVE; w.dist.stride = INF_WAVE;
        w.lens = lens; w.lens_stride = 1;
        err = sta_inflate::inflate_stream(comp + b.in_off, b.in_len, out + b.out_off, b.out_len, s_crc, w, &got, &crc);
        if (!err && got != b.out_len) err = sta_inflate::ERR_SIZE;
        if (!err && crc != b.crc32) err = sta_inflate::ERR_CRC;
    }
    if (status) status[i] = (uint32_t)err;
    if (err) { atomicAdd(&bad[0], 1ull); atomicMin(&bad[1], (unsigned long long)i); }
}

void sta_launch_bgzf_inflate(hipStream_t s, const uint8_t *comp, uint64_t comp_bytes, const sta_bgzf_block *blocks, uint64_t n_blocks,
                             uint8_t *out, uint64_t out_cap, uint32_t *status, unsigned long long *bad)
{
    if (!n_blocks) return;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)((n_blocks + INF_WAVE - 1) / INF_WAVE)), dim3(INF_WAVE), 0, s, comp, comp_bytes, blocks, n_blocks, out, out_cap, status, bad);
}
