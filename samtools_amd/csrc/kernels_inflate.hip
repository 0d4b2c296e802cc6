// kernels_inflate.hip -- BGZF blocks inflated on the device, one wave per block (gfx950).
//
// What it replaces on the host: bgzf.c inflate_block() under sam_read1 (bam_plcmd.c:409, bam2depth.c:541-543) -- and, in this engine,
// the decode threads' inflate (host_bgzf.cpp bgzf_inflate_block), which bounded every file -> text run on a host with few cores: the
// compressed file is uploaded as it is, and its tens of thousands of independent 64 KiB deflate streams are decoded side by side.
// The decoder itself (bit reader over a lane-held input window, Huffman tables and a 16 KiB ring of the output in LDS) is
// bgzf_inflate_dev.h; this file is the launch: block b of the table -> out[out_off, out_off + isize), status[b] = 0 or why not.
#include <hip/hip_runtime.h>
#include "sta_dev.h"
#include "bgzf_inflate_dev.h"

__global__ void __launch_bounds__(64) k_bgzf_inflate(const uint8_t *__restrict__ comp, const StaBgzfBlock *__restrict__ blocks, int n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status, const bgzi::Consts C)
{
    __shared__ bgzi::Lds L;
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    if (b >= n_blocks) return;
    const StaBgzfBlock blk = blocks[b];
    uint8_t *dst = out + blk.out_off;
    const uint32_t align = (uint32_t)((uintptr_t)dst & 15);
    int st = bgzi::ST_SIZE;
    if (blk.isize <= 65536u) st = bgzi::inflate_block(L, C, comp + blk.comp_off, (int64_t)blk.clen, blk.isize, align, dst);
    if (lane == 0) status[b] = (uint32_t)st;
}

void sta_launch_bgzf_inflate(hipStream_t s, const uint8_t *comp, const StaBgzfBlock *blocks, int n_blocks, uint8_t *out, uint32_t *status)
{
    if (n_blocks <= 0) return;
    static const bgzi::Consts C = [] { bgzi::Consts c; bgzi::fill_consts(c); return c; }();
    hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)n_blocks), dim3(64), 0, s, comp, blocks, n_blocks, out, status, C);
}
