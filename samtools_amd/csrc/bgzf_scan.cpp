// bgzf_scan.cpp -- walks the block headers of BGZF bytes already in host memory (SAM specification 4.1: gzip member with the
// BC extra subfield carrying BSIZE) and lists, per block, where its raw DEFLATE data lies and what it inflates to; the table
// is what k_bgzf_inflate (kernels_inflate.hip) works from.  SURVEY.md 8(f)-2; replaces the header parsing of HTSlib's
// bgzf_read_block / check_header (HTSlib is absent from the reference tree: the format is the specification's).
#include "../../include/samtools_amd.h"
#include <cstring>

extern "C" int sta_bgzf_scan(const void *bytes, uint64_t n, sta_bgzf_block *blocks, uint64_t cap, uint64_t *n_blocks, uint64_t *out_bytes)
{
    if (!bytes && n) return STA_ERR_ARG;
    const uint8_t *p = (const uint8_t *)bytes;
    uint64_t o = 0, nb = 0, total = 0;
    while (o < n) {
        if (n - o < 18) return STA_ERR_IO;
        const uint8_t *h = p + o;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return STA_ERR_IO;
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (n - o < 12 + (uint64_t)xlen + 8) return STA_ERR_IO;
        uint32_t bsize = 0; bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *s = h + 12 + x;
            const uint32_t sl = s[2] | (uint32_t)s[3] << 8;
            if (s[0] == 'B' && s[1] == 'C' && sl == 2 && x + 6 <= xlen) { bsize = (s[4] | (uint32_t)s[5] << 8) + 1u; found = true; }
            x += 4 + sl;
        }
        if (!found || bsize < 12 + xlen + 8 || n - o < bsize) return STA_ERR_IO;
        uint32_t crc, isize;
        memcpy(&crc, h + bsize - 8, 4); memcpy(&isize, h + bsize - 4, 4);
        if (isize > 65536) return STA_ERR_IO;
        if (blocks) {
            if (nb >= cap) return STA_ERR_ARG;
            sta_bgzf_block &b = blocks[nb];
            b.in_off = o + 12 + xlen; b.in_len = bsize - 12 - xlen - 8; b.out_len = isize; b.out_off = total; b.crc32 = crc; b.reserved = 0;
        }
        ++nb; total += isize; o += bsize;
    }
    if (n_blocks) *n_blocks = nb;
    if (out_bytes) *out_bytes = total;
    return STA_OK;
}
