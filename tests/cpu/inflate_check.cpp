// inflate_check.cpp -- the drivers' block decoder (samtools_amd/csrc/host_inflate.h) against zlib.  Test infrastructure.
//   inflate_check file.bam|file.gz ...     every BGZF block of the files: same bytes, same CRC as zlib
//   inflate_check --gen N seed             N generated deflate streams (stored / fixed / dynamic, every level and strategy, data of several kinds)
//   inflate_check --fuzz N seed file.bam   N damaged copies of the file's blocks: no crash; wherever zlib accepts the stream, the same bytes
#include "../../samtools_amd/csrc/host_inflate.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <zlib.h>
using namespace sta;

static bool zlib_inflate(const uint8_t *in, size_t n, std::vector<uint8_t> &out)
{
    out.resize(1 << 16);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n; zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    const bool ok = rc == Z_STREAM_END && zs.avail_in == 0;
    inflateEnd(&zs);
    return ok;
}

// one comparison; strict: zlib accepted the stream, so the fast decoder must too
static long n_checked = 0, n_bytes = 0, n_rejected = 0;
static bool check(const uint8_t *in, size_t n, bool strict)
{
    std::vector<uint8_t> want;
    const bool zok = zlib_inflate(in, n, want);
    std::vector<uint8_t> inbuf(n + 8, 0xA5); memcpy(inbuf.data(), in, n);      // exactly the slack the contract promises
    std::vector<uint8_t> got((1 << 16) + 16, 0x5A);
    size_t gl = 0;
    const int rc = fast_inflate(inbuf.data(), n, got.data(), 1 << 16, &gl);
    ++n_checked;
    if (zok) {
        if (rc != 0 || gl != want.size() || memcmp(got.data(), want.data(), gl) != 0) {
            fprintf(stderr, "MISMATCH: zlib ok (%zu bytes), fast rc=%d len=%zu\n", want.size(), rc, gl);
            return false;
        }
        if (fast_crc32(got.data(), gl) != (uint32_t)crc32(crc32(0L, Z_NULL, 0), want.data(), (uInt)want.size())) { fprintf(stderr, "CRC MISMATCH\n"); return false; }
        n_bytes += (long)gl;
    } else {
        ++n_rejected;
        if (strict) { fprintf(stderr, "zlib rejected a stream that was expected to be valid\n"); return false; }
    }
    return true;
}

static bool each_block(const std::vector<uint8_t> &raw, std::vector<std::pair<size_t, size_t>> &blocks)
{
    size_t p = 0;
    while (p + 18 <= raw.size()) {
        if (raw[p] != 0x1f || raw[p + 1] != 0x8b) return false;
        const size_t xlen = raw[p + 10] | (size_t)raw[p + 11] << 8;
        if (xlen != 6 || raw[p + 12] != 'B' || raw[p + 13] != 'C') return false;       // (the files of the test-suite: one BC subfield)
        const size_t bs = (raw[p + 16] | (size_t)raw[p + 17] << 8) + 1;
        if (p + bs > raw.size() || bs < 26) return false;
        blocks.push_back({ p + 18, bs - 26 });
        p += bs;
    }
    return true;
}

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v; FILE *f = fopen(path, "rb");
    if (!f) return v;
    uint8_t buf[1 << 16]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc >= 4 && !strcmp(argv[1], "--gen")) {
        const int n = atoi(argv[2]); std::mt19937_64 rng((uint64_t)atoll(argv[3]));
        for (int it = 0; it < n; ++it) {
            const size_t len = (size_t)(rng() % 65281);                                   // 0 .. 0xff00
            std::vector<uint8_t> data(len);
            const int kind = (int)(rng() % 6);
            for (size_t i = 0; i < len; ++i) {
                switch (kind) {
                case 0: data[i] = (uint8_t)rng(); break;                                  // incompressible
                case 1: data[i] = (uint8_t)"ACGT"[rng() & 3]; break;                      // four symbols
                case 2: data[i] = (uint8_t)(i % 7 == 0 ? rng() : 'x'); break;             // long runs (distance 1)
                case 3: data[i] = (uint8_t)(i >= 300 && (rng() % 10) ? data[i - 300 + (rng() % 3)] : rng() % 40 + 33); break;      // matches at medium distance
                case 4: data[i] = (uint8_t)(i >= 30000 && (rng() % 50) ? data[i - 30000] : rng()); break;                           // long distances
                default: data[i] = (uint8_t)((i * 2654435761u) >> 24); break;
                }
            }
            const int level = (int)(rng() % 10), strategy = (int[]){ Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED }[rng() % 5];
            z_stream zs; memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, level, Z_DEFLATED, -15, (int)(1 + rng() % 9), strategy);
            std::vector<uint8_t> comp(deflateBound(&zs, (uLong)len) + len / 8 + 4096);     // (room for the flush markers of the piecewise runs)
            zs.next_in = data.data(); zs.avail_in = (uInt)len; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
            // sometimes in pieces with full flushes: several blocks, stored blocks in between
            if (rng() % 3 == 0 && len > 100) { zs.avail_in = (uInt)(len / 3); deflate(&zs, Z_FULL_FLUSH); zs.avail_in = (uInt)(len - len / 3); }
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate failed\n"); return 1; }
            const size_t cl = comp.size() - zs.avail_out;
            deflateEnd(&zs);
            if (!check(comp.data(), cl, true)) { fprintf(stderr, "generated stream %d (kind %d level %d strategy %d len %zu)\n", it, kind, level, strategy, len); return 1; }
        }
        printf("gen: %ld streams, %ld bytes identical\n", n_checked, n_bytes);
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "--fuzz")) {
        const int n = atoi(argv[2]); std::mt19937_64 rng((uint64_t)atoll(argv[3]));
        const std::vector<uint8_t> raw = slurp(argv[4]);
        std::vector<std::pair<size_t, size_t>> blocks;
        if (!each_block(raw, blocks) || blocks.empty()) { fprintf(stderr, "not a BGZF file of the expected kind\n"); return 1; }
        for (int it = 0; it < n; ++it) {
            const auto &b = blocks[rng() % blocks.size()];
            std::vector<uint8_t> d(raw.begin() + (long)b.first, raw.begin() + (long)(b.first + b.second));
            const int how = (int)(rng() % 4);
            if (how == 0 && !d.empty()) { for (int k = 0, m = 1 + (int)(rng() % 4); k < m; ++k) d[rng() % d.size()] ^= (uint8_t)(1u << (rng() % 8)); }
            else if (how == 1 && !d.empty()) d.resize(rng() % d.size());                                  // truncated
            else if (how == 2 && !d.empty()) { const size_t a = rng() % d.size(); for (size_t i = a; i < d.size() && i < a + 16; ++i) d[i] = (uint8_t)rng(); }
            else { d.resize(d.size() + rng() % 32, (uint8_t)rng()); }                                     // trailing garbage
            if (!check(d.data(), d.size(), false)) { fprintf(stderr, "fuzz case %d (how %d)\n", it, how); return 1; }
        }
        printf("fuzz: %ld streams, zlib rejected %ld, the rest identical (%ld bytes)\n", n_checked, n_rejected, n_bytes);
        return 0;
    }
    for (int a = 1; a < argc; ++a) {
        const std::vector<uint8_t> raw = slurp(argv[a]);
        std::vector<std::pair<size_t, size_t>> blocks;
        if (!each_block(raw, blocks)) { fprintf(stderr, "%s: not a BGZF file of the expected kind\n", argv[a]); return 1; }
        for (auto &b : blocks) if (!check(raw.data() + b.first, b.second, true)) { fprintf(stderr, "%s: block at %zu\n", argv[a], b.first); return 1; }
    }
    printf("files: %ld blocks, %ld bytes identical\n", n_checked, n_bytes);
    return 0;
}
