// tests/cpu/inflate_emul.cpp -- TEST INFRASTRUCTURE: the wave functions of samtools_amd/csrc/bgzf_inflate_dev.h (what every wave of
// the device kernel k_bgzf_inflate executes) run on the CPU as one lane, against zlib: raw deflate streams of every block type
// (stored, fixed, dynamic; levels 0-9, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE), sizes 0 .. 65 280, contents from incompressible to one
// repeated byte, BAM-like records, and damaged streams (which must end with a non-zero status, never out of bounds).
//   g++ -O1 -std=c++17 -I samtools_amd/csrc tests/cpu/inflate_emul.cpp -lz -o inflate_emul ; inflate_emul [rounds] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <zlib.h>
#include "bgzf_inflate_dev.h"

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &in, int level, int strategy)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, in.size()) + 64);
    zs.next_in = const_cast<Bytef *>(in.data()); zs.avail_in = (uInt)in.size();
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 1u);
    auto rnd = [&](int n) { return (int)(rng() % (unsigned)n); };
    static bgzi::Lds L;
    bgzi::Consts C; bgzi::fill_consts(C);
    long n_ok = 0, n_bad = 0, n_damaged = 0, n_damaged_caught = 0;
    for (int it = 0; it < rounds; ++it) {
        const int kind = it % 8;
        int n = it % 11 == 0 ? rnd(40) : it % 5 == 0 ? 65280 - rnd(3) : 1 + rnd(65280);
        if (it == 0) n = 0;
        std::vector<uint8_t> data((size_t)n);
        if (kind == 0) for (auto &c : data) c = (uint8_t)rng();
        else if (kind == 1) for (auto &c : data) c = (uint8_t)"ACGT"[rnd(4)];
        else if (kind == 2) { uint8_t v = (uint8_t)rnd(256); for (auto &c : data) { if (rnd(200) == 0) v = (uint8_t)rnd(256); c = v; } }
        else if (kind == 3) { for (int i = 0; i < n; ++i) data[(size_t)i] = (uint8_t)(i < 40 || rnd(30) == 0 ? rnd(256) : data[(size_t)(i - 1 - rnd(i < 300 ? i : 300))]); }
        else if (kind == 4) { const int per = 2 + rnd(70); for (int i = 0; i < n; ++i) data[(size_t)i] = (uint8_t)(i < per ? rnd(256) : data[(size_t)(i - per)]); }
        else if (kind == 5) {
            // BAM-like: records of a fixed part, a counting name, packed bases close to an earlier record's, plateaued qualities
            size_t o = 0; int rec = 0;
            while (o < data.size()) {
                char nm[32]; const int ln = snprintf(nm, sizeof nm, "read_%07d", 1000000 + rec++);
                uint8_t r[400]; int k = 0;
                for (int i = 0; i < 36; ++i) r[k++] = (uint8_t)(i < 8 ? i * 7 : i < 12 ? rec >> (8 * (i - 8)) : 0x11 * (i & 3));
                memcpy(r + k, nm, (size_t)ln + 1); k += ln + 1;
                for (int i = 0; i < 75; ++i) r[k++] = (uint8_t)(o > 400 && rnd(50) ? data[o - 300 + (size_t)i] : rnd(256));
                for (int i = 0; i < 150; ++i) r[k++] = (uint8_t)(rnd(10) ? 37 : 2 + rnd(38));
                for (int i = 0; i < k && o < data.size(); ++i) data[o++] = r[i];
            }
        } else if (kind == 6) for (int i = 0; i < n; ++i) data[(size_t)i] = (uint8_t)((i * 2654435761u) >> (24 + rnd(2)));
        else {
            // stretches copied from 17 000 .. 32 000 bytes back: matches that reach behind the decoder's 16 KiB ring
            for (int i = 0; i < n;) {
                const int back = 17000 + rnd(15000), run = 20 + rnd(400);
                for (int k = 0; k < run && i < n; ++k, ++i) data[(size_t)i] = (uint8_t)(i >= back && rnd(40) ? data[(size_t)(i - back)] : rnd(256));
            }
        }
        const int level = it % 13 == 0 ? 0 : 1 + rnd(9);
        const int strat = it % 17 == 3 ? Z_FIXED : it % 19 == 4 ? Z_HUFFMAN_ONLY : it % 23 == 5 ? Z_RLE : Z_DEFAULT_STRATEGY;
        std::vector<uint8_t> comp = deflate_raw(data, level, strat);
        const size_t clen = comp.size();
        const bool damage = it % 9 == 8 && clen > 4;
        if (damage) { ++n_damaged; const int k = 1 + rnd(3); for (int j = 0; j < k; ++j) comp[(size_t)rnd((int)clen)] ^= (uint8_t)(1 << rnd(8)); }
        comp.resize(clen + 2048, 0);
        // the destination: `align` = its address mod 16 (a 16-aligned buffer entered at that offset), guard bytes either side
        const uint32_t align = (uint32_t)rnd(16);
        alignas(16) static uint8_t outbuf[65536 + 64];
        memset(outbuf, 0xee, sizeof outbuf);
        uint8_t *dst = outbuf + 16 + align;
        const int st = bgzi::inflate_block(L, C, comp.data(), (int64_t)clen, (uint32_t)n, align, dst);
        if (damage) {
            if (st != 0) ++n_damaged_caught;
            else if (n && memcmp(dst, data.data(), (size_t)n) != 0) { /* a damaged stream may decode to other bytes of the right size: the CRC catches that */ }
            if (dst[n] != 0xee || dst[-1] != 0xee) { fprintf(stderr, "OUT OF BOUNDS write on a damaged stream, it %d\n", it); ++n_bad; }
            continue;
        }
        if (st != 0 || (n && memcmp(dst, data.data(), (size_t)n) != 0) || dst[n] != 0xee || dst[-1] != 0xee) {
            if (n_bad < 5) fprintf(stderr, "MISMATCH it %d kind %d n %d level %d strat %d clen %zu status %d\n", it, kind, n, level, strat, clen, st);
            ++n_bad;
        } else ++n_ok;
    }
    printf("streams %ld identical, %ld wrong; damaged %ld (status != 0 for %ld)\n", n_ok, n_bad, n_damaged, n_damaged_caught);
    return n_bad ? 1 : 0;
}
