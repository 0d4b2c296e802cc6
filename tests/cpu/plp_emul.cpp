// plp_emul.cpp -- CPU harness of the mpileup tile kernels (test infrastructure; never linked into the library, never a fallback).
//
// Runs the step functions of samtools_amd/csrc/plp_tile.h -- the code k_mplp_len_rm and k_mplp_emit_tile execute per thread --
// in plain loops over the threads of a workgroup / the lanes of a wave, with arrays standing in for LDS, and prints the pileup
// text.  tests/test_plp_emul.py diffs that text against the oracle (`mpileup -B`), so the measuring logic (depth marks, packed
// quality compare, '^' / '$' bytes, general CIGARs), the 16 x 64 tile conversion and the column walk are checked byte for byte
// without a GPU.  What it cannot show: wave-level synchronisation, LDS aliasing and the flush -- the -m gpu suite covers those.
//
//   plp_emul <dir> [min_baseQ] [tile_cap] [no_ends] [all]
// <dir> holds the staged arrays of ONE file as raw little-endian dumps (written by the test): pos.i32 end.i32 info.u32 lq.i32
// cig_off.u32 b8.u32 cigar.u32 seq.u8 qual.u8 ref.u8 and meta.txt ("n_reads n_cols tname").
// Build twice: -DPLP_EMUL_ANY=0 runs every wave-uniform branch always, =1 per lane (both are what a ballot can say).
#if PLP_EMUL_ANY
#define PLP_WAVE_ANY(x) (x)
#else
#define PLP_WAVE_ANY(x) (true)
#endif
#include "../../samtools_amd/csrc/plp_tile.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

thread_local char *plp_host_lds = nullptr;

template <class T> static std::vector<T> load(const std::string &path)
{
    std::vector<T> v;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "plp_emul: cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T) + 1);
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "plp_emul: short read %s\n", path.c_str()); exit(2); }
    fclose(f);
    v.resize((size_t)n / sizeof(T));
    return v;
}

// the straightforward column walk (file order, one entry at a time): used for the waves the tile kernel leaves to k_mplp_emit_deep
static void reference_rows(const StaReadsDev &R, const StaWinDev &W, const MplpDevPar &P, int pa, int pb, const std::vector<uint64_t> &offs, std::string &out)
{
    for (int p = pa; p < pb; ++p) {
        const uint64_t a = offs[(size_t)(p - W.col_beg)], b = offs[(size_t)(p - W.col_beg) + 1];
        if (b == a) continue;
        std::string seq, qual, mqs; unsigned cnt = 0;
        std::vector<char> buf(1 << 16);
        for (int64_t r = 0; r < R.n; ++r) {
            if (!(R.info[r] & RI_KEEP) || R.pos[r] > p || R.end[r] <= p) continue;
            Entry e; e.r = r; e.rpos = R.pos[r]; e.rend = R.end[r]; e.info = R.info[r]; e.lq = R.l_qseq[r]; e.boff = (uint64_t)R.base_off8[r] << 3;
            e.rs = resolve_general(R.cigar + R.cig_off[r], (int)(R.cig_off[r + 1] - R.cig_off[r]), e.rpos, p);
            const int c = e.rs.is_del ? placeholder_qual(R, r, e.rs.qpos, e.lq, e.boff, p) : (e.rs.qpos < e.lq ? (int)R.qual[e.boff + (uint64_t)e.rs.qpos] : 0);
            if (c < P.min_baseQ) continue;
            Sink<false> s; s.cur = 0; s.g = buf.data();
            token_write<false>(R, W, P, e, p, s);
            seq.append(buf.data(), (size_t)(s.g - buf.data()));
            qual.push_back((char)(c + 33 < 126 ? c + 33 : 126));
            { const int m = (int)((R.info[r] >> RI_MAPQ_SHIFT) & 0xff) + 33; mqs.push_back((char)(m > 126 ? 126 : m)); }
            ++cnt;
        }
        const int64_t apos = W.origin + p;
        std::string row = std::string(W.tname, (size_t)W.tname_len) + "\t" + std::to_string(apos + 1) + "\t" + ((W.ref && apos < W.ref_len) ? W.ref[apos] : 'N')
                        + "\t" + std::to_string(cnt) + "\t" + (cnt ? seq : "*") + "\t" + (cnt ? qual : "*") + (P.mq_col ? std::string("\t") + (cnt ? mqs : "*") : std::string()) + "\n";
        if (row.size() != b - a) { fprintf(stderr, "plp_emul: reference row of column %d has %zu bytes, the measuring pass said %llu\n", p, row.size(), (unsigned long long)(b - a)); exit(3); }
        memcpy(&out[a], row.data(), row.size());
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: plp_emul <dir> [min_baseQ] [tile_cap] [no_ends] [all] [mq_col]\n"); return 2; }
    const std::string d = argv[1];
    const int min_baseQ = argc > 2 ? atoi(argv[2]) : 13;
    const uint32_t tile_cap = argc > 3 ? (uint32_t)atoi(argv[3]) : 12288u;
    const int no_ends = argc > 4 ? atoi(argv[4]) : 0;
    const int all = argc > 5 ? atoi(argv[5]) : 0;
    const int mq_col = argc > 6 ? atoi(argv[6]) : 0;          // -s: the mapping-quality column rides in the tile kernels
    long long n_reads = 0, n_cols = 0; char tname[256] = "";
    { FILE *f = fopen((d + "/meta.txt").c_str(), "r"); if (!f || fscanf(f, "%lld %lld %255s", &n_reads, &n_cols, tname) != 3) { fprintf(stderr, "plp_emul: bad meta.txt\n"); return 2; } fclose(f); }
    auto pos = load<int32_t>(d + "/pos.i32"), end = load<int32_t>(d + "/end.i32"), lq = load<int32_t>(d + "/lq.i32");
    auto info = load<uint32_t>(d + "/info.u32"), cig_off = load<uint32_t>(d + "/cig_off.u32"), b8 = load<uint32_t>(d + "/b8.u32"), cigar = load<uint32_t>(d + "/cigar.u32");
    auto seq = load<uint8_t>(d + "/seq.u8"), qual = load<uint8_t>(d + "/qual.u8"), ref = load<uint8_t>(d + "/ref.u8");
    std::vector<int32_t> maxend((size_t)n_reads);
    { int32_t m = INT32_MIN; for (long long i = 0; i < n_reads; ++i) { if ((info[(size_t)i] & RI_KEEP) && end[(size_t)i] > m) m = end[(size_t)i]; maxend[(size_t)i] = m; } }

    StaReadsDev R; memset(&R, 0, sizeof R);
    R.n = n_reads; R.pos = pos.data(); R.l_qseq = lq.data(); R.cig_off = cig_off.data(); R.base_off8 = b8.data(); R.cigar = cigar.data();
    R.seq = seq.data(); R.qual_in = qual.data(); R.qual = qual.data(); R.n_bases_total = qual.size();
    R.end = end.data(); R.maxend = maxend.data(); R.info = info.data();
    StaWinDev W; memset(&W, 0, sizeof W);
    W.col_beg = 0; W.col_end = (int32_t)n_cols; W.origin = 0; W.tid = 0; W.tlen = n_cols; W.nfiles = 1; W.files = &R;
    W.ref = (const char *)ref.data(); W.ref_len = (int64_t)ref.size(); W.tname = tname; W.tname_len = (int32_t)strlen(tname);
    MplpDevPar P; memset(&P, 0, sizeof P);
    P.min_baseQ = min_baseQ; P.no_ends = no_ends; P.all = all; P.tlen = n_cols; P.tag_sep = ','; P.mq_col = mq_col;

    // wave_range_indexed (kernels_plp.hip): from the first read starting at or beyond p0, back over the (up to 64, else searched)
    // earlier reads whose prefix-maximum end still reaches p0; up to the first read starting at or beyond the next 64-column group
    auto read_range = [&](int p0, int plast, long long &rlo, long long &rhi) {
        const long long start = std::lower_bound(pos.begin(), pos.end(), p0) - pos.begin();
        const int next_group = (plast | 63) + 1;
        rhi = std::lower_bound(pos.begin(), pos.end(), next_group) - pos.begin();
        int cnt = 0;
        for (long long i = start - 64; i < start; ++i) if (i >= 0 && maxend[(size_t)i] > p0) ++cnt;
        if (cnt == 64 && start > 64) rlo = std::upper_bound(maxend.begin(), maxend.begin() + (start - 64), p0) - maxend.begin();
        else rlo = start - cnt;
        if (rlo > rhi) rlo = rhi;
    };

    // ---- k_mplp_len_rm, one workgroup (LEN_THREADS threads) per LEN_TC columns ----
    std::vector<uint32_t> line_len((size_t)n_cols);
    std::vector<uint2> colinfo((size_t)n_cols);
    for (long long c0 = 0; c0 < n_cols; c0 += LEN_TC) {
        static LenLds L;
        const int t0 = (int)c0, ntile = (int)std::min<long long>(LEN_TC, n_cols - c0), t1 = t0 + ntile;
        uint32_t total[LEN_THREADS][4]; bool any[LEN_THREADS][4];
        memset(total, 0, sizeof total); memset(any, 0, sizeof any);
        for (int t = 0; t < LEN_THREADS; ++t) len_clear(L, t);
        read_range(t0, t1 - 1, L.rlo, L.rhi);
        for (long long b0 = L.rlo; b0 < L.rhi; b0 += LEN_THREADS) {
            for (int t = 0; t < LEN_THREADS; ++t) len_step_a(L, t, R, P, t0, t1, b0);
            for (int t = 0; t < LEN_THREADS; ++t) len_step_b(L, t, R, P, t0, t1);
            for (int t = 0; t < LEN_THREADS; ++t)
                for (int gi = t >> 6; gi < L.gcount; gi += LEN_THREADS / 64) len_step_c(L, gi, t & 63, 64, b0, R, P, t0, t1);
            L.gcount = 0;
        }
        for (int t = 0; t < LEN_THREADS; ++t) len_scan_1(L, t);
        for (int t = 0; t < LEN_THREADS; ++t) len_scan_2(L, t);
        for (int t = 0; t < LEN_THREADS; ++t) len_scan_3(L, t);
        for (int t = 0; t < LEN_THREADS; ++t) len_file_result(L, t, len_scan_4(L, t), ntile, colinfo.data() + c0, total[t], any[t], P.mq_col != 0);
        for (int t = 0; t < LEN_THREADS; ++t)
            for (int i = 0; i < 4; ++i) {
                const int c = 4 * t + i;
                if (c >= ntile) continue;
                const int64_t apos = W.origin + t0 + c;
                const bool exists = any[t][i] || (P.all && apos < P.tlen);
                uint32_t len = 0;
                if (exists) len = (uint32_t)W.tname_len + 1 + (uint32_t)dec_digits((unsigned long long)(apos + 1)) + 1 + 1 + total[t][i] + 1;
                line_len[(size_t)(c0 + c)] = len | (any[t][i] ? 0x80000000u : 0u);
            }
    }
    std::vector<uint64_t> offs((size_t)n_cols + 1, 0);
    for (long long c = 0; c < n_cols; ++c) offs[(size_t)c + 1] = offs[(size_t)c] + (line_len[(size_t)c] & 0x7fffffffu);
    std::string out((size_t)offs[(size_t)n_cols], '?');

    // ---- k_mplp_emit_tile, one wave per 64 columns ----
    const uint32_t slice = (tile_cap + 48 + 15) & ~15u;
    std::vector<char> lds(slice + TILE_LDS_BYTES + 64);
    long long n_tile = 0, n_deep = 0;
    for (long long c0 = 0; c0 < n_cols; c0 += 64) {
        const long long c1 = std::min<long long>(c0 + 64, n_cols);
        const int p0 = (int)c0, plast = (int)c1 - 1;
        const uint64_t o0 = offs[(size_t)c0], o1 = offs[(size_t)c1], wbytes = o1 - o0;
        if (wbytes == 0) continue;
        if (wbytes > tile_cap) { reference_rows(R, W, P, p0, (int)c1, offs, out); ++n_deep; continue; }
        ++n_tile;
        std::fill(lds.begin(), lds.end(), (char)0x55);
        plp_host_lds = lds.data();
        const uint32_t base = 0, mis = (uint32_t)(o0 & 15), dump = base + slice - 8;
        TileLds &T = *reinterpret_cast<TileLds *>(lds.data() + base + slice);
        TileLane st[64];
        for (int lane = 0; lane < 64; ++lane) {
            const bool active = p0 + lane < W.col_end;
            const uint64_t my0 = active ? offs[(size_t)(c0 + lane)] : o1, my1 = active ? offs[(size_t)(c0 + lane + 1)] : o1;
            tile_row_head(T, st[lane], lane, W, base + mis + (uint32_t)(my0 - o0), my1 > my0, W.origin + p0 + lane);
        }
        for (int lane = 0; lane < 2 * TILE_SLOTS; ++lane) tile_zero_column(T, lane);
        for (int lane = 0; lane < 4; ++lane) tile_refpack(T, lane);
        long long rlo, rhi;
        read_range(p0, plast, rlo, rhi);
        for (int lane = 0; lane < 64; ++lane) tile_file_head(st[lane], lane, st[lane].exists ? colinfo[(size_t)(c0 + lane)] : make_uint2(0u, 0u), dump, P.mq_col != 0);
        for (long long b0 = rlo; b0 < rhi; b0 += 64) {
            std::vector<int> live;
            for (int lane = 0; lane < 64; ++lane) {
                const long long ri = b0 + lane;
                if (ri < rhi && tile_read_is_live(info[(size_t)ri], pos[(size_t)ri], end[(size_t)ri], p0, plast)) live.push_back(lane);
            }
            const int nlive = (int)live.size();
            for (int first = 0; first < nlive; first += TILE_SLOTS) {
                const int ns = std::min(TILE_SLOTS, nlive - first);
                for (int i = 0; i < ns; ++i) { const long long ri = b0 + live[(size_t)(first + i)]; tile_set_slot(T, i, live[(size_t)(first + i)], ri, info[(size_t)ri], pos[(size_t)ri], end[(size_t)ri], b8[(size_t)ri]); }
                unsigned long long sm = 0;
                for (int lane = 0; lane < 64; ++lane) if (tile_phase1(T, lane, ns, R, P, p0, W.ref != nullptr) && (lane & 3) == 0) sm |= 1ull << lane;
                for (int s = 0; s < ns;) {
                    if (s + 4 <= ns && ((sm >> (4 * s)) & 0x1111ull) == 0x1111ull) {
                        for (int lane = 0; lane < 64; ++lane) tile_phase2_rows4(T, s, st[lane].col, st[lane].cur_s, st[lane].cur_q, st[lane].mq_d);
                        s += 4; continue;
                    }
                    for (int lane = 0; lane < 64; ++lane) {
                        if ((sm >> (4 * s)) & 1ull) tile_phase2_row(T, s, st[lane].col, st[lane].cur_s, st[lane].cur_q, st[lane].mq_d);
                        else tile_phase2_mixed(T, s, st[lane], R, W, P, b0, p0 + lane, p0);
                    }
                    ++s;
                }
            }
        }
        for (int lane = 0; lane < 64; ++lane) { tile_file_tail(st[lane]); if (st[lane].exists) lds[st[lane].cur] = '\n'; }
        memcpy(&out[(size_t)o0], lds.data() + base + mis, (size_t)wbytes);
    }
    fwrite(out.data(), 1, out.size(), stdout);
    fprintf(stderr, "plp_emul: %lld waves through the tile functions, %lld through the reference walk (rows beyond %u bytes)\n", n_tile, n_deep, tile_cap);
    return 0;
}
