// host_scan.cpp -- sta_io_scan: run a file through the drivers' reader (and optionally pump + stager) without a device.
// Test / benchmark hook for the host plumbing either side of the engine.
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_bamout.h"
#include "host_pump.h"
#include "host_stage.h"
#include "host_chunk.h"
#include "host_bgzf.h"
#include <cstdlib>
#include <climits>

using namespace sta;

namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void u64(uint64_t x) { h = (h ^ x) * 1099511628211ull; }
    void bytes(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull; u64(n); }
};
void fold(Fnv &f, const Rec &r)
{
    f.u64((uint64_t)(int64_t)r.tid); f.u64((uint64_t)r.pos); f.u64(r.flag); f.u64(r.mapq); f.u64((uint64_t)r.l_qseq);
    f.u64((uint64_t)(int64_t)r.mtid); f.u64((uint64_t)r.mpos); f.u64((uint64_t)r.isize); f.u64((uint64_t)r.rlen);
    f.bytes(r.qname.data(), r.qname.size()); f.bytes(r.cigar.data(), r.cigar.size() * 4);
    f.bytes(r.seq.data(), r.seq.size()); f.bytes(r.qual.data(), r.qual.size());
    f.u64(r.has_bq); f.u64(r.has_zq); if (r.has_bq) f.bytes(r.bq.data(), r.bq.size()); f.bytes(r.rg.data(), r.rg.size());
}
}  // namespace

extern "C" int sta_io_scan(const char *path, int threads, int stage, uint64_t *n_records, uint64_t *checksum)
{
    if (!path) return -1;
    std::string err;
    std::vector<std::unique_ptr<AlnReader>> readers;
    {
        // several inputs (the drivers' multi-file windows): paths separated by '\n'
        std::string all(path);
        size_t a = 0;
        while (a <= all.size()) {
            size_t b = all.find('\n', a);
            if (b == std::string::npos) b = all.size();
            if (b > a) { readers.push_back(AlnReader::open(all.substr(a, b - a), &err, threads)); if (!readers.back()) return -1; }
            a = b + 1;
        }
        if (readers.empty()) return -1;
    }
    Fnv f; uint64_t n = 0;
    if (!stage) {
        for (auto &rd : readers) {
            Rec r; int st;
            while ((st = rd->next(r)) > 0) { fold(f, r); ++n; }
            if (st < 0) return -2;
        }
    } else {
        // what driver_mpileup does between the reader and sta_stage_window, minus the device: stage 1 = one decoded record
        // at a time (Pump), stage 2 = chunk slices (ChunkPump); both must stage byte-identical windows.
        PumpConfig pc; pc.window_cols = 1 << 20; pc.nref_limit = readers[0]->header().nref();
        if (const char *e = getenv("STA_WINDOW_COLS")) pc.window_cols = std::max<long long>(1, atoll(e));
        if (const char *e = getenv("STA_WINDOW_READS")) pc.max_reads = std::max<long long>(1, atoll(e));
        // the lanes' template state (host_names.h) is part of the comparison: mpileup's overlap hash by default, depth -s's name hash
        // with STA_SCAN_DEPTH_S=1
        if (getenv("STA_SCAN_DEPTH_S")) { pc.tpl = PumpConfig::TPL_DEPTH; pc.use_endpos = true; pc.depth_filter.flag = 4 | 256 | 512 | 1024; }
        else { pc.tpl = PumpConfig::TPL_MPLP; pc.pushed = [](const Rec &r) { return !(r.flag & 0x704) && r.mapq >= 1; }; }
        std::unique_ptr<WindowSource> src;
        if (stage == 2) src.reset(new ChunkPump(readers, pc, threads > 0 ? threads : io_default_threads()));
        else src.reset(new Pump(readers, pc));
        WindowSource &pump = *src;
        std::vector<StagedFile> staged;
        for (;;) {
            int tid = pump.next_tid();
            if (pump.error() || tid < 0) break;
            int64_t cursor = pump.next_pos(tid);
            for (;;) {
                if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
                cursor = std::max(cursor, std::min(pump.carry_next_covered(cursor), pump.next_pos(tid)));
                int64_t ce = pump.fill_staged(tid, cursor, cursor + pc.window_cols, staged);
                if (pump.error()) break;
                pump.pair_staged(staged);
                if (pump.next_pos(tid) == INT64_MAX) {
                    int64_t me = pump.carry_max_end();
                    if (me != INT64_MIN) ce = std::min(ce, std::max(me, cursor));
                }
                static const bool nosum = getenv("STA_SCAN_NOSUM") != nullptr;     // timing runs: skip the (byte-serial) checksum
                f.u64((uint64_t)tid); f.u64((uint64_t)cursor); f.u64((uint64_t)ce);
                for (size_t fi = 0; fi < staged.size(); ++fi) {
                    const StagedFile &sf = staged[fi];
                    for (size_t i = 0; i < sf.pos.size(); ++i) if (sf.pos[i] >= 0 && !(sf.aux[i] & STA_AUX_ACCEPTED)) ++n;
                    f.u64((uint64_t)sf.n());
                    if (nosum) continue;
                    f.bytes(sf.pos.data(), sf.pos.size() * 4); f.bytes(sf.flag.data(), sf.flag.size() * 2); f.bytes(sf.mapq.data(), sf.mapq.size());
                    f.bytes(sf.aux.data(), sf.aux.size()); f.bytes(sf.l_qseq.data(), sf.l_qseq.size() * 4); f.bytes(sf.mtid.data(), sf.mtid.size() * 4);
                    f.bytes(sf.mpos.data(), sf.mpos.size() * 8); f.bytes(sf.isize.data(), sf.isize.size() * 4);
                    f.bytes(sf.cig_off.data(), sf.cig_off.size() * 4); f.bytes(sf.base_off8.data(), sf.base_off8.size() * 4); f.bytes(sf.name_off.data(), sf.name_off.size() * 4);
                    f.bytes(sf.cigar.data(), sf.cigar.size() * 4); f.bytes(sf.qual.data(), sf.qual.size()); f.bytes(sf.seq.data(), sf.seq.size());
                    f.bytes(sf.names.data(), sf.names.size()); f.u64(sf.any_bq); if (sf.any_bq) f.bytes(sf.bq.data(), sf.bq.size());
                    f.u64((uint64_t)sf.tpl); f.bytes(sf.clip.data(), sf.clip.size() * 8); f.bytes(sf.mate.data(), sf.mate.size() * 4);
                    uint64_t spans = 0;
                    for (size_t i = 0; i < sf.pos.size(); ++i) spans = spans * 3 + (pump.staged_has_span(fi, i) ? 1 : 0);
                    f.u64(spans);
                }
                // pretend the depth cap removed every 97th read with a span, so that drop() is part of the comparison
                if (getenv("STA_SCAN_DROP"))
                    for (size_t fi = 0; fi < staged.size(); ++fi) {
                        const StagedFile &sf = staged[fi];
                        std::vector<char> dr(sf.pos.size(), 0);
                        for (size_t i = 0; i < dr.size(); ++i) dr[i] = (i % 97 == 96) && !(sf.aux[i] & STA_AUX_ACCEPTED) && pump.staged_has_span(fi, i);
                        pump.drop(fi, dr);
                    }
                pump.retire(ce);
                cursor = std::max(cursor, ce);
            }
            if (pump.error()) break;
            pump.drop_tid_carry();
        }
        if (pump.error()) return -2;
    }
    if (n_records) *n_records = n;
    if (checksum) *checksum = f.h;
    return 0;
}

// sta_io_scan for one region of one file, the way a `-r` run reads it: with a BAI beside the BAM the reader starts at the linear
// index's offset for the region start and stops at the first record beyond the region; without one it filters the whole file.
// Both must deliver the same records (count + checksum).  *used_index says which happened.
extern "C" int sta_io_scan_region(const char *path, const char *region, int threads, int use_index, uint64_t *n_records, uint64_t *checksum, int *used_index)
{
    if (!path || !region) return -1;
    std::string err;
    std::unique_ptr<AlnReader> rd = AlnReader::open(path, &err, threads);
    if (!rd) return -1;
    int tid; int64_t beg, end;
    if (!parse_region(rd->header(), region, &tid, &beg, &end)) return -3;
    rd->set_region(tid, beg, end);
    bool used = false;
    if (use_index) {
        std::unique_ptr<BaiIndex> ix = BaiIndex::load_for(path);
        if (ix && ix->older_than_data()) fprintf(stderr, "[W::samtools_amd] The index file is older than the data file: %s\n", path);
        if (ix) {                                           // (an older index is used with HTSlib's warning: driver_shard.h seek_readers_by_index)
            const uint64_t v = ix->start_offset(tid, beg);
            if (v == UINT64_MAX) { if (n_records) *n_records = 0; if (checksum) *checksum = Fnv().h; if (used_index) *used_index = 1; return 0; }
            used = rd->seek_voffset(v);
        }
    }
    Fnv f; uint64_t n = 0;
    Rec r; int st;
    while ((st = rd->next(r)) > 0) { fold(f, r); ++n; }
    if (st < 0) return -2;
    if (n_records) *n_records = n;
    if (checksum) *checksum = f.h;
    if (used_index) *used_index = used ? 1 : 0;
    return 0;
}

// The same through the BAM writer (host_bamout.h): level 0 = stored blocks (calmd -u), otherwise compressed (-b).  Host only.
extern "C" int sta_io_write_bam(const char *path, const char *out_path, int level)
{
    if (!path || !out_path) return STA_ERR_ARG;
    std::string err;
    auto rd = AlnReader::open(path, &err);
    if (!rd) return STA_ERR_IO;
    rd->set_keep_aux(true);
    FILE *fo = fopen(out_path, "wb");
    if (!fo) return STA_ERR_IO;
    const Header &h = rd->header();
    BamWriter bw(fo, level);
    bool ok = bw.header(h, h.text);
    Rec r;
    int st = 1;
    while (ok && (st = rd->next(r)) > 0) ok = bw.record(h, r, r.seq.data(), r.qual.data(), r.auxv);
    ok = ok && bw.close();
    const bool bad = fclose(fo) != 0;
    return st < 0 || bad || !ok ? STA_ERR_IO : STA_OK;
}

// Every record of `path` read with the drivers' reader and written back as SAM text behind the header: what calmd's writer does to
// a record it does not change (the aux fields go through their sam_format1 text form, host_io.h Rec::auxv).  Host only.
extern "C" int sta_io_write_sam(const char *path, const char *out_path)
{
    if (!path || !out_path) return STA_ERR_ARG;
    std::string err;
    auto rd = AlnReader::open(path, &err);
    if (!rd) return STA_ERR_IO;
    rd->set_keep_aux(true);
    FILE *fo = fopen(out_path, "w");
    if (!fo) return STA_ERR_IO;
    const Header &h = rd->header();
    fwrite(h.text.data(), 1, h.text.size(), fo);
    if (!h.text.empty() && h.text.back() != '\n') fputc('\n', fo);
    Rec r; std::string line;
    int st;
    while ((st = rd->next(r)) > 0) {
        format_sam_record(h, r, r.seq.data(), r.qual.data(), r.auxv, line);
        fwrite(line.data(), 1, line.size(), fo);
    }
    const bool bad = fclose(fo) != 0;
    return st < 0 || bad ? STA_ERR_IO : STA_OK;
}

// Every contig of a reference FASTA through the drivers' loader (host_io.h Fasta: index-driven when a .fai lies beside a plain file,
// whole-file parsing otherwise; STA_FASTA_WHOLE=1 forces the latter): count, total bases and a checksum over names and bases in file
// order; order != 0 fetches the contigs from the last to the first instead (the read-ahead then never helps).  *lazy = 1 when the
// index was used.  Host only.
extern "C" int sta_io_fasta_scan(const char *path, int order, uint64_t *n_contigs, uint64_t *n_bases, uint64_t *checksum, int *lazy)
{
    if (!path) return STA_ERR_ARG;
    auto fa = Fasta::load(path);
    if (!fa) return STA_ERR_IO;
    const std::vector<std::string> &names = fa->names();
    std::vector<uint64_t> h(names.size(), 0);
    uint64_t nb = 0;
    for (size_t k = 0; k < names.size(); ++k) {
        const size_t i = order ? names.size() - 1 - k : k;
        const std::string *s = fa->fetch(names[i]);
        if (!s) return STA_ERR_IO;
        Fnv f; f.bytes(names[i].data(), names[i].size()); f.u64(s->size()); f.bytes(s->data(), s->size());
        h[i] = f.h; nb += s->size();
    }
    Fnv all; for (uint64_t x : h) all.u64(x);
    if (n_contigs) *n_contigs = names.size();
    if (n_bases) *n_bases = nb;
    if (checksum) *checksum = all.h;
    if (lazy) *lazy = fa->lazy() ? 1 : 0;
    return STA_OK;
}
