// host_scan.cpp -- sta_io_scan: run a file through the drivers' reader (and optionally pump + stager) without a device.
// Test / benchmark hook for the host plumbing either side of the engine.
#include "../../include/samtools_amd.h"
#include "host_io.h"
#include "host_pump.h"
#include "host_stage.h"
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
    readers.push_back(AlnReader::open(path, &err, threads));
    if (!readers[0]) return -1;
    Fnv f; uint64_t n = 0;
    if (!stage) {
        Rec r; int st;
        while ((st = readers[0]->next(r)) > 0) { fold(f, r); ++n; }
        if (st < 0) return -2;
    } else {
        // what driver_mpileup does between the reader and sta_stage_window, minus the device
        PumpConfig pc; pc.window_cols = 1 << 20;
        Pump pump(readers, pc);
        std::vector<std::vector<const Rec *>> reads;
        StagedFile sf;
        for (;;) {
            int tid = pump.next_tid();
            if (pump.error() || tid < 0) break;
            int64_t cursor = pump.next_pos(tid);
            for (;;) {
                if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
                if (!pump.has_carry()) cursor = std::max(cursor, pump.next_pos(tid));
                int64_t ce = pump.fill(tid, cursor, cursor + pc.window_cols, reads);
                if (pump.error()) break;
                sf.clear();
                for (const Rec *r : reads[0]) { if (r->pos >= cursor) ++n; sf.add(*r, cursor, nullptr, nullptr); }
                sf.finish();
                f.u64((uint64_t)sf.n()); f.bytes(sf.pos.data(), sf.pos.size() * 4); f.bytes(sf.cigar.data(), sf.cigar.size() * 4);
                f.bytes(sf.qual.data(), sf.qual.size()); f.bytes(sf.seq.data(), sf.seq.size()); f.bytes(sf.names.data(), sf.names.size());
                pump.retire(ce);
                cursor = ce;
            }
            pump.drop_tid_carry();
        }
        if (pump.error()) return -2;
    }
    if (n_records) *n_records = n;
    if (checksum) *checksum = f.h;
    return 0;
}
