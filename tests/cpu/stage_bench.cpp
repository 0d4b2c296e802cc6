// stage_bench.cpp -- the producer side of the drivers' window loop (decode threads -> ChunkPump -> StagedFile windows) without a
// device: how fast can the host hand windows over?  Test / measurement infrastructure (scripts/stage_bench.sh), not shipped.
//   stage_bench in.bam [io_threads=8] [window_cols=1048576] [mpileup=0]
#include "../../samtools_amd/csrc/host_chunk.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace sta;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: stage_bench in.bam [io_threads] [window_cols] [mpileup]\n"); return 2; }
    const int threads = argc > 2 ? atoi(argv[2]) : 8;
    const int64_t wcols = argc > 3 ? atoll(argv[3]) : (1 << 20);
    const bool mplp = argc > 4 && atoi(argv[4]);
    std::string err;
    std::vector<std::unique_ptr<AlnReader>> readers;
    auto r = AlnReader::open(argv[1], &err, threads);
    if (!r) { fprintf(stderr, "open: %s\n", err.c_str()); return 1; }
    if (argc > 5 && atoi(argv[5])) {
        // the serial section of the chunk lane alone: raw record groups off the BGZF stream
        sta::pvector<uint8_t> raw; int64_t nrec = 0, tot = 0, groups = 0; size_t bytes = 0;
        const double a = now();
        while (r->raw_group(raw, 1 << 20, &nrec) > 0) { tot += nrec; bytes += raw.size(); ++groups; }
        printf("raw_group: %lld groups, %lld records, %zu bytes in %.3f s\n", (long long)groups, (long long)tot, bytes, now() - a);
        return 0;
    }
    readers.push_back(std::move(r));
    PumpConfig pc; pc.window_cols = wcols; pc.use_endpos = !mplp;
    if (getenv("STA_FAKE_GPU_INFLATE")) pc.inflate_device = 0;       // (tests/cpu/gpu_inflate_stub.cpp: the feeder / parser threading without a GPU)
    if (mplp) { pc.tpl = PumpConfig::TPL_MPLP; pc.pushed = [](const Rec &rec) { return !(rec.flag & 4); }; }
    const double t0 = now();
    ChunkPump pump(readers, pc, threads);
    std::vector<StagedFile> ring[3];
    int64_t windows = 0, reads = 0, bases = 0, mates = 0; uint64_t sum = 0; double t_fill = 0, t_pair = 0;
    for (;;) {
        const int tid = pump.next_tid();
        if (tid < 0 || pump.error()) break;
        int64_t cursor = pump.next_pos(tid);
        for (;;) {
            if (pump.next_pos(tid) == INT64_MAX && !pump.has_carry()) break;
            std::vector<StagedFile> &st = ring[windows % 3];
            const double a = now();
            const int64_t ce = pump.fill_staged(tid, cursor, cursor + wcols, st);
            t_fill += now() - a;
            if (pump.error()) break;
            if (mplp) {
                // what driver_mpileup.cpp does on the producer thread behind fill_staged(): the window's new reads visit the overlap hash
                const double b = now();
                pump.pair_staged(st);
                t_pair += now() - b;
                for (int32_t m : st[0].mate) mates += m >= 0;
            }
            const sta_reads v = st[0].view();
            reads += v.n_reads; bases += (int64_t)v.n_bases_total;
            if (v.n_bases_total) sum += v.qual[0] + v.qual[v.n_bases_total - 1] + v.seq[0];
            ++windows;
            pump.retire(ce);
            cursor = ce > cursor ? ce : cursor + 1;
        }
        pump.drop_tid_carry();
    }
    const double dt = now() - t0;
    if (pump.error()) { fprintf(stderr, "error: %s\n", pump.error_text()); return 1; }
    printf("decode wait %.3f s, staging %.3f s | ", pump.stats().wait_s, pump.stats().stage_s);
    printf("windows %lld reads %lld padded bases %lld  wall %.3f s  fill_staged %.3f s  = %.0f Mbases/s (checksum %llu)\n",
           (long long)windows, (long long)reads, (long long)bases, dt, t_fill, bases / 1e6 / dt, (unsigned long long)sum);
    if (mplp) printf("pair_staged %.3f s = %.1f ns per read, %lld reads found a partner\n", t_pair, t_pair * 1e9 / (double)(reads ? reads : 1), (long long)mates);
    return 0;
}
