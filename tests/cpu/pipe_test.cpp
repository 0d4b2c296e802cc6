// Unit test of the drivers' window pipeline (samtools_amd/csrc/driver_pipeline.h) with a fake device stage: submission order is
// the output order, held jobs can be submitted twice, wait() sees the device results, errors stop the output.  No GPU needed.
#include "../../samtools_amd/csrc/driver_pipeline.h"
#include <cassert>
#include <cstring>
#include <string>
#include <unistd.h>
using namespace sta;

int main()
{
    char path[] = "/tmp/sta_pipe_testXXXXXX";
    int fd = mkstemp(path);
    FILE *out = fdopen(fd, "w+");
    std::string want;
    {
        WinPipe pipe(3, [](WinJob &j, int) {
            if (j.tid == 777) return -1;
            usleep((useconds_t)((j.cb * 37) % 700));                  // uneven device times
            char buf[64]; int n = snprintf(buf, sizeof buf, "win %lld all=%d\n", (long long)j.cb, j.all_mode);
            j.text.assign(buf, buf + n); j.out_bytes = (uint64_t)n; j.info.n_data_cols = (uint64_t)(j.cb % 3);
            return 0;
        }, out, "write error\n", 2);
        for (int k = 0; k < 200; ++k) {
            WinJob *j = pipe.acquire();
            j->tid = 0; j->cb = k; j->ce = k + 1; j->have_reads = false; j->write = true; j->hold = false; j->all_mode = 0;
            if (k % 17 == 5) {
                // the "-a before the first data column" pattern: measure without writing, then submit again
                j->write = false; j->hold = true;
                pipe.submit(j);
                assert(pipe.wait(j) == 0);
                assert(j->info.n_data_cols == (uint64_t)(k % 3));
                if (j->info.n_data_cols) { j->all_mode = 1; j->write = true; j->hold = false; pipe.submit(j); want += "win " + std::to_string(k) + " all=1\n"; }
                else pipe.release(j);
            } else {
                pipe.submit(j);
                if (k % 23 == 0) assert(pipe.wait(j) == 0);
                want += "win " + std::to_string(k) + " all=0\n";
            }
        }
        assert(pipe.drain() == 0);
        // an error in the device stage: reported, later jobs are not written
        WinJob *j = pipe.acquire(); j->tid = 777; j->cb = 1000; j->write = true; j->hold = false; pipe.submit(j);
        WinJob *k = pipe.acquire(); k->tid = 0; k->cb = 1001; k->write = true; k->hold = false; pipe.submit(k);
        assert(pipe.drain() < 0);
        assert(pipe.error() < 0);
    }
    {
        // the text ring (one device thread): three pieces of 16 bytes, so that a window's text is several pieces, the device stage waits for
        // the writer, and the writer writes pieces of a job that is still on the device; held and unwritten jobs mixed in
        TextRing ring; ring.piece = 16;
        std::vector<std::vector<char>> store(3, std::vector<char>(16));
        for (auto &b : store) ring.buf.push_back(b.data());
        WinPipe *pp = nullptr;
        WinPipe pipe(4, [&pp](WinJob &j, int) {
            usleep((useconds_t)((j.cb * 53) % 400));
            std::string t;
            for (int r = 0; r < 1 + (int)(j.cb % 5); ++r) t += "ring " + std::to_string(j.cb) + " row " + std::to_string(r) + " all=" + std::to_string(j.all_mode) + "\n";
            j.info.n_data_cols = (uint64_t)(j.cb % 3); j.out_bytes = 0;
            if (!j.write) return 0;
            // (fetch_text with a fake engine: the same acquire / push protocol)
            for (size_t off = 0; off < t.size(); off += 16) {
                const size_t n = t.size() - off < 16 ? t.size() - off : 16;
                int idx = -1; char *b = pp->ring_acquire(&idx);
                if (!b) return -1;
                memcpy(b, t.data() + off, n);
                pp->ring_push(&j, idx, n);
            }
            j.ringed = true; j.out_bytes = t.size();
            return 0;
        }, out, "write error\n", 1);
        pp = &pipe;
        pipe.use_ring(&ring);
        assert(pipe.ring_on());
        for (int k = 0; k < 150; ++k) {
            WinJob *j = pipe.acquire();
            j->tid = 0; j->cb = k; j->ce = k + 1; j->have_reads = false; j->write = true; j->hold = false; j->all_mode = 0;
            auto text_of = [&](int all) { std::string t; for (int r = 0; r < 1 + k % 5; ++r) t += "ring " + std::to_string(k) + " row " + std::to_string(r) + " all=" + std::to_string(all) + "\n"; return t; };
            if (k % 13 == 4) {
                j->write = false; j->hold = true;
                pipe.submit(j);
                assert(pipe.wait(j) == 0);
                if (j->info.n_data_cols) { j->all_mode = 1; j->write = true; j->hold = false; pipe.submit(j); want += text_of(1); }
                else pipe.release(j);
            } else { pipe.submit(j); want += text_of(0); }
        }
        assert(pipe.drain() == 0);
    }
    fflush(out);
    rewind(out);
    std::string got; char buf[4096]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, out)) > 0) got.append(buf, n);
    fclose(out); unlink(path);
    if (got != want) { fprintf(stderr, "pipeline output differs: %zu vs %zu bytes\n", got.size(), want.size()); return 1; }
    printf("pipe_test OK (%zu bytes in order)\n", got.size());
    return 0;
}
