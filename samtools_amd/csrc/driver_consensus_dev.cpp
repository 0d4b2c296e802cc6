// driver_consensus_dev.cpp -- `samtools-amd consensus`: binds the command driver (driver_consensus.cpp) to the device engine.
#include "cons_host.h"
#include <cstdio>
#include <cstdlib>

// the product's window compute: stage, run on the device, fetch.  No device, no consensus.
extern "C" int sta_main_consensus(int argc, char **argv)
{
    if (sta_device_count() < 1) { fprintf(stderr, "samtools consensus: no usable HIP device (the MI355X engine has no CPU fallback)\n"); return 2; }
    sta_engine *eng = nullptr;
    int dev = 0;
    if (const char *e = getenv("STA_DEVICE")) dev = atoi(e);
    if (sta_engine_create(&eng, dev, nullptr) != STA_OK) { fprintf(stderr, "samtools consensus: no usable HIP device\n"); return 2; }
    sta::ConsCompute compute = [eng](const sta_window &w, const sta_cons_params &p, sta::ConsWindowOut &out, std::string &err) -> int {
        if (sta_stage_window(eng, &w) != STA_OK || sta_consensus_run(eng, &p, &out.info) != STA_OK) { err = sta_last_error(eng); return -1; }
        out.ins.resize((size_t)(w.col_end - w.col_beg));
        out.cols.resize((size_t)out.info.n_cols);
        if (p.want_pileup) { out.col_off.resize((size_t)out.info.n_cols + 1); out.seq.resize((size_t)out.info.n_entries + 1); out.qual.resize((size_t)out.info.n_entries + 1); }
        const int rc = sta_fetch_consensus(eng, out.ins.data(), out.cols.data(), p.want_pileup ? out.col_off.data() : nullptr,
                                           p.want_pileup ? out.seq.data() : nullptr, p.want_pileup ? out.qual.data() : nullptr);
        if (rc != STA_OK) { err = sta_last_error(eng); return -1; }
        return 0;
    };
    const int ret = sta::consensus_cli(argc, argv, compute);
    sta_engine_destroy(eng);
    return ret;
}
