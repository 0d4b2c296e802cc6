// cons_host.h -- host side of the consensus path: lookup tables (libm), the per-window result the command driver formats,
// and the driver entry shared by the product (device compute) and the CPU harness of tests/cpu (emulated compute).
#pragma once
#include "../../include/samtools_amd.h"
#include "cons_core.h"
#include <functional>
#include <string>
#include <vector>

namespace sta {

// consensus_init() x2 + the static tables of bam_consensus_tab.h (bam_consensus.c:740-883, 3424-3445)
void cons_build_tables(const sta_cons_params &p, cons::Tables &t);
cons::Par cons_par(const sta_cons_params &p);

struct ConsWindowOut {
    sta_cons_info info;
    std::vector<int32_t> ins;
    std::vector<sta_cons_col> cols;
    std::vector<uint64_t> col_off;
    std::vector<char> seq, qual;
};
// stages and computes one window (file 0 of w); <0 on error with the message in err
typedef std::function<int(const sta_window &w, const sta_cons_params &p, ConsWindowOut &out, std::string &err)> ConsCompute;

// `consensus` command line (bam_consensus.c:3082-3593): option parsing, windows, FASTA / FASTQ / pileup writers
int consensus_cli(int argc, char **argv, const ConsCompute &compute);

}  // namespace sta
