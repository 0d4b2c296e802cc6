// fasta_threads.cpp -- four threads fetch every contig of an indexed FASTA at once, two of them from the last contig to the first,
// twenty times over: run under ThreadSanitizer by tests/test_fasta_index.py (the loader's read-ahead thread is retired and replaced
// by whichever fetch gets there first).  Test infrastructure.
#include "../../samtools_amd/csrc/host_io.h"
#include <cstdio>
#include <thread>
#include <vector>
using namespace sta;
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    for (int rep = 0; rep < 20; ++rep) {
        auto fa = Fasta::load(argv[1]);
        if (!fa || !fa->lazy()) { printf("not index-driven\n"); return 1; }
        std::vector<std::thread> th; std::vector<size_t> tot(4, 0);
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                const auto &nm = fa->names();
                for (size_t k = 0; k < nm.size(); ++k) { const size_t i = (t & 1) ? nm.size() - 1 - k : k; const std::string *s = fa->fetch(nm[i]); tot[(size_t)t] += s ? s->size() : 0; }
            });
        for (auto &x : th) x.join();
        if (tot[0] != tot[1] || tot[0] != tot[2] || tot[0] != tot[3] || !tot[0]) { printf("MISMATCH\n"); return 1; }
    }
    printf("ok\n");
    return 0;
}
