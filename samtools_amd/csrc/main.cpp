// main.cpp -- `samtools-amd` command: the two sub-commands of the hot path, dispatched the way
// bamtk.c:248,270 dispatches them in the reference.
#include "../../include/samtools_amd.h"
extern "C" int sta_main_plpdump(int argc, char **argv);   // diagnostic client of the bam_plp_* surface (driver_plpdump.cpp)
extern "C" int sta_main_bedcov(int argc, char **argv);    // bedcov.c column loop on the engine iterator (driver_bedcov.cpp)
extern "C" int sta_main_coverage(int argc, char **argv);  // coverage.c tabular loop on the engine iterator (driver_coverage.cpp)
#include <cstdio>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "Usage: samtools-amd <mpileup|depth|consensus|bedcov|coverage|stats|plpdump|glf|calmd> [options]\n%s\n", sta_version());
        return 1;
    }
    sta_exit_after_main(1);      // (the drivers of the hot path end the process once their text is out: no teardown of page-locked pools / the runtime)
    if (strcmp(argv[1], "mpileup") == 0) return sta_main_mpileup(argc - 1, argv + 1);
    if (strcmp(argv[1], "depth") == 0) return sta_main_depth(argc - 1, argv + 1);
    if (strcmp(argv[1], "consensus") == 0) return sta_main_consensus(argc - 1, argv + 1);
    if (strcmp(argv[1], "calmd") == 0) return sta_main_calmd(argc - 1, argv + 1);
    if (strcmp(argv[1], "glf") == 0) return sta_main_glf(argc - 1, argv + 1);
    if (strcmp(argv[1], "plpdump") == 0) return sta_main_plpdump(argc - 1, argv + 1);
    if (strcmp(argv[1], "bedcov") == 0) return sta_main_bedcov(argc - 1, argv + 1);
    if (strcmp(argv[1], "coverage") == 0) return sta_main_coverage(argc - 1, argv + 1);
    if (strcmp(argv[1], "stats") == 0) return sta_main_stats(argc - 1, argv + 1);
    fprintf(stderr, "[main] unrecognized command '%s'\n", argv[1]);
    return 1;
}
