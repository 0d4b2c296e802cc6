/*
 * oracle/o_main.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * Dispatch mirroring bamtk.c:248,270 for the two subcommands on the hot path.
 */
#include <stdio.h>
#include <string.h>

int o_main_mpileup(int argc, char *argv[]);
int o_main_depth(int argc, char *argv[]);
int o_main_plpdump(int argc, char *argv[]);
int o_main_bedcov(int argc, char *argv[]);
int o_main_coverage(int argc, char *argv[]);
int o_main_stats(int argc, char *argv[]);
int o_main_glf(int argc, char *argv[]);
int o_main_calmd(int argc, char *argv[]);
int o_main_consensus(int argc, char *argv[]);

int main(int argc, char *argv[])
{
    if (argc < 2) { fprintf(stderr, "usage: oracle_samtools <mpileup|depth|plpdump|bedcov|coverage|stats|glf|calmd|consensus> [options]\n"); return 1; }
    if (strcmp(argv[1], "mpileup") == 0) return o_main_mpileup(argc - 1, argv + 1);
    if (strcmp(argv[1], "depth") == 0) return o_main_depth(argc - 1, argv + 1);
    if (strcmp(argv[1], "plpdump") == 0) return o_main_plpdump(argc - 1, argv + 1);
    if (strcmp(argv[1], "bedcov") == 0) return o_main_bedcov(argc - 1, argv + 1);
    if (strcmp(argv[1], "coverage") == 0) return o_main_coverage(argc - 1, argv + 1);
    if (strcmp(argv[1], "stats") == 0) return o_main_stats(argc - 1, argv + 1);
    if (strcmp(argv[1], "calmd") == 0) return o_main_calmd(argc - 1, argv + 1);
    if (strcmp(argv[1], "consensus") == 0) return o_main_consensus(argc - 1, argv + 1);
    if (strcmp(argv[1], "glf") == 0) return o_main_glf(argc - 1, argv + 1);
    fprintf(stderr, "[main] unrecognized command '%s'\n", argv[1]);
    return 1;
}
