// driver_globalopts.h -- the reference's "global" long options for the commands of this path.
//
// Every samtools subcommand splices SAM_OPT_GLOBAL_OPTIONS(...) into its getopt table (sam_opts.h:63-71) and hands what getopt returns
// for them to parse_sam_global_opt (sam_opts.c:81-171).  mpileup enables --input-fmt-option, --reference, --write-index and --verbosity
// (bam_plcmd.c:1098: SAM_OPT_GLOBAL_OPTIONS('-', 0, '-', '-', 0, '-')), depth those and --threads / -@ (bam2depth.c:765); the disabled
// ones stay in the table with the value '?', so that naming them prints the usage.  A script that runs under samtools must run here:
// the options are parsed and checked as the reference checks them; most have nothing to set in this engine's own BAM / SAM reader
// (they steer HTSlib's CRAM codec, which is out of scope -- SURVEY.md section 2b).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <climits>
#include <string>
#include <strings.h>
#include <getopt.h>

namespace sta {

enum { GOPT_INPUT_FMT = 0x100, GOPT_INPUT_FMT_OPTION, GOPT_OUTPUT_FMT, GOPT_OUTPUT_FMT_OPTION, GOPT_REFERENCE, GOPT_NTHREADS, GOPT_WRITE_INDEX, GOPT_VERBOSITY };

// SAM_OPT_GLOBAL_OPTIONS(o1 .. o6): 0 = long option only, '-' = disabled (the entry stays, with the value '?'), c = short option c
#define STA_GOPT_VAL(val, defval) ((val) == '-' ? '?' : (val) ? (val) : (defval))
#define STA_GLOBAL_OPTIONS(o1, o2, o3, o4, o5, o6) \
    { "input-fmt", required_argument, NULL, STA_GOPT_VAL(o1, sta::GOPT_INPUT_FMT) }, \
    { "input-fmt-option", required_argument, NULL, STA_GOPT_VAL(o2, sta::GOPT_INPUT_FMT_OPTION) }, \
    { "output-fmt", required_argument, NULL, STA_GOPT_VAL(o3, sta::GOPT_OUTPUT_FMT) }, \
    { "output-fmt-option", required_argument, NULL, STA_GOPT_VAL(o4, sta::GOPT_OUTPUT_FMT_OPTION) }, \
    { "reference", required_argument, NULL, STA_GOPT_VAL(o5, sta::GOPT_REFERENCE) }, \
    { "threads", required_argument, NULL, STA_GOPT_VAL(o6, sta::GOPT_NTHREADS) }, \
    { "write-index", no_argument, NULL, sta::GOPT_WRITE_INDEX }, \
    { "verbosity", required_argument, NULL, sta::GOPT_VERBOSITY }

struct GlobalArgs {                 // sam_global_args (sam_opts.h:34-41), as far as this engine has a use for it
    std::string reference;          // --reference FILE
    int nthreads = 0, write_index = 0;
    int verbosity = 3;              // hts_verbose (HTS_LOG_WARNING)
};

inline int &global_verbosity() { static int v = 3; return v; }      // what --verbosity set last (the drivers' own [W::] lines ask it)

// parse_int_value (sam_opts.c:43-59): the whole string must be an int
inline bool gopt_parse_int(const char *s, int *out)
{
    char *end = nullptr;
    errno = 0;
    const long v = strtol(s, &end, 10);
    if (end == s || *end != '\0' || errno == ERANGE || v < INT_MIN || v > INT_MAX) return false;
    *out = (int)v;
    return true;
}

// hts_opt_add (HTSlib hts.c, absent from the reference tree: the key names as its manual page lists them, samtools.1 "GLOBAL COMMAND
// OPTIONS" / htslib's hts_opt_add table): KEY or KEY=VALUE, every key in lower or upper case; an unknown key is an error.  All of them
// steer HTSlib's own readers and writers (CRAM above all); `filter` would make sam_read1 drop records by an expression this engine does
// not evaluate, so it is refused rather than ignored.
inline int gopt_check_format_option(const char *arg)
{
    static const char *const keys[] = {
        "decode_md", "verbosity", "seqs_per_slice", "bases_per_slice", "slices_per_container", "embed_ref", "no_ref", "pos_delta", "ignore_md5",
        "use_bzip2", "use_rans", "use_lzma", "use_tok", "use_fqz", "use_arith", "fast", "normal", "small", "archive", "reference", "version",
        "multi_seq_per_slice", "nthreads", "cache_size", "required_fields", "lossy_names", "name_prefix", "store_md", "store_nm", "block_size",
        "level", "filter", "fastq_aux", "fastq_barcode", "fastq_rnum", "fastq_casava", "fastq_name2" };
    std::string key(arg);
    const size_t eq = key.find('=');
    if (eq != std::string::npos) key.resize(eq);
    bool lower = true, upper = true;
    for (char c : key) { if (c >= 'a' && c <= 'z') upper = false; if (c >= 'A' && c <= 'Z') lower = false; }
    if (lower || upper)
        for (const char *k : keys)
            if (!strcasecmp(k, key.c_str())) {
                if (!strcasecmp(k, "filter")) { fprintf(stderr, "[E::hts_opt_add] the MI355X engine's reader evaluates no filter expression: '%s' is not supported\n", arg); return -1; }
                return 0;
            }
    fprintf(stderr, "[E::hts_opt_add] Unknown option '%s'\n", key.c_str());
    return -1;
}

// parse_sam_global_opt (sam_opts.c:81-171): c = what getopt_long returned; looked up in lopts to find the long name it stands for
inline int parse_global_opt(int c, const char *optarg, const struct option *lopt, GlobalArgs *ga)
{
    for (; lopt->name; ++lopt) {
        if (c != lopt->val) continue;
        if (!strcmp(lopt->name, "input-fmt") || !strcmp(lopt->name, "output-fmt")) return 0;       // (never enabled on this path)
        if (!strcmp(lopt->name, "input-fmt-option") || !strcmp(lopt->name, "output-fmt-option")) return gopt_check_format_option(optarg);
        if (!strcmp(lopt->name, "reference")) { ga->reference = optarg; return 0; }
        if (!strcmp(lopt->name, "threads")) {
            if (!gopt_parse_int(optarg, &ga->nthreads)) { fprintf(stderr, "Invalid threads value.\n"); return -1; }
            return 0;
        }
        if (!strcmp(lopt->name, "write-index")) { ga->write_index = 1; return 0; }
        if (!strcmp(lopt->name, "verbosity")) {
            if (!gopt_parse_int(optarg, &ga->verbosity)) { fprintf(stderr, "Invalid verbosity value.\n"); return -1; }
            global_verbosity() = ga->verbosity;
            return 0;
        }
    }
    fprintf(stderr, "Unexpected global option.\n");
    return -1;
}

}  // namespace sta
