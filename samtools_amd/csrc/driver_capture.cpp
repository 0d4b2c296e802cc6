// driver_capture.cpp -- the command drivers with their text handed back in memory instead of written to a file.
//
// A sharded run (samtools_amd/shard.py) gathers every rank's block of text on rank 0; the drivers' writer thread
// appends to a FILE*, so the capture form gives it a memory stream and returns the buffer: no temporary file between the
// driver and the gather.  The reference has no counterpart (its column loop prints as it goes, bam_plcmd.c:663-868).
#include "driver_shard.h"
#include "../../include/samtools_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace sta {
static thread_local FILE *t_capture = nullptr;
FILE *driver_default_out() { return t_capture ? t_capture : stdout; }
bool driver_out_is_borrowed(FILE *f) { return f == stdout || (t_capture && f == t_capture); }
}  // namespace sta

extern "C" int sta_main_capture(int argc, char **argv, char **text, uint64_t *n_bytes)
{
    if (!text || !n_bytes || argc < 1 || !argv || !argv[0]) return STA_ERR_ARG;
    *text = nullptr; *n_bytes = 0;
    const bool mp = !strcmp(argv[0], "mpileup"), dp = !strcmp(argv[0], "depth");
    if (!mp && !dp) return STA_ERR_ARG;
    char *buf = nullptr; size_t len = 0;
    FILE *ms = open_memstream(&buf, &len);
    if (!ms) return STA_ERR_IO;
    sta::t_capture = ms;
    const int rc = mp ? sta_main_mpileup(argc, argv) : sta_main_depth(argc, argv);
    sta::t_capture = nullptr;
    if (fclose(ms) != 0) { free(buf); return STA_ERR_IO; }
    *text = buf; *n_bytes = (uint64_t)len;
    return rc;
}

extern "C" void sta_capture_free(char *text) { free(text); }
