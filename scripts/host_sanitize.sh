# Builds the host input plumbing (BGZF worker pool, parse-ahead reader, record lane, chunk lane, stager, sta_io_scan) with
# ThreadSanitizer and with AddressSanitizer + UBSan and runs both lanes over a few inputs.  No device needed.
# usage: bash scripts/host_sanitize.sh file1.bam [file2.sam ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=${TMPDIR:-/tmp}/sta_san; mkdir -p $T
cat > $T/scan_main.cpp <<EOT
#include "$R/include/samtools_amd.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
    uint64_t n = 0, h = 0;
    int rc = sta_io_scan(argv[1], atoi(argv[2]), atoi(argv[3]), &n, &h);
    printf("rc=%d n=%llu h=%016llx\n", rc, (unsigned long long)n, (unsigned long long)h);
    return rc ? 1 : 0;
}
EOT
S=$R/samtools_amd/csrc
for san in thread address,undefined; do
  g++ -std=c++17 -g -O1 -fsanitize=$san -fno-omit-frame-pointer -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Wno-deprecated-declarations \
      $S/host_bgzf.cpp $S/host_inflate.cpp $S/host_bamout.cpp $S/host_io.cpp $S/host_chunk.cpp $S/host_pump.cpp $S/host_names.cpp $S/host_stage.cpp $S/host_scan.cpp $S/host_pinned.cpp $S/host_mods.cpp $R/tests/cpu/gpu_inflate_stub.cpp $T/scan_main.cpp \
      -o $T/scan_${san%%,*} -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lz -pthread
done
export STA_NO_PINNED=1 ASAN_OPTIONS=detect_leaks=0
for f in "$@"; do
  for env in "STA_WINDOW_COLS=700" "STA_WINDOW_COLS=64 STA_WINDOW_READS=7 STA_SCAN_DROP=1"; do
    for st in 0 1 2; do
      for b in thread address; do
        out=$(env $env $T/scan_$b "$f" 4 $st 2>&1 | grep -E "rc=|SUMMARY|runtime error" | sort | uniq -c | tr '\n' ' ')
        echo "$b stage=$st [$env] $(basename "$f"): $out"
      done
    done
  done
done
