# builds tests/cpu/stage_bench.cpp against the host sources (no device code) -> /tmp/stage_bench
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/samtools_amd/csrc
g++ -O2 -g -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$R/include $R/tests/cpu/stage_bench.cpp \
    $C/host_io.cpp $C/host_bgzf.cpp $C/host_inflate.cpp $C/host_chunk.cpp $C/host_stage.cpp $C/host_pump.cpp $C/host_names.cpp $C/host_pinned.cpp $C/host_mods.cpp $R/tests/cpu/gpu_inflate_stub.cpp \
    -o ${1:-/tmp/stage_bench} -pthread -lz -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
