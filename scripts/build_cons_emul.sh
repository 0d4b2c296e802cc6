#!/bin/sh
# builds the CPU harness of the consensus path (tests/cpu/cons_emul.cpp: test infrastructure) into $1
set -e
cd "$(dirname "$0")/.."
S=samtools_amd/csrc
g++ -O1 -g -std=c++17 -ffp-contract=off -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpu/cons_emul.cpp $S/driver_consensus.cpp $S/cons_host.cpp \
    $S/host_io.cpp $S/host_bgzf.cpp $S/host_inflate.cpp $S/host_pump.cpp $S/host_names.cpp $S/host_chunk.cpp $S/host_stage.cpp $S/host_mods.cpp $S/host_pinned.cpp tests/cpu/gpu_inflate_stub.cpp \
    -o "$1" -pthread -lz -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
