#!/bin/bash
# Round 6, GPU session H: the drivers with the text ring (6 x 8 MiB page-locked pieces instead of one page-locked text buffer per slot), plain-memory
# staging pools for inputs below 1 GiB (STA_PIN), eight pipeline slots and a producer that no longer waits for the runtime: whole -m gpu suite,
# then the e2e commands (plain runs = what bench.py's e2e times; traces with STA_DRIVER_TIMING=3), and STA_PIN=1 for comparison.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -o timeout=240 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
STA_E2E_TIMING=3 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_trace.log 2>&1; grep -E "plain|driver timing" $O/e2e_trace.log | cut -c1-420
STA_PIN=1 STA_E2E_TIMING=1 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_pin1.log 2>&1; grep -E "plain" $O/e2e_pin1.log | cut -c1-300
true
