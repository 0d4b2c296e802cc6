#!/bin/bash
# Round 6, GPU session AI: the 3.8-Gbase file lane, where the device thread (copies and kernels of a window one after the other on one stream)
# now holds the producer up ("waiting for a slot" 0.17-0.32 s of 1.1-1.3): two device threads (STA_DEV_THREADS=2), 4 M-column windows, both.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ai; mkdir -p $O
STA_E2E_BIG=1 STA_E2E_TIMING=1 STA_E2E_VARIANTS="-;STA_DEV_THREADS=2;STA_WINDOW_COLS=4194304;STA_WINDOW_COLS=4194304,STA_DEV_THREADS=2" timeout 1200 python scripts/e2e_bench_shape.py 2000000 64 > $O/e2e.log 2>&1; grep -E "input|plain" $O/e2e.log | cut -c1-220
