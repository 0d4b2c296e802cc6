#!/bin/bash
# Round 6, GPU session G: per-window trace (STA_DRIVER_TIMING=3) of the three e2e commands on the half-Gbase input.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
STA_E2E_TIMING=3 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_trace.log 2>&1; grep -E "window|plain" $O/e2e_trace.log | cut -c1-220 | head -150
true
