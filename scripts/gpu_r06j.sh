#!/bin/bash
# Round 6, GPU session J: where a run's last 0.1-0.2 s go: timeline marks up to the fast exit (STA_FAST_EXIT=1 with STA_DRIVER_TIMING=2) against the
# parent's wall clock, for the three e2e commands on the half-Gbase input.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
STA_E2E_TIMING=2 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_exit.log 2>&1; grep -E "plain|wall =|timeline" $O/e2e_exit.log | cut -c1-120
true
