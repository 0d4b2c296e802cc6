#!/bin/bash
# Round 6, GPU session A (first device run of the round): the whole -m gpu suite on the build with the host-kept name hashes
# (host_names.h), the driver's form of the default bench run, and the e2e commands with the process timeline (STA_DRIVER_TIMING=2:
# where the 0.28 s outside the window pipeline go).  Budget: ~12 GPU-minutes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
TAG=r06a TESTS=1 FULL="" WL="mpileup30_B depth30 mpileup30_EA_pairs" STATS="" bash scripts/gpu_record.sh
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 900 $O/bench_default.json; tail -3 $O/bench_default.err
STA_E2E_TIMING=2 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_timeline.log 2>&1; cat $O/e2e_timeline.log | cut -c1-400
true
