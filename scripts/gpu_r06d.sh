#!/bin/bash
# Round 6, GPU session D: counters of k_mplp_emit_deep<true> (3.07 ms at three waves per SIMD, 4.13 at two: latency, not issue?) -- SQ wait /
# instruction counters and FETCH / WRITE_SIZE of mpileup30_B_sOx.  ~4 GPU-minutes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
bash scripts/gpu_sq.sh mpileup30_B_sOx r06d/sq > $O/sq.log 2>&1; tail -6 $O/sq.log | cut -c1-900
TAG=r06d bash scripts/gpu_pmc.sh mpileup30_B_sOx > $O/pmc.log 2>&1; tail -6 $O/pmc.log | cut -c1-300
true
