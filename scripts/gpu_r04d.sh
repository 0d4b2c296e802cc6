#!/bin/bash
# round 4, GPU session D: SQ counters of the class-S kernel, with and without its row stream
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
bash scripts/gpu_r04c.sh
bash scripts/gpu_sq.sh mpileup30 r04d/sq_mode0 > $O/sq_mode0.log 2>&1; grep "k_baq7s" $O/sq_mode0.log
STA_BAQ7S_MODE=2 bash scripts/gpu_sq.sh mpileup30 r04d/sq_mode2 > $O/sq_mode2.log 2>&1; grep "k_baq7s" $O/sq_mode2.log
