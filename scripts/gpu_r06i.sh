#!/bin/bash
# Round 6, GPU session I: a 3.8-Gbase input (the bench sample on 64 contigs, BAM level 1, ~1.3 GB): depth -a / mpileup -B -f / mpileup -f with
# page-locked staging (the default for inputs of 1 GiB and more) and without (STA_PIN=0).  VERDICT r05 item 3 asks >= 2 500 Mbases/s net for depth -a.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
df -h /tmp | tail -1
STA_E2E_BIG=1 STA_E2E_TIMING=1 timeout 1500 python scripts/e2e_bench_shape.py 2000000 ${COPIES:-64} > $O/e2e_big.log 2>&1; grep -E "input|plain" $O/e2e_big.log | cut -c1-300
true
