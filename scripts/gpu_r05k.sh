#!/bin/bash
# Round 5, GPU session K: the round's final record of HEAD -- whole -m gpu suite, smoke, default bench run (the driver's form), workloads, kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
TAG=r05k TESTS=1 FULL="" WL="mpileup30_B depth30 mpileup300 mpileup30_indel mpileup30_trim mpileup30_EA_pairs mpileup30_B_sOx" STATS="mpileup30 mpileup30_B mpileup30_B_sOx" bash scripts/gpu_record.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
true
