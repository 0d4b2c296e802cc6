#!/bin/bash
# Round 5, GPU session Y: final record after the hunt4 fixes -- whole -m gpu suite, smoke, default bench run, the workloads the fixes touch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
TAG=r05y TESTS=1 FULL="" WL="mpileup30_EA_pairs mpileup30_B_sOx mpileup30_B" STATS="" bash scripts/gpu_record.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; tail -3 $O/bench_default.err
true
