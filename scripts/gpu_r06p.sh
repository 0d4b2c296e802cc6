#!/bin/bash
# Round 6, GPU session P: mpileup300's step is 9.46 ms with 8.3 ms of kernels (round 5: 8.49): which change costs a millisecond outside the kernels?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
for v in 1 0; do
  STA_BAQ_LIST_SORT=$v python bench.py --steps 10 --warmup 3 --workload mpileup300 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpileup300 sort=$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
done 2>&1 | tee $O/m300.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-e2e --workload mpileup300 > $R/$O/prof.log 2>&1
head -14 $R/$O/prof/*kernel_stats.csv | cut -c1-150; head -8 $R/$O/prof/*memory_copy_stats.csv 2>/dev/null | cut -c1-150
true
