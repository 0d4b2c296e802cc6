#!/bin/bash
# Round 6, GPU session Z: does a larger resident window pay?  The class-S kernel's tail (13 108 groups on 2 048 persistent waves = 6.4 rounds) and
# the fixed parts of a step (launch gaps, the plan's round trip) are per window: mpileup30 at 4 M (default), 8 M, 16 M columns per step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
for cols in 4194304 8388608 16777216; do
  python bench.py --steps 8 --warmup 2 --workload mpileup30 --cols $cols --no-cpu-baseline --no-pmc --no-e2e 2>$O/err_$cols.log | tail -1 > $O/bench_$cols.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$cols.json').read()); print('cols $cols', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})" || tail -3 $O/err_$cols.log
done 2>&1 | tee $O/bench.log
