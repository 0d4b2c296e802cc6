#!/bin/bash
# Round 6, GPU session AD: depth30 at 8 M (default), 32 M and 64 M columns per step: how much of its 0.32 ms is per window.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
for cols in 8388608 33554432 67108864; do
  python bench.py --steps 20 --warmup 3 --workload depth30 --cols $cols --no-cpu-baseline --no-pmc --no-e2e 2>$O/err_$cols.log | tail -1 > $O/bench_$cols.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$cols.json').read()); print('depth30 cols $cols', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})" || tail -3 $O/err_$cols.log
done 2>&1 | tee $O/bench.log
for cols in 16777216; do
  python bench.py --steps 20 --warmup 3 --workload mpileup30_B --cols $cols --no-cpu-baseline --no-pmc --no-e2e 2>$O/err_B$cols.log | tail -1 > $O/bench_B$cols.json
  python -c "import sys,json; d=json.loads(open('$O/bench_B$cols.json').read()); print('mpileup30_B cols $cols', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})" || tail -3 $O/err_B$cols.log
done 2>&1 | tee -a $O/bench.log
