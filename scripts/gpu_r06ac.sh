#!/bin/bash
# Round 6, GPU session AC: the asymptote of the window size: mpileup30 at 32 M and 64 M columns per step (16 M: 20 834-21 212 Mbases/s).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
for cols in 33554432 67108864; do
  ( time python bench.py --steps 6 --warmup 2 --workload mpileup30 --cols $cols --no-cpu-baseline --no-pmc --no-e2e 2>$O/err_$cols.log | tail -1 > $O/bench_$cols.json ) 2>&1 | grep real
  python -c "import sys,json; d=json.loads(open('$O/bench_$cols.json').read()); print('cols $cols', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})" || tail -3 $O/err_$cols.log
done 2>&1 | tee $O/bench.log
