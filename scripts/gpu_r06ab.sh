#!/bin/bash
# Round 6, GPU session AB: mpileup300 on a 2 M-column window (four 512 K pieces; was one 512 K window: 8.0 ms = 19 580 Mbases/s in session S),
# and mpileup30 again after the piece size became the workload's own test size.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
for wl in mpileup300 mpileup30; do
  ( time python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json ) 2>&1 | grep real
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['config']['window_cols_per_gpu'], round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:7]})"
done 2>&1 | tee $O/bench.log
python bench.py --steps 10 --warmup 3 --workload mpileup300 --cols 524288 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpileup300 at 512K', round(d['ms_per_step'],3), round(d['value']))" | tee -a $O/bench.log
