#!/bin/bash
# Round 6, GPU session AP: where consensus30 / consensus30_simple / calmd30 spend their step (rows f3 / f4: not looked at since round 2-3).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ap; mkdir -p $O
for wl in consensus30 consensus30_simple calmd30; do
  python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['config']['window_cols_per_gpu'], round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:10]})"
done 2>&1 | tee $O/bench.log
