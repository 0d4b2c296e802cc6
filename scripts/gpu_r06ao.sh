#!/bin/bash
# Round 6, GPU session AO: the other piecewise workloads on their larger windows (mpileup300_B 2 M, mpileup100[_B] 4 M, _indel / _trim 16 M columns).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ao; mkdir -p $O
for wl in mpileup300_B mpileup100 mpileup100_B mpileup30_indel mpileup30_trim; do
  python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['config']['window_cols_per_gpu'], round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:5]})"
done 2>&1 | tee $O/bench.log
