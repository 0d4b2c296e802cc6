#!/bin/bash
# Round 6, GPU session AJ: the indel-rich and trimmed inputs on 16 M-column windows (4 M: 7.19 / 6.37 ms = 17 506 / 18 092 Mbases/s in the record).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aj; mkdir -p $O
for wl in mpileup30_indel mpileup30_trim; do
  python bench.py --steps 8 --warmup 2 --workload $wl --cols 16777216 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl 16M', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:7]})"
done 2>&1 | tee $O/bench.log
