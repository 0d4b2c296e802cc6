#!/bin/bash
# Round 6, GPU session AF: the round's record after the overlap, glf and window-size changes: the whole -m gpu suite, smoke(), the default
# bench run (as the driver runs it), the other workloads' bench lines, rocprofv3 kernel stats of the headline, the 3.8-Gbase file lane.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06af; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -o timeout=600 -n 4 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print('default', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['parity_check']['identical'], d['e2e']['identical_to_oracle']['all'], {k: round(v['wall_s'],3) for k, v in d['e2e']['commands'].items()})"
for wl in mpileup30_B mpileup300 mpileup300_B mpileup100 mpileup30_EA_pairs mpileup30_B_pairs mpileup30_indel mpileup30_trim mpileup30_B_s mpileup30_B_sOx depth30 glf30; do
  python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['config']['window_cols_per_gpu'], round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:5]})"
done 2>&1 | tee $O/bench_others.log
R=$GRAFT_REPO_ROOT; ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-e2e > $R/$O/prof.log 2>&1 ); head -6 $O/prof/p_kernel_stats.csv | cut -c1-60,150-260
STA_E2E_BIG=1 STA_E2E_TIMING=1 timeout 900 python scripts/e2e_bench_shape.py 2000000 64 > $O/e2e_big.log 2>&1; grep -E "input|plain" $O/e2e_big.log | cut -c1-200
true
