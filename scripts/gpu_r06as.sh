#!/bin/bash
# Round 6, GPU session AS: after k_baq_list's two instantiations: the whole -m gpu suite, the indel-rich workload at both window sizes, the default run.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06as; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -o timeout=600 -n 4 ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
for cols in 4194304 16777216; do
  python bench.py --steps 10 --warmup 3 --workload mpileup30_indel --cols $cols --no-cpu-baseline --no-pmc --no-e2e --verify 2>/dev/null | tail -1 > $O/bench_indel_$cols.json
  python -c "import sys,json; d=json.loads(open('$O/bench_indel_$cols.json').read()); print('mpileup30_indel $cols', round(d['ms_per_step'],3), round(d['value']), (d.get('verify') or {}).get('identical'), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
done 2>&1 | tee $O/bench.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print('default', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['parity_check']['identical'], d['e2e']['identical_to_oracle']['all'])" | tee -a $O/bench.log
