#!/bin/bash
# one workload under a list of environment settings (';' separated in $SETS), one box
cd $GRAFT_REPO_ROOT
IFS=';' read -ra S <<< "$SETS"
for e in "${S[@]}"; do
  env $e python bench.py --steps ${STEPS:-5} --warmup 2 --workload ${WL:-mpileup30_B_sOx} --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
done
