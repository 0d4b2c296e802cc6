#!/bin/bash
# A/B of an environment knob inside ONE box: $A vs $B (e.g. A="STA_TILE_PREFETCH=0" B="STA_TILE_PREFETCH=1"), alternating three times
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in "$A" "$B"; do
  env $v python bench.py --steps ${STEPS:-20} --warmup 3 --workload ${WL:-mpileup30_B} --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]}, (d.get('parity_check') or {}).get('identical'))"
done; done
