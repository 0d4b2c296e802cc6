#!/bin/bash
# A/B of two builds of the library inside ONE box (box-to-box spread is +-3 %): samtools_amd/lib/libprev.so vs libnew.so, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in prev new; do
  cp samtools_amd/lib/lib$v.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload ${WL:-mpileup30} --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]})"
done; done
cp samtools_amd/lib/libnew.so samtools_amd/lib/libsamtools_amd.so
