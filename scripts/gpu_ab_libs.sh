#!/bin/bash
# A/B/C of several builds of the library inside ONE box: samtools_amd/lib/lib<name>.so for name in $LIBS, alternating $REPS times
cd $GRAFT_REPO_ROOT
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
for rep in $(seq 1 ${REPS:-2}); do for v in $LIBS; do
  cp samtools_amd/lib/lib$v.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps ${STEPS:-20} --warmup 3 --workload ${WL:-mpileup30_B} --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]})"
done; done
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
