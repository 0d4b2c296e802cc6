#!/bin/bash
# Round 5, GPU session D: suite on the dense class-S list + the threshold table in every BAQ kernel; A/B against round 4's library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
( time timeout 1200 python -m pytest tests -m gpu -q -x -o timeout=600 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
run() { # lib workload
  cp samtools_amd/lib/lib$1.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:5]})"
}
for rep in 1 2; do for wl in mpileup30 mpileup30_indel mpileup30_trim mpileup300; do for v in prev new; do run $v $wl; done; done; done 2>&1 | tee $O/ab.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
true
