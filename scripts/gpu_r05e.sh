#!/bin/bash
# Round 5, GPU session E: A/B of the dense class-S list against the session-C library (one box), parity of the changed kernels, the e2e breakdown
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
( time timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py tests/test_gpu_synth.py -q -x -o timeout=600 -p no:cacheprovider -k "mpileup30_trim or mpileup30_indel or mpileup30] or env_only or baq or EA_pairs or 3files" ) > $O/pytest_sel.log 2>&1; tail -4 $O/pytest_sel.log
run() { # lib workload
  cp samtools_amd/lib/lib$1.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:7]})"
}
for rep in 1 2 3; do for wl in mpileup30 mpileup30_indel mpileup30_trim; do for v in sesC new; do run $v $wl; done; done; done 2>&1 | tee $O/ab.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
timeout 600 python scripts/e2e_bench_shape.py 2>&1 | tee $O/e2e_breakdown.log
true
