#!/bin/bash
# Round 5, GPU session H: the generic walker's single-walk emit after the cursor fix -- its tests, then A/B on mpileup30_B_sOx
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py -q -o timeout=240 -p no:cacheprovider -k "generic_walker or mpileup_sO or output_extra or 79" ) > $O/pytest_generic.log 2>&1; tail -3 $O/pytest_generic.log
( time timeout 600 python -m pytest tests/test_gpu_benchsize_parity.py -q -o timeout=400 -p no:cacheprovider -k "sOx" ) > $O/pytest_sox.log 2>&1; tail -3 $O/pytest_sox.log
run() { # env-assignment label
  env $1 python bench.py --steps 10 --warmup 3 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
}
for rep in 1 2; do run STA_GENERIC_PASSES=1 passes; run STA_GENERIC_PASSES=0 1walk; run "STA_GENERIC_PASSES=0 STA_GENERIC_LDS_CAP=4096" 1walk_cap4k; run "STA_GENERIC_PASSES=0 STA_GENERIC_LDS_CAP=16384" 1walk_cap16k; done 2>&1 | tee $O/ab_generic.log
true
