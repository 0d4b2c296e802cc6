#!/bin/bash
# Round 5, GPU session G: the generic walker's single-walk emit (A/B against the per-column walks on mpileup30_B_sOx; kill criterion: less
# than 1.5x on k_mplp_emit), then the round's final record of HEAD: whole -m gpu suite, default bench run, workloads, kernel stats.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py -q -x -o timeout=240 -p no:cacheprovider -k "generic_walker or mpileup_sO or output_extra" ) > $O/pytest_generic.log 2>&1; tail -3 $O/pytest_generic.log
run() { # env-assignment label
  env $1 python bench.py --steps 10 --warmup 3 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
}
for rep in 1 2; do run STA_GENERIC_PASSES=1 passes; run STA_GENERIC_PASSES=0 1walk; done 2>&1 | tee $O/ab_generic.log
TAG=r05g TESTS=1 FULL="mpileup30" WL="mpileup30_B depth30 mpileup300 mpileup30_indel mpileup30_trim mpileup30_EA_pairs mpileup30_B_sOx" STATS="mpileup30 mpileup30_B mpileup30_B_sOx" bash scripts/gpu_record.sh
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
true
