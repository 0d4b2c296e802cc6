#!/bin/bash
# Round 5, GPU session R: where the generic single-walk emit's 8.9 ms goes (STA_GENERIC_DIAG, timing only): no writing walk / writing walk without extras
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
run() { env $1 python bench.py --steps 8 --warmup 2 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]})"; }
for v in STA_GENERIC_DIAG=0 STA_GENERIC_DIAG=1 STA_GENERIC_DIAG=3 STA_GENERIC_DIAG=0; do run $v; done 2>&1 | tee $O/generic_diag.log
true
