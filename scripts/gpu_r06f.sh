#!/bin/bash
# Round 6, GPU session F: what the read-major emit costs at 30x WITHOUT extra columns (STA_EMIT_DEEP=1 on mpileup30_B): the base of session E's
# 3 637 vector instructions per strip.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
STA_EMIT_DEEP=1 python bench.py --steps 10 --warmup 3 --workload mpileup30_B --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B deep', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
STA_EMIT_DEEP=1 bash scripts/gpu_sq.sh mpileup30_B r06f/sq > $O/sq.log 2>&1; grep "emit_deep" $O/sq.log | cut -c1-900
STA_EMIT_DEEP=1 python bench.py --steps 10 --warmup 3 --workload mpileup30_B_s --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B_s deep', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
true
