#!/bin/bash
# Round 5, GPU session T: the single-walk generic emit with whole rows through LDS (one wave per workgroup, 30 / 40 KB slices) against eight-byte stores
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
run() { env $1 python bench.py --steps 8 --warmup 2 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]}, d['output_sha256'][:10])"; }
for v in STA_GENERIC_LDS_CAP=8192 STA_GENERIC_LDS_CAP=30720 STA_GENERIC_LDS_CAP=40960 STA_GENERIC_LDS_CAP=52000; do run $v; done 2>&1 | tee $O/generic_lds.log
true
