#!/bin/bash
# Round 6, GPU session AK: the list kernels held to 128 workgroups each (strided walk) against one workgroup per four groups (STA_BAQ_LIST_MAX_WG=100000):
# mpileup30_indel at 4 M and 16 M columns (16 M before: 27.9 ms, the gather waiting 2.4 ms for a register file), mpileup30 at 16 M; then 64 / 256.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ak; mkdir -p $O
for cfg in "mpileup30_indel 4194304" "mpileup30_indel 16777216" "mpileup30 16777216"; do set -- $cfg; for mw in 100000 128 64 256; do
  STA_BAQ_LIST_MAX_WG=$mw python bench.py --steps 8 --warmup 2 --workload $1 --cols $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python -c "import sys,json; d=json.loads(open('$O/b.json').read()); print('$1 $2 max_wg $mw', round(d['ms_per_step'],3), round(d['value']), d['output_sha256'][:10], {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:5]})"
done; done 2>&1 | tee $O/bench.log
