#!/bin/bash
# Round 6, GPU session AH: k_mplp_emit_tile held to five waves per SIMD (96 registers, 13 spilled dwords; STA_TILE_DIAG=5) against the product (121
# registers, four waves): mpileup30_B at 4 M and 16 M columns, text hash included.  Kill: not faster by 3 %.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ah; mkdir -p $O
for cols in 4194304 16777216; do for dg in 0 5 0 5; do
  STA_TILE_DIAG=$dg python bench.py --steps 20 --warmup 3 --workload mpileup30_B --cols $cols --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python -c "import sys,json; d=json.loads(open('$O/b.json').read()); print('cols $cols diag $dg', round(d['ms_per_step'],3), d['output_sha256'][:12], {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]})"
done; done 2>&1 | tee $O/bench.log
