#!/bin/bash
# Round 6, GPU session L: phase budget of k_mplp_emit_tile (mpileup30_B): time and SQ instruction counters with the column walk, the conversion and
# the flush taken out one at a time (STA_TILE_DIAG=1 / 2 / 3, wrong text) -- VERDICT r05 item 6's "ISA-level budget".
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
for d in 0 1 2 3; do
  STA_TILE_DIAG=$d python bench.py --steps 20 --warmup 5 --workload mpileup30_B --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('diag $d', round(d['ms_per_step'],4), {k: round(x,4) for k,x in list(d['kernels_ms_per_step'].items())[:3]})"
  STA_TILE_DIAG=$d bash scripts/gpu_sq.sh mpileup30_B r06l/sq$d > $O/sq$d.log 2>&1; grep "emit_tile" $O/sq$d.log | cut -c1-700
done 2>&1 | tee $O/tile_budget.log
true
