#!/bin/bash
# Round 6, GPU session X (diagnostic build, results not parity-clean): where k_name_groups' 0.29 ms goes -- the kernel cut short behind the
# table probe (1), behind the chain walk (2), in front of the pairs' resolution (3), whole (0).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
for dg in 0 7 8; do
  STA_OLAP_DIAG=$dg python bench.py --steps 20 --warmup 3 --workload mpileup30_B_pairs --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/b_$dg.json
  python -c "import sys,json; d=json.loads(open('$O/b_$dg.json').read()); print('diag $dg', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:8]})"
done 2>&1 | tee $O/diag.log
