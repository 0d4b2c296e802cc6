#!/bin/bash
# Round 6, GPU session AM: the working quality copy on its own stream beside k_prep_reads (STA_NO_QP_STREAM=1: behind it, as before): mpileup30 at 16 M
# and 4 M columns, mpileup30_EA_pairs; text hashes.  Kill: not faster.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06am; mkdir -p $O
for cfg in "mpileup30 16777216" "mpileup30 4194304" "mpileup30_EA_pairs 4194304"; do set -- $cfg; for nq in 1 "" 1 ""; do
  STA_NO_QP_STREAM=$nq python bench.py --steps 10 --warmup 3 --workload $1 --cols $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/b.json
  python -c "import sys,json; d=json.loads(open('$O/b.json').read()); print('$1 $2 no_qp_stream=$nq', round(d['ms_per_step'],3), round(d['value']), d['output_sha256'][:10], {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items()) if k in ('baq_s','prep_reads','qual_prep')})"
done; done 2>&1 | tee $O/bench.log
