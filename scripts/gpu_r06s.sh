#!/bin/bash
# Round 6, GPU session S: (1) the whole -m gpu suite after the plan's stream re-ordering (the working quality copy and the list's class order run
# during the host round trip; the side streams wait for an event instead of a synchronised stream); (2) bench lines with the timed steps
# bracketing the roofline's kernel only (STA_BENCH_PROFILE_ALL=1: every launch, as before).  Kill: any parity failure; mpileup30 not below 6.7 ms.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -o timeout=240 -n 4 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for wl in mpileup30 mpileup300 mpileup30_indel mpileup30_B mpileup30_EA_pairs depth30; do
  for all in "" 1; do
  STA_BENCH_PROFILE_ALL=$all python bench.py --steps 20 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_${wl}_all$all.json
  python -c "import sys,json; d=json.loads(open('$O/bench_${wl}_all$all.json').read()); print('$wl all=$all', round(d['ms_per_step'],3), round(d['value']), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:7]})"
  done
done 2>&1 | tee $O/bench.log
true
