#!/bin/bash
# scratch: A/B a runtime setting on one workload (bench only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-try}; mkdir -p $O
for v in base $VARIANTS; do
  case $v in
    base) envs="" ;;
    nointr) envs="HSA_ENABLE_INTERRUPT=0" ;;
    spin) envs="HIP_LAUNCH_BLOCKING=0 GPU_MAX_HW_QUEUES=2" ;;
    *) envs="$v" ;;
  esac
  for wl in ${WL:-mpileup30_B}; do
    env $envs timeout 200 python bench.py --steps ${STEPS:-20} --warmup 5 --workload $wl --no-cpu-baseline --no-pmc > $O/${wl}_$v.json 2> $O/${wl}_$v.err
    tail -1 $O/${wl}_$v.json | python -c 'import sys,json
d=json.loads(sys.stdin.read()); print("'$v'", d["config"]["workload"], round(d["ms_per_step"],4), "ms", json.dumps({k: round(x,3) for k,x in list(d["kernels_ms_per_step"].items())[:4]}))' 2>/dev/null || tail -2 $O/${wl}_$v.err
  done
done
