# usage: [PROFILE=tag] [SKIPTESTS=1] bash scripts/gpu_check.sh [pytest-k-expr]   (runs on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIPTESTS" ]; then
( time timeout 900 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} ) > gpurun_out/pytest_gpu.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
fi
for wl in mpileup30 mpileup30_B depth30; do
  extra=""; [ "$wl" != mpileup30 ] && extra="--no-cpu-baseline"
  timeout 600 python bench.py --steps 5 --warmup 2 --workload $wl $extra > gpurun_out/bench_$wl.log 2>&1
  tail -1 gpurun_out/bench_$wl.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in d["kernels_ms_per_step"].items()}))'
done
if [ -n "$PROFILE" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$PROFILE -o $PROFILE -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$PROFILE.log 2>&1
  ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_$PROFILE | head
fi
