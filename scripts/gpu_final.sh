# Round-end validation on the GPU box: full -m gpu suite, smoke(), benches, rocprofv3 kernel stats and PMC traffic.
# usage: TAG=r01 bash scripts/gpu_final.sh
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r01}
mkdir -p gpurun_out/final
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/final/pytest_gpu.log 2>&1
tail -3 gpurun_out/final/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
for wl in mpileup30 mpileup30_B depth30 mpileup300; do
  extra=""; [ "$wl" != mpileup30 ] && extra="--no-cpu-baseline"
  timeout 600 python bench.py --steps 5 --warmup 2 --workload $wl $extra > gpurun_out/final/bench_$wl.json 2> gpurun_out/final/bench_$wl.err
  tail -1 gpurun_out/final/bench_$wl.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in d["kernels_ms_per_step"].items()}))'
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -o $TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_B -o $TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload mpileup30_B > $R/gpurun_out/final/prof_B.log 2>&1
cd $R
TAG=$TAG bash scripts/gpu_pmc.sh mpileup30 > gpurun_out/final/pmc.log 2>&1; tail -8 gpurun_out/final/pmc.log
timeout 300 python scripts/e2e_cli.py 2000000 > gpurun_out/final/e2e.log 2>&1; tail -4 gpurun_out/final/e2e.log
