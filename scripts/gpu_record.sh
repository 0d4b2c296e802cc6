#!/bin/bash
# One GPU-box session: (optionally) the -m gpu suite, bench.py over a list of workloads, rocprofv3 kernel stats for some of them.
# Everything lands in gpurun_out/$TAG/; what is to be kept is copied into profiles/ by hand afterwards (gpurun_out is scratch).
#   TAG=r03a TESTS=1 WL="mpileup30 mpileup30_B" FULL="mpileup30" STATS="mpileup30 mpileup30_B" bash scripts/gpu_record.sh
# WL: workloads benched with --no-cpu-baseline --no-pmc; FULL: workloads benched with the CPU baseline, parity check and live PMC
# traffic (the driver's form); STATS: workloads re-run under rocprofv3 --kernel-trace --stats; TESTS: 1 = whole suite, or a pytest -k expression
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03}; O=gpurun_out/$TAG; mkdir -p $O
STEPS=${STEPS:-10}; WARM=${WARM:-3}
if [ -n "$TESTS" ]; then
  # (a hung kernel must not eat the GPU budget: 240 s per test, 900 s for the suite -- it needs ~160 s)
  if [ "$TESTS" = 1 ]; then ( time timeout 900 python -m pytest tests -m gpu -q -x -o timeout=240 ) > $O/pytest_gpu.log 2>&1
  else ( time timeout 900 python -m pytest tests -m gpu -q -o timeout=240 -k "$TESTS" ) > $O/pytest_gpu.log 2>&1; fi
  tail -4 $O/pytest_gpu.log
fi
summ() { tail -1 $1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read())
    print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:9]}), "parity", (d.get("parity_check") or {}).get("identical"))
except Exception as e: print("no json:", e)'; }
for wl in $FULL; do
  timeout 400 python bench.py --steps $STEPS --warmup $WARM --workload $wl $BENCH_EXTRA > $O/bench_${wl}_full.json 2> $O/bench_${wl}_full.err; summ $O/bench_${wl}_full.json
done
for wl in $WL; do
  timeout 200 python bench.py --steps $STEPS --warmup $WARM --workload $wl --no-cpu-baseline --no-pmc $BENCH_EXTRA > $O/bench_$wl.json 2> $O/bench_$wl.err; summ $O/bench_$wl.json
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wl in $STATS; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$wl -o $TAG -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-pmc --workload $wl $BENCH_EXTRA > $R/$O/prof_$wl.log 2>&1
  f=$(ls $R/$O/prof_$wl/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$O/${wl}_kernel_stats.csv && head -8 $f | cut -c1-150
done
cd $R
for wl in $SQ; do bash scripts/gpu_sq.sh $wl $TAG/sq_$wl > $O/sq_$wl.log 2>&1; tail -12 $O/sq_$wl.log; done
[ -n "$AFTER" ] && bash -c "$AFTER"
true
