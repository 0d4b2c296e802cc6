# round 2, call B: first run of the single-pass kernels (k_mplp_fused / k_depth_fused)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -3 $O/smoke.log
if [ $rc -ne 0 ]; then
  STA_MPLP_LEGACY=1 timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_legacy.log 2>&1; echo "legacy smoke rc=$?"; tail -3 $O/smoke_legacy.log
  exit 1
fi
# goldens + synthetic + plp api + cov etc: the whole GPU suite in parallel (256 host cores)
(timeout 900 python -m pytest tests -q -m gpu -n 24 --deselect tests/test_gpu_benchsize_parity.py -x > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/pytest_all.log) &
(timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log) &
wait
tail -n 6 $O/pytest_all.log; tail -n 6 $O/parity.log
for wl in mpileup30_B mpileup300_B depth30 mpileup300 mpileup30; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$wl.json").read())
    print("$wl", round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", {k: round(v,3) for k,v in d["kernels_ms_per_step"].items()})
except Exception as e: print("$wl failed", e)
PY
done
for wl in mpileup30_B mpileup300_B; do
  STA_MPLP_LEGACY=1 timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_${wl}_legacy.json 2> $O/bench_${wl}_legacy.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${wl}_legacy.json").read())
    print("$wl legacy", round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", {k: round(v,3) for k,v in d["kernels_ms_per_step"].items()})
except Exception as e: print("$wl legacy failed", e)
PY
done
for lb in 5000 6144 10240; do
  STA_MPLP_LBUF=$lb timeout 300 python bench.py --workload mpileup30_B --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_lbuf$lb.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_lbuf$lb.json").read()); print("lbuf $lb", round(d["ms_per_step"],3), "ms", round(d["kernels_ms_per_step"].get("mplp_fused",0),3))
except Exception as e: print("lbuf $lb failed", e)
PY
done
cd /tmp
for wl in mpileup30_B depth30 mpileup300_B; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1
  head -12 $GRAFT_REPO_ROOT/$O/prof_$wl/p_kernel_stats.csv 2>/dev/null | cut -c1-150
done
