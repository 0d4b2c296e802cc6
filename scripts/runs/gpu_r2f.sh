cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $O/smoke.log
[ $rc -ne 0 ] && exit 1
(timeout 900 python -m pytest tests -q -m gpu -n 24 --deselect tests/test_gpu_benchsize_parity.py > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/pytest_all.log) &
(timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log) &
wait
tail -n 6 $O/pytest_all.log; tail -n 6 $O/parity.log
run() { name=$1; wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read())
    print("$name", round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:6]})
except Exception as e: print("$name failed", e)
PY
}
run mpileup30 mpileup30 X=1
run mpileup300 mpileup300 X=1
run depth30 depth30 X=1
run d30_sub2 depth30 STA_DEPTH_SUB=2
run d30_sub8 depth30 STA_DEPTH_SUB=8
run d30_sub1 depth30 STA_DEPTH_SUB=1
