# round 2, call C: second version of the single-pass kernels (ticket batches, packed tile, batched loads, listed giant waves)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 180 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 $O/smoke.log
[ $rc -ne 0 ] && exit 1
(timeout 900 python -m pytest tests -q -m gpu -n 24 --deselect tests/test_gpu_benchsize_parity.py -x > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/pytest_all.log) &
(timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -x -q -m gpu -k "bench_window" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log) &
wait
tail -n 4 $O/pytest_all.log; tail -n 4 $O/parity.log
run() { # name, env..., workload
  name=$1; wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read())
    print("$name", round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:4]})
except Exception as e: print("$name failed", e)
PY
}
run mpileup30_B mpileup30_B X=1
run mpileup300_B mpileup300_B X=1
run depth30 depth30 X=1
run m30B_tpb1 mpileup30_B STA_FUSED_TPB=1
run m30B_tpb8 mpileup30_B STA_FUSED_TPB=8
run d30_tpb1 depth30 STA_FUSED_TPB=1
run d30_tpb16 depth30 STA_FUSED_TPB=16
run m30B_lb6144 mpileup30_B STA_MPLP_LBUF=6144
run m30B_lb10240 mpileup30_B STA_MPLP_LBUF=10240
run m300B_lb10240 mpileup300_B STA_MPLP_LBUF=10240
run m300B_lb4096 mpileup300_B STA_MPLP_LBUF=4096
bash scripts/gpu_sq.sh mpileup30_B r2c/sq_m30B > $O/sq_m30B.txt 2>&1; tail -4 $O/sq_m30B.txt
bash scripts/gpu_sq.sh depth30 r2c/sq_d30 > $O/sq_d30.txt 2>&1; tail -4 $O/sq_d30.txt
