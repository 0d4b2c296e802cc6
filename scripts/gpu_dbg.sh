cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
( timeout 400 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py tests/test_gpu_deep_emit.py tests/test_gpu_fullsize.py -m gpu -q -x -o timeout=150 2>&1 | tail -4 | cut -c1-300 )
( timeout 300 python -m pytest tests/test_gpu_benchsize_parity.py -m gpu -q -x -o timeout=250 -k "indel or mpileup30_B or mpileup100_B or hotspot" 2>&1 | tail -3 | cut -c1-300 )
for wl in mpileup30_indel mpileup30_B; do
  timeout 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/dbg/bench_$wl.json 2> gpurun_out/dbg/bench_$wl.err
  echo "$wl rc=$?"; tail -1 gpurun_out/dbg/bench_$wl.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:7]})
except Exception as e: print("nojson", e)'
done
