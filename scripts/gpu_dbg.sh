cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
for pct in 115 130 150 200; do
  STA_TILE_CAP_PCT=$pct timeout 200 python bench.py --workload mpileup30_B --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/dbg/b.json 2> gpurun_out/dbg/b.err
  echo "pct $pct rc=$?"; tail -1 gpurun_out/dbg/b.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:3]})
except Exception as e: print("nojson", e)'
done
