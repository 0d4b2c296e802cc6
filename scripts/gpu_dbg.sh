cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
for ov in 0 4 8 16; do
  STA_BAQ_OVERLAP=$ov timeout 300 python bench.py --workload mpileup30 --steps 5 --warmup 2 --no-pmc > gpurun_out/dbg/bench_ov$ov.json 2> gpurun_out/dbg/bench_ov$ov.err
  echo "overlap $ov rc=$?"; tail -1 gpurun_out/dbg/bench_ov$ov.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:5]}, (d.get("parity_check") or {}).get("identical"))
except Exception as e: print("nojson", e)'
done
STA_BAQ_OVERLAP=8 STA_BAQ_DEC=1 timeout 300 python bench.py --workload mpileup30 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json
d=json.loads(sys.stdin.read()); print("dec1 ov8", round(d["value"]), round(d["ms_per_step"],3))'
