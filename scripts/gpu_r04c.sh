#!/bin/bash
# round 4, GPU session C: class-S kernel variants (row-stream mode, diagnostic without the stream)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04c}; mkdir -p $O
summ() { tail -1 $1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read())
    print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:7]}), "parity", (d.get("parity_check") or {}).get("identical"))
except Exception as e: print("no json:", e)'; }
for v in ${VARIANTS:-"X=1" "STA_BAQ7S_MODE=1" "STA_BAQ7S_MODE=2" "STA_BAQ7S_MODE=2,STA_BAQ7S_WAVES_PER_CU=4"}; do
  envs=$(echo $v | tr ',' ' ')
  env $envs timeout 200 python bench.py --steps 10 --warmup 3 --workload ${WL:-mpileup30} --no-cpu-baseline --no-pmc > $O/v_$v.json 2> $O/v_$v.err; echo "$v: $(summ $O/v_$v.json)"
done
if [ -n "$FULL" ]; then timeout 600 python bench.py --steps 10 --warmup 3 --workload mpileup30 --no-pmc > $O/bench_mpileup30_full.json 2> $O/bench_mpileup30_full.err; summ $O/bench_mpileup30_full.json; fi
if [ -n "$TESTS" ]; then ( time timeout 900 python -m pytest $TESTS -m gpu -q -x -o timeout=300 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; fi
