#!/bin/bash
# round 4, GPU session S: several workloads with one build (small-kernel folding), optional tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04s}; mkdir -p $O
summ() { tail -1 $1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read())
    print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:9]}), "parity", (d.get("parity_check") or {}).get("identical"))
except Exception as e: print("no json:", e)'; }
for w in ${WLS:-mpileup30_B depth30 mpileup30}; do
  env $ENVS timeout 300 python bench.py --steps ${STEPS:-20} --warmup 3 --workload $w --no-cpu-baseline --no-pmc --no-e2e > $O/b_$w.json 2> $O/b_$w.err; echo "$(summ $O/b_$w.json)"; tail -2 $O/b_$w.err
done
if [ -n "$TESTS" ]; then ( time timeout ${TT:-1500} python -m pytest $TESTS -m gpu -q -x -o timeout=300 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; fi
