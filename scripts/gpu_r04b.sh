#!/bin/bash
# round 4, GPU session B: the class-S BAQ kernel -- parity (bench sample vs oracle, BAQ tests), then A/B timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
summ() { tail -1 $1 | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read())
    print(d["config"]["workload"], round(d["value"]), "Mb/s", round(d["ms_per_step"],3), "ms", json.dumps({k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:7]}), "parity", (d.get("parity_check") or {}).get("identical"))
except Exception as e: print("no json:", e)'; }
timeout 600 python bench.py --steps 10 --warmup 3 --workload mpileup30 --no-pmc > $O/bench_mpileup30_full.json 2> $O/bench_mpileup30_full.err; summ $O/bench_mpileup30_full.json; tail -3 $O/bench_mpileup30_full.err
for v in "STA_BAQ_CLASS_S=0" "STA_BAQ7S_LEAD=0" "STA_BAQ7S_LEAD=3" "STA_BAQ7S_WAVES_PER_CU=4" "STA_BAQ7S_WAVES_PER_CU=6"; do
  env $v timeout 200 python bench.py --steps 10 --warmup 3 --workload mpileup30 --no-cpu-baseline --no-pmc > $O/v_$v.json 2> $O/v_$v.err; echo "$v: $(summ $O/v_$v.json)"
done
for wl in mpileup30_indel mpileup300 mpileup30_EA_pairs; do timeout 300 python bench.py --steps 5 --warmup 2 --workload $wl --no-cpu-baseline --no-pmc > $O/bench_$wl.json 2> $O/bench_$wl.err; summ $O/bench_$wl.json; done
( time timeout 900 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py tests/test_gpu_calmd.py -m gpu -q -x -o timeout=300 ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
