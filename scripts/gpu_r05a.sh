#!/bin/bash
# Round 5, GPU session A: parity of every k_baq7s build, A/B timings inside one box, XCD mapping of the emit kernels with live PMC
# traffic, SQ counters of k_baq7s itself (modes 0 and 16).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -q -k "env_only" -o timeout=300 -p no:cacheprovider ) > $O/parity_modes.log 2>&1; tail -5 $O/parity_modes.log
cp samtools_amd/lib/libsamtools_amd.so samtools_amd/lib/libnew.so
run() { # lib mode workload extra-env
  cp samtools_amd/lib/lib$1.so samtools_amd/lib/libsamtools_amd.so
  env STA_BAQ7S_MODE=$2 $4 python bench.py --steps 10 --warmup 3 --workload $3 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 mode=$2 $4', '$3', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
}
for rep in 1 2; do
  run prev 0 mpileup30
  for m in 0 16 32 48 64 80; do run new $m mpileup30; done
done 2>&1 | tee $O/ab_baq.log
for rep in 1 2; do for x in 0 1; do run new 16 mpileup30_B STA_XCD_MAP=$x; run new 16 mpileup300_B STA_XCD_MAP=$x; done; done 2>&1 | tee $O/ab_xcd.log
cp samtools_amd/lib/libnew.so samtools_amd/lib/libsamtools_amd.so
for x in 0 1; do
  STA_XCD_MAP=$x timeout 300 python bench.py --steps 5 --warmup 2 --workload mpileup30_B --no-cpu-baseline --no-e2e > $O/bench_mpileup30_B_xcd$x.json 2> $O/bench_mpileup30_B_xcd$x.err
  tail -1 $O/bench_mpileup30_B_xcd$x.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd=$x', d['ms_per_step'], d['roofline'])"
done 2>&1 | tee $O/xcd_traffic.log
for m in 0 16; do
  STA_BAQ7S_MODE=$m bash scripts/gpu_sq.sh mpileup30 r05a/sq_m$m > $O/sq_m$m.log 2>&1; grep -E 'k_baq7s' $O/sq_m$m.log
done
true
