#!/bin/bash
# Round 6, GPU session K: depth30 with k_depth_fused split into count | wave scan | emit (no ticket, no look-back chain) against the single launch.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_benchsize_parity.py tests/test_gpu_fullsize.py -m gpu -q -o timeout=240 -k "depth" ) > $O/pytest_depth.log 2>&1; tail -3 $O/pytest_depth.log
for rep in 1 2; do for form in fused split; do
  STA_DEPTH_FORM=$form python bench.py --steps 20 --warmup 5 --workload depth30 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$form', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['kernels_ms_per_step'].items()}, d['output_sha256'][:12])"
done; done 2>&1 | tee $O/ab_depth.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_depth30 -o r06k -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-e2e --workload depth30 > $R/$O/prof_depth30.log 2>&1
f=$(ls $R/$O/prof_depth30/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$O/depth30_kernel_stats.csv && head -8 $f | cut -c1-160
true
