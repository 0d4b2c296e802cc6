#!/bin/bash
# Round 6, GPU session Y: k_glf_cols with a few LDS slots per column instead of one byte per key (37.5 KB of LDS per wave = one wave per SIMD;
# 6.6 ms for 4 M columns, 1.4 % of the HBM roof): glf30 at STA_GLF_SLOTS=0 (as before) | 32 | 64, the glf tests on the device.
# Kill: parity; neither slot form below 3.3 ms.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_glf.py tests/test_gpu_benchsize_parity.py -m gpu -q -x -o timeout=240 -n 4 -k "glf" ) > $O/pytest_glf.log 2>&1; tail -3 $O/pytest_glf.log
for sl in 16 32 auto; do
  STA_GLF_SLOTS=$([ $sl = auto ] || echo $sl) python bench.py --steps 10 --warmup 2 --workload glf30 --no-cpu-baseline --no-pmc --no-e2e --verify 2>/dev/null | tail -1 > $O/bench_glf30_$sl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_glf30_$sl.json').read()); print('glf30 slots=$sl', round(d['ms_per_step'],3), round(d['value']), d.get('parity_check'), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
done 2>&1 | tee $O/bench.log
