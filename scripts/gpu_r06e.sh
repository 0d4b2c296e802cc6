#!/bin/bash
# Round 6, GPU session E: the extra-column emit with compile-time column steps (v_writelane cursors, no per-column mask registers, two dword / two
# qword store classes, 32-bit table decimals): parity on the device, the step, SQ instruction counters (session D: 4 171 vector instructions per strip).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_benchsize_parity.py -m gpu -q -o timeout=240 -k "extra_columns or generic_walker or output_extra or sOx" ) > $O/pytest_xf.log 2>&1; tail -3 $O/pytest_xf.log
python bench.py --steps 10 --warmup 3 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sOx', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
bash scripts/gpu_sq.sh mpileup30_B_sOx r06e/sq > $O/sq.log 2>&1; grep "emit_deep" $O/sq.log | cut -c1-900
true
