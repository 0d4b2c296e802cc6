#!/bin/bash
# Round 6, GPU session C: the extra-column emit after its rewrite (units = separator + field, two overlapping stores, table decimals; session B:
# k_mplp_emit_deep<true> 4.08 ms, VALU-bound) compiled for two (libxf2) and three (libxf3) waves per SIMD; its tests on the device;
# per-window trace of the three e2e commands (STA_DRIVER_TIMING=3: where the device thread's time per window goes).  ~8 GPU-minutes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_benchsize_parity.py -m gpu -q -o timeout=240 -k "extra_columns or generic_walker or output_extra or sOx" ) > $O/pytest_xf.log 2>&1; tail -3 $O/pytest_xf.log
LIBS="xf2 xf3" REPS=2 STEPS=10 WL=mpileup30_B_sOx bash scripts/gpu_ab_libs.sh 2>&1 | tee $O/ab_xf.log
STA_E2E_TIMING=3 timeout 600 python scripts/e2e_bench_shape.py > $O/e2e_trace.log 2>&1; grep -E "plain run|window 0\]|window 1\]|window 2\]|window 7\]|window 15\]|timeline" $O/e2e_trace.log | cut -c1-260 | head -80
true
