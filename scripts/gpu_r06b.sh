#!/bin/bash
# Round 6, GPU session B: first device run of the read-major extra columns (k_mplp_len_rm<true> / k_mplp_emit_deep<true>: DPP prefix sums and
# overlapping unaligned stores have only run on the CPU emulation so far), of the one-round-trip BAQ plan and of the CLI's fast exit.
# Whole -m gpu suite; mpileup30_B_sOx with the walkers (STA_XFAST=0) and read-major; the headline and the indel workload; kernel stats of
# the _sOx window; the e2e commands.  Kill criterion for the extra columns: _sOx step > 3.5 ms -> profile before anything else.  ~15 GPU-minutes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
TAG=r06b TESTS=1 FULL="" WL="mpileup30_B_sOx mpileup30 mpileup30_indel mpileup30_B" STATS="mpileup30_B_sOx" bash scripts/gpu_record.sh
STA_XFAST=0 timeout 200 python bench.py --steps 10 --warmup 3 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc > $O/bench_sOx_walkers.json 2> $O/bench_sOx_walkers.err
tail -1 $O/bench_sOx_walkers.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("walkers:", round(d["ms_per_step"],3), "ms", d["kernels_ms_per_step"])'
timeout 600 python scripts/e2e_bench_shape.py > $O/e2e.log 2>&1; cut -c1-330 $O/e2e.log
true
