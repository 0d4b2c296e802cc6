#!/bin/bash
# Round 6, GPU session O ("final" record of the round's state): whole -m gpu suite, smoke(), the default bench.py run exactly as the driver runs it
# (CPU baseline, live PMC traffic, parity check, e2e), the other workloads without baseline, rocprofv3 kernel stats of the headline and of
# mpileup30_B / _B_sOx / _indel / depth30.  ~12 GPU-minutes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -o timeout=240 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
TAG=r06o TESTS="" FULL="" WL="mpileup30_B depth30 mpileup30_EA_pairs mpileup30_indel mpileup30_B_sOx mpileup300 mpileup300_B mpileup100 mpileup30_trim mpileup30_B_s" STATS="mpileup30 mpileup30_B mpileup30_B_sOx mpileup30_indel depth30" bash scripts/gpu_record.sh
true
