#!/bin/bash
# Round 5, GPU session X: the -d cap bound from the input lane's spans (BAM + small windows), hunt4 seeds 6 / 7, cap tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py -q -o timeout=240 -p no:cacheprovider -k "depth_cap or three_records" ) > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 400 python scripts/hunt4.py 6 7 > $O/hunt4.log 2>&1; tail -1 $O/hunt4.log
( time timeout 600 python -m pytest tests -m gpu -q -o timeout=300 -p no:cacheprovider -k "maxcnt or cap or max_depth" ) > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
true
