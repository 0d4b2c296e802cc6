#!/bin/bash
# Round 5, what was left of the GPU budget (0.9 min): the window-cut fix (retire() of both input lanes) through its regression test on the device
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z4; mkdir -p $O
( time timeout 28 python -m pytest tests/test_gpu_synth.py -q -m gpu -o timeout=25 -p no:cacheprovider -k "supplementary_upstream or three_records" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
true
