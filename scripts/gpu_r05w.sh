#!/bin/bash
# Round 5, GPU session W: overlap_remove by name for templates with three records in the hash (hunt4 seeds 4 / 5), the hunts again, overlap tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05w; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py -q -o timeout=240 -p no:cacheprovider -k "three_records or generic_walker or adjust_mq" ) > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 600 python scripts/hunt4.py 4 5 6 > $O/hunt4.log 2>&1; tail -1 $O/hunt4.log
timeout 600 python scripts/hunt3.py 1 2 3 4 > $O/hunt3.log 2>&1; grep -c "^ok" $O/hunt3.log; grep "^FAIL" $O/hunt3.log | head -5
( time timeout 800 python -m pytest tests -m gpu -q -o timeout=300 -p no:cacheprovider -k "overlap or pairs or olap or goldens or plp_api or cabi" ) > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
true
