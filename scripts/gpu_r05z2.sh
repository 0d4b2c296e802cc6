#!/bin/bash
# Round 5, last GPU call (3.6 GPU-minutes were left): the three product changes of the round's second half on the real device --
# k_prep_reads' -6 / missing-QUAL rule, cap_mapq on records without SEQ, the fenced passes of the device inflate's match copy -- through
# their regression tests, the hand-derived -C vectors, the device inflate tests, and one hunt5 seed; no torch import anywhere.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z2; mkdir -p $O
( time timeout 150 python -m pytest tests/test_gpu_synth.py tests/test_gpu_bgzf_inflate.py tests/test_cap_mapq_vectors.py tests/test_gpu_calmd.py -q -m gpu -o timeout=120 -p no:cacheprovider \
    -k "illumina13 or records_without_seq or adjust_mq or bgzf or inflate or vectors or cap_mapq or blocks or damaged or every_block" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time HUNT5_CASES=25 timeout 100 python scripts/hunt5.py 31 ) > $O/hunt5_seed31.log 2>&1; tail -2 $O/hunt5_seed31.log
true
