#!/bin/bash
# Round 5, the very last GPU call (2.9 GPU-minutes left): the golden and synthetic parity files in full on the device after the
# round's last kernel changes (k_prep_reads -6 rule, cap_mapq on SEQ-less records, inflate match-copy fence).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z3; mkdir -p $O
( time timeout 130 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_synth.py tests/test_gpu_deep_emit.py -q -m gpu -o timeout=120 -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
true
