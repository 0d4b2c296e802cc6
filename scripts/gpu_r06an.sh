#!/bin/bash
# Round 6, GPU session AN: the sharded launcher writing every rank's block in place (STA_SHARD_PWRITE=1, no gather): the shard driver tests.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06an; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_shard_driver.py tests/test_bench_pieces.py -q -x -o timeout=300 -n 4 ) > $O/pytest_shard.log 2>&1; tail -4 $O/pytest_shard.log
