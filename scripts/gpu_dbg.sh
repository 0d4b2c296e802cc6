cd $GRAFT_REPO_ROOT
( timeout 400 python -m pytest tests/test_gpu_shard_driver.py tests/test_kputd.py tests/test_gpu_goldens.py -m gpu -q -o timeout=200 2>&1 | tail -15 | cut -c1-400 )
