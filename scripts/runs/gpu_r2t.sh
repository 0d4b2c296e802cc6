cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2t; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_cabi_client.py -q -m gpu -n 12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 30 $O/pytest.log
