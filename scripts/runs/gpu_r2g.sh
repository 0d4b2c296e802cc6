cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu -n 24 --deselect tests/test_gpu_benchsize_parity.py > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/pytest_all.log) &
(timeout 900 python scripts/e2e_big.py 8 4375000 /dev/shm/sta_e2e > $O/e2e_big.log 2>&1; echo "e2e rc=$?" >> $O/e2e_big.log) &
wait
tail -n 5 $O/pytest_all.log; cat $O/e2e_big.log
