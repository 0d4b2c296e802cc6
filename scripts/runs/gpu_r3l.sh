cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
E2E_QUICK=1 E2E_THREADS="16/1,16/4,24/4,24/8,32/8" timeout 900 python scripts/e2e_big.py 8 4375000 /dev/shm/sta_e2e > $O/e2e_stage_threads.log 2>&1; echo "e2e rc=$?" >> $O/e2e_stage_threads.log
cat $O/e2e_stage_threads.log | cut -c1-330
