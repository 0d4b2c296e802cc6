#!/bin/bash
# round 4, GPU session A: TCC counter calibration, SQ counters of the BAQ pair at bench size, the self-launching bench test
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04a; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp
  $GRAFT_REPO_ROOT/scripts/ubench/pmc_calib > $O/calib_plain.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$n -o c -- $GRAFT_REPO_ROOT/scripts/ubench/pmc_calib > $O/calib_$n.log 2>&1
  done )
python - <<PY
import csv, glob
for f in sorted(glob.glob("$O/calib_*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        print(f.split("/")[-2], row["Kernel_Name"][:40], row["Counter_Name"], row["Counter_Value"])
PY
cat $O/calib_plain.log
bash scripts/gpu_sq.sh mpileup30 r04a/sq_mpileup30 > $O/sq_mpileup30.log 2>&1; grep "k_baq" $O/sq_mpileup30.log
( time timeout 900 python -m pytest tests/test_bench_launch.py -m gpu -q -x -o timeout=800 ) > $O/pytest_launch.log 2>&1; tail -5 $O/pytest_launch.log
