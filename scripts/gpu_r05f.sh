#!/bin/bash
# Round 5, GPU session F: the trimmed-reads workload (generator fixed) old vs new class-S grouping, decode-thread sweep and the fast exit on the
# bench's e2e shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
( time timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -q -x -o timeout=600 -p no:cacheprovider -k "mpileup30_trim" ) > $O/pytest_trim.log 2>&1; tail -3 $O/pytest_trim.log
run() { # lib workload
  cp samtools_amd/lib/lib$1.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:7]})"
}
for rep in 1 2; do for v in sesC new; do run $v mpileup30_trim; done; done 2>&1 | tee $O/ab_trim.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
for t in 16 14 12 8; do echo "== STA_IO_THREADS=$t"; STA_IO_THREADS=$t timeout 300 python scripts/e2e_bench_shape.py 2>&1 | grep -v GPU_INFLATE | cut -c1-330; done 2>&1 | tee $O/e2e_threads.log
echo "== no fast exit"; STA_NO_FAST_EXIT=1 timeout 300 python scripts/e2e_bench_shape.py 2>&1 | grep -v GPU_INFLATE | cut -c1-120 | tee -a $O/e2e_threads.log
true
