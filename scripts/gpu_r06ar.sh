#!/bin/bash
# Round 6, GPU session AR: kernels_baq.hip under other instruction-scheduling strategies of the compiler (max-ilp | occupancy bias 0 | the AMDGPU
# register-pressure trackers) against the product build, one box, text hashes: the class-S kernel is issue-bound with two waves per SIMD.
cd $GRAFT_REPO_ROOT
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
for rep in 1 2; do for v in base ilp bias0 trk; do
  cp samtools_amd/lib/libv_$v.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload mpileup30 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['value']), d['output_sha256'][:10], {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:1]})"
done; done 2>&1 | tee gpurun_out/r06ar_bench.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
