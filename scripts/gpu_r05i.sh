#!/bin/bash
# Round 5, GPU session I: the single-walk generic emit with wide text loads and the read header asked for one read ahead;
# 128 registers (four waves per SIMD, some spills) against what the allocator takes by itself (libnoocc.so).  Kill: < 1.25x on k_mplp_emit.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py tests/test_kputd.py -q -o timeout=240 -p no:cacheprovider -k "generic_walker or mpileup_sO or output_extra or 79 or kputd" ) > $O/pytest_generic.log 2>&1; tail -3 $O/pytest_generic.log
( time timeout 600 python -m pytest tests/test_gpu_benchsize_parity.py -q -o timeout=400 -p no:cacheprovider -k "sOx" ) > $O/pytest_sox.log 2>&1; tail -3 $O/pytest_sox.log
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
run() { # lib label
  cp $1 samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
}
for rep in 1 2; do run /tmp/lib_keep.so occ4; run samtools_amd/lib/libnoocc.so noocc; done 2>&1 | tee $O/ab_generic.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
true
