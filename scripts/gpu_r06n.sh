#!/bin/bash
# Round 6, GPU session N: the BAQ list in class order (k_baq_list_partition: each list kernel's workgroups exist only for its own groups) --
# mpileup30_indel and the headline with and without (STA_BAQ_LIST_SORT=0), BAQ parity tests and the bench-size hashes.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
for rep in 1 2; do for v in 1 0; do for wl in mpileup30_indel mpileup30; do
  STA_BAQ_LIST_SORT=$v python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl sort=$v', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:5]}, d['output_sha256'][:12])"
done; done; done 2>&1 | tee $O/list_sort.log
( timeout 800 python -m pytest tests/test_gpu_benchsize_parity.py tests/test_gpu_synth.py -m gpu -q -o timeout=240 -k "indel or baq or trim or mpileup30 or EA" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
true
