#!/bin/bash
# Round 6, GPU session M: k_mplp_emit_tile without its maxend probe (the group's first live read comes from k_mplp_len_rm's third wave):
# mpileup30_B step and kernel times, sha256 against the recorded one.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
for rep in 1 2 3; do
python bench.py --steps 20 --warmup 5 --workload mpileup30_B --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', round(d['ms_per_step'],4), {k: round(x,4) for k,x in list(d['kernels_ms_per_step'].items())[:4]}, d['output_sha256'][:12])"
done 2>&1 | tee $O/tile.log
( timeout 600 python -m pytest tests/test_gpu_benchsize_parity.py -m gpu -q -o timeout=240 ) > $O/pytest_bench.log 2>&1; tail -2 $O/pytest_bench.log
true
