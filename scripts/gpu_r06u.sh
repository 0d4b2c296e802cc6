#!/bin/bash
# Round 6, GPU session U: pairs of two plain reads resolved by the whole wave (kernels_overlap.hip resolve_pairs_wave): k_name_groups was 0.50 ms of
# the 0.66 ms overlap pass of mpileup30_EA_pairs.  Parity: the overlap / pair tests + the bench's parity_check; kill: overlap not below 0.4 ms.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -o timeout=240 -n 4 -k "overlap or olap or pair or mate or template or regression" ) > $O/pytest_olap.log 2>&1; tail -3 $O/pytest_olap.log
for wl in mpileup30_EA_pairs mpileup30_B_pairs mpileup30; do
  python bench.py --steps 20 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', round(d['ms_per_step'],3), round(d['value']), d.get('parity_check'), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:8]})"
done 2>&1 | tee $O/bench.log
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-e2e --workload mpileup30_EA_pairs > $R/$O/prof.log 2>&1
head -12 $R/$O/prof/p_kernel_stats.csv | cut -c1-150
