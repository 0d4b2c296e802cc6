#!/bin/bash
# Round 6, GPU session Q: (1) the sharded -C test on the device; (2) row heads with 32-bit decimals: mpileup300 / mpileup300_B / mpileup30_B_sOx /
# mpileup30_B / mpileup30 against the final record of session O (emit_deep 0.579 / 0.55, _sOx 3.45 ms); kill: any of them slower by > 2 %.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_shard_driver.py -m gpu -q -x -o timeout=240 ) > $O/pytest_shard.log 2>&1; tail -3 $O/pytest_shard.log
for wl in mpileup300 mpileup300_B mpileup30_B_sOx mpileup30_B mpileup30; do
  python bench.py --steps 10 --warmup 3 --workload $wl --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_$wl.json
  python -c "import sys,json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:6]})"
done 2>&1 | tee $O/bench.log
true
