#!/bin/bash
# Round 6, GPU session AL: after the list kernels' strided walk: the whole -m gpu suite again, the default bench run, and a second batch of parity
# hunts on the device build (hunt9 12 seeds, hunt8 / hunt6 / hunt5 two fresh seeds each).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06al; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -o timeout=600 -n 4 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print('default', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['parity_check']['identical'], d['e2e']['identical_to_oracle']['all'])"
python bench.py --steps 10 --warmup 3 --workload mpileup30_indel --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 > $O/bench_mpileup30_indel.json
timeout 900 python scripts/hunt9.py 920 932 > $O/hunt9.log 2>&1; tail -1 $O/hunt9.log
HUNT8_CASES=40 timeout 1200 python scripts/hunt8.py 9811 9812 > $O/hunt8.log 2>&1; tail -1 $O/hunt8.log
HUNT6_CASES=40 timeout 1200 python scripts/hunt6.py 9611 9612 > $O/hunt6.log 2>&1; tail -1 $O/hunt6.log
HUNT5_CASES=40 timeout 1200 python scripts/hunt5.py 9511 9512 > $O/hunt5.log 2>&1; tail -1 $O/hunt5.log
grep -h -v "^ok\|^skip\|^seed" $O/hunt*.log | sort | uniq -c | head
