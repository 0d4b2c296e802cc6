#!/bin/bash
# Round 6, GPU session AU: after the consensus fix (a window without a base): the whole -m gpu suite, the two hunt seeds that found it, the default run.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06au; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -o timeout=600 -n 4 ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
HUNT6_CASES=40 timeout 900 python scripts/hunt6.py 9621 > $O/hunt6.log 2>&1; tail -1 $O/hunt6.log
HUNT5_CASES=40 timeout 900 python scripts/hunt5.py 9522 > $O/hunt5.log 2>&1; tail -1 $O/hunt5.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; grep real $O/bench_default.time
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print('default', round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['parity_check']['identical'], d['e2e']['identical_to_oracle']['all'])"
