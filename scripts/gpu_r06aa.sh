#!/bin/bash
# Round 6, GPU session AA: the headline window at 16 M columns (session Z: 18 639 -> 20 949 Mbases/s): (1) `python bench.py` as the driver runs
# it, with its wall time; (2) the whole-window parity test at that size; (3) the sharded launcher at --gpus 1 on the new default.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06aa; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['config']['window_cols_per_gpu'], d['roofline'], d['parity_check']['identical'], d['cpu_baseline']['value'], {k: (v.get('seconds') if isinstance(v, dict) else v) for k, v in d['e2e'].items()} if isinstance(d.get('e2e'), dict) else None)" 2>&1 | cut -c1-900
( time timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -m gpu -q -x -o timeout=600 -k "headline_window" ) > $O/pytest_headline.log 2>&1; tail -5 $O/pytest_headline.log
