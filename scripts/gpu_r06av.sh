#!/bin/bash
# Round 6, GPU session AV: the last commit (sta_bgzf_inflate clears a stale sticky error): the whole -m gpu suite once more and smoke().
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06av; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -o timeout=600 -n 4 ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpileup30', round(d['ms_per_step'],3), round(d['value']))"
