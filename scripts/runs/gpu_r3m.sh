cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
E2E_QUICK=1 E2E_THREADS="16/1,16/4,24/4" timeout 600 python scripts/e2e_big.py 8 4375000 /dev/shm/sta_e2e > $O/e2e_stage_threads.log 2>&1; echo "e2e rc=$?" >> $O/e2e_stage_threads.log
cat $O/e2e_stage_threads.log | cut -c1-330
rm -rf /dev/shm/sta_e2e
(time python -m pytest tests/ -x -q -m gpu) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read()); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_check']['identical'], d['cpu_baseline']['value'])"
