cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_consensus.py -q -m gpu -n 12 -x > $O/pytest_cons.log 2>&1; echo "pytest rc=$?" >> $O/pytest_cons.log)
tail -n 25 $O/pytest_cons.log
for w in consensus30_simple consensus30; do
timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; tail -n 3 $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read()); print('$w', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
done
