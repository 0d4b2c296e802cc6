cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2n; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_consensus.py -q -m gpu -n 12 -x > $O/pytest_cons.log 2>&1; echo "pytest rc=$?" >> $O/pytest_cons.log)
tail -n 5 $O/pytest_cons.log
for w in consensus30_simple consensus30; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; tail -n 2 $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read()); print('$w', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline'])"
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cons30 -- python $GRAFT_REPO_ROOT/bench.py --workload consensus30 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; find $O/prof_cons30 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {}'
