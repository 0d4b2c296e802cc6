cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
for w in consensus30_simple consensus30; do
timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; tail -n 3 $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read()); print('$w', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
done
