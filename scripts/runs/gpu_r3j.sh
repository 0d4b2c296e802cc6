cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
(time python -m pytest tests/ -x -q -m gpu) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read()); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_check']['identical'], d['cpu_baseline']['value'])"
for w in mpileup300_B mpileup30_B depth30; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-pmc > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read()); print('$w', d['value'], round(d['ms_per_step'],3), d['parity_check']['identical'], {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})"; done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof300 -o m300 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload mpileup300_B --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/prof300 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/mpileup300_B_kernel_stats.csv; head -8 $O/mpileup300_B_kernel_stats.csv | cut -c1-200
rm -rf $O/prof300
