cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -n 16 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 5 $O/pytest.log
for w in depth30 mpileup30_B mpileup30; do
timeout 400 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read()); print('$w', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['traffic'])"
done
