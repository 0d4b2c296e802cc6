cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_shard_driver.py -q -m gpu -n 12 > $O/shard.log 2>&1; echo "shard rc=$?" >> $O/shard.log) &
(timeout 900 python -m pytest tests -q -m gpu -n 16 --deselect tests/test_gpu_benchsize_parity.py --deselect tests/test_gpu_shard_driver.py > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $O/pytest_all.log) &
wait
tail -n 12 $O/shard.log; tail -n 8 $O/pytest_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_mpileup30.json 2> $O/bench_mpileup30.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_mpileup30.json').read()); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:900]); print(d.get('parity_check')); print(d['cpu_baseline'])"
for wl in mpileup30_B depth30; do timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 > $O/bench_$wl.json 2> $O/bench_$wl.err; python -c "
import json; d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600]); print(d.get('parity_check'))"; done
