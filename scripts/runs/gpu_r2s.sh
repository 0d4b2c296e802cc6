cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2s; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_cabi_client.py tests/test_gpu_consensus.py -q -m gpu -n 12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 30 $O/pytest.log
timeout 400 python bench.py --workload consensus30 --steps 5 --warmup 2 > $O/bench_consensus30.json 2> $O/bench.err; tail -n 2 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_consensus30.json').read()); print(d['value'], d['ms_per_step'], d.get('parity_check'))"
