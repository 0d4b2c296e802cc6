cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu -n 16 -k "depth or goldens or bedcov or coverage or shard" --deselect tests/test_gpu_benchsize_parity.py > $O/pytest_depth.log 2>&1; echo "pytest rc=$?" >> $O/pytest_depth.log) &
(timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "depth" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log) &
wait
tail -n 4 $O/pytest_depth.log; tail -n 4 $O/parity.log
timeout 300 python bench.py --workload depth30 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_depth30.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_depth30.json').read()); print('depth30', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['traffic'])"
STA_BENCH_BACKEND=gloo STA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --cols 1048576 --verify --no-pmc 2> $O/bench_n2.err | grep '^{' > $O/bench_n2.json; python -c "
import json; d=json.loads(open('$O/bench_n2.json').read()); print('n2', d['value'], d['verify'], d['config']['reads_per_gpu'])"
STA_BENCH_BACKEND=gloo STA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 3 --workload depth30 --steps 2 --warmup 1 --cols 1048576 --verify --no-pmc 2> $O/bench_n3.err | grep '^{' > $O/bench_n3.json; python -c "
import json; d=json.loads(open('$O/bench_n3.json').read()); print('n3 depth', d['value'], d['verify'])"
