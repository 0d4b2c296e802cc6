cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
(time python -m pytest tests/ -x -q -m gpu) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read()); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_check']['identical'], d['cpu_baseline']['value'])"
STA_BENCH_BACKEND=gloo STA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --cols 1048576 --verify --no-pmc 2> $O/bench_n2.err | grep '^{' > $O/bench_n2.json; python -c "
import json; d=json.loads(open('$O/bench_n2.json').read()); print('n2', d['value'], d['verify'])"
timeout 1500 python scripts/e2e_cons.py 8 4375000 /dev/shm/sta_e2e_cons8 > $O/e2e_cons8.log 2>&1; cat $O/e2e_cons8.log
