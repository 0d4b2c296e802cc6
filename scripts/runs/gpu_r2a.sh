# round 2, call A: bench-size parity tests, external C client, new bench line (live PMC), sharded bench at N=2 on one device (gloo)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
(timeout 900 python -m pytest tests/test_gpu_benchsize_parity.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log) &
(timeout 600 python -m pytest tests/test_gpu_cabi_client.py -q -m gpu -n 4 > $O/cabi.log 2>&1; echo "cabi rc=$?" >> $O/cabi.log) &
wait
tail -5 $O/parity.log $O/cabi.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_mpileup30.json 2> $O/bench_mpileup30.err; echo "bench rc=$?"; tail -c 1500 $O/bench_mpileup30.json
STA_BENCH_BACKEND=gloo STA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --cols 1048576 --verify --no-pmc > $O/bench_n2.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"; tail -c 600 $O/bench_n2.json; tail -5 $O/bench_n2.err
