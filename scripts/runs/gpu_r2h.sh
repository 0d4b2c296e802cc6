cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_shard_driver.py -q -m gpu -n 12 > $O/shard.log 2>&1; echo "shard rc=$?" >> $O/shard.log) &
(timeout 900 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_synth.py -q -m gpu -n 12 > $O/gold.log 2>&1; echo "gold rc=$?" >> $O/gold.log) &
wait
tail -n 12 $O/shard.log; tail -n 3 $O/gold.log
for sub in 16 32; do
  STA_DEPTH_SUB=$sub timeout 300 python bench.py --workload depth30 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline > $O/bench_d30_sub$sub.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/bench_d30_sub$sub.json').read()); print('depth sub $sub', round(d['ms_per_step'],3), {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())[:3]})"
done
STA_BENCH_BACKEND=gloo STA_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --cols 1048576 --verify --no-pmc > $O/bench_n2.json 2> $O/bench_n2.err; echo "bench n2 rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_n2.json').read()); print('n2', d['value'], d['verify'])"
cd /tmp
for wl in mpileup30 mpileup30_B depth30 mpileup300; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 1 --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1
  head -8 $GRAFT_REPO_ROOT/$O/prof_$wl/p_kernel_stats.csv 2>/dev/null | cut -c1-140
done
