cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n2
for wl in mpileup30 mpileup300; do
  STA_BENCH_ONE_DEVICE=1 STA_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl $( [ $wl = mpileup30 ] && echo --verify ) > gpurun_out/r03n2/bench_n2_$wl.json 2> gpurun_out/r03n2/bench_n2_$wl.err
  echo "$wl rc=$?"; tail -1 gpurun_out/r03n2/bench_n2_$wl.json | python -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print(round(d["value"]), d["n_gpus"], round(d["ms_per_step"],3), d.get("verify"), json.dumps(d.get("per_rank"))[:500])
except Exception as e: print("nojson", e)'
  tail -3 gpurun_out/r03n2/bench_n2_$wl.err | cut -c1-300
done
# NCCL (RCCL) with both ranks on the one device: expected to be refused by RCCL (duplicate GPU); shown for the record
STA_BENCH_ONE_DEVICE=1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 2 --warmup 1 --workload mpileup30_B > gpurun_out/r03n2/bench_n2_rccl.json 2> gpurun_out/r03n2/bench_n2_rccl.err; echo "rccl one-device rc=$?"; tail -2 gpurun_out/r03n2/bench_n2_rccl.err | cut -c1-300
