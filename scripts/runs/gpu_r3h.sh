cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
(STA_EMIT_DEEP=1 timeout 900 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_synth.py tests/test_gpu_shard_driver.py -q -m gpu -n 16 > $O/pytest_deep.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deep.log)
tail -n 12 $O/pytest_deep.log
run() { python bench.py --workload $2 --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if 'mplp' in k})"; }
run auto mpileup300_B
STA_EMIT_DEEP=0 run lanes mpileup300_B
STA_EMIT_DEEP=1 run deep mpileup30_B
STA_EMIT_DEEP=0 run lanes mpileup30_B
