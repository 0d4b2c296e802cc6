cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py -q -m gpu -n 16 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 3 $O/pytest.log
run() { python bench.py --workload mpileup30 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())[:4]})"; }
run prefetch
STA_BAQ_PREFETCH=0 run noprefetch
