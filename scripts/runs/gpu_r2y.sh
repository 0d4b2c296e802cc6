cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2y; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_synth.py -q -m gpu -n 16 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 3 $O/pytest.log
for w in mpileup30_B mpileup300_B; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['kernels_ms_per_step'])"; done
