cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2u; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu -n 16 -k "depth or d1_ or d2_ or d3_ or d4_ or d5_ or d6_ or d7_ or d8_ or large_pos" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 4 $O/pytest.log
run() { python bench.py --workload depth30 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['kernels_ms_per_step'].get('depth_fused'))"; }
run default
STA_DEPTH_LBUF=8192 run lbuf8192
STA_DEPTH_LBUF=2048 run lbuf2048
STA_DEPTH_TICKET=0 run noticket
STA_DEPTH_TICKET=0 STA_DEPTH_LBUF=2048 run noticket_lbuf2048
STA_DEPTH_TICKET=0 STA_DEPTH_LBUF=8192 run noticket_lbuf8192
