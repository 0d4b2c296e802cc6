#!/bin/bash
# Round 5, GPU session M: where k_depth_fused's 0.218 ms goes (diagnostic builds with wrong text: no EMIT / no look-back wait / no COUNT marks), ticket off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
run() { env $1 python bench.py --steps 20 --warmup 5 --workload depth30 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k: round(x,4) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"; }
for v in STA_DEPTH_DIAG=0 STA_DEPTH_TICKET=0 STA_DEPTH_DIAG=1 STA_DEPTH_DIAG=2 STA_DEPTH_DIAG=3 STA_DEPTH_DIAG=0; do run $v; done 2>&1 | tee $O/depth_diag.log
true
