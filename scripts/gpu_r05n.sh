#!/bin/bash
# Round 5, GPU session N: k_depth_fused with the workgroup-wide look-back (2 048 predecessors per round trip) against the wave look-back; depth tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
( time timeout 800 python -m pytest tests -m gpu -q -o timeout=300 -p no:cacheprovider -k "depth" ) > $O/pytest_depth.log 2>&1; tail -4 $O/pytest_depth.log
run() { env $1 python bench.py --steps 20 --warmup 5 --workload depth30 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k: round(x,4) for k,x in list(d['kernels_ms_per_step'].items())[:4]}, d['output_sha256'][:12])"; }
for v in STA_DEPTH_LOOKBACK=wave STA_DEPTH_LOOKBACK=block STA_DEPTH_LOOKBACK=wave STA_DEPTH_LOOKBACK=block "STA_DEPTH_LOOKBACK=block STA_DEPTH_TICKET=0"; do run "$v"; done 2>&1 | tee $O/depth_ab.log
true
