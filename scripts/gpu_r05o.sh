#!/bin/bash
# Round 5, GPU session O: the workgroup-wide look-back with 256 / 512 predecessors per round trip (K = 1, 2) -- K = 8 was slower than the wave version
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
run() { cp $1 samtools_amd/lib/libsamtools_amd.so; env $2 python bench.py --steps 20 --warmup 5 --workload depth30 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['ms_per_step'],4), {k: round(x,4) for k,x in list(d['kernels_ms_per_step'].items())[:4]}, d['output_sha256'][:12])"; }
for rep in 1 2; do run /tmp/lib_keep.so STA_DEPTH_LOOKBACK=wave; run samtools_amd/lib/libk1.so STA_DEPTH_LOOKBACK=block; run samtools_amd/lib/libk2.so STA_DEPTH_LOOKBACK=block; done 2>&1 | tee $O/depth_ab.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
true
