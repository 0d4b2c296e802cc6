#!/bin/bash
# round 4, GPU session J: file -> text with device staging on / off (1.05 Gbases), the default bench line with its e2e object
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
( E2E_THREADS=16/4 timeout 900 python scripts/e2e_big.py 8 4375000 > $O/e2e_stage_device.log 2>&1 ) ; grep -E "best|input|io_threads|parity|IDENT|DIFF" $O/e2e_stage_device.log | cut -c1-250
( STA_STAGE_DEVICE=0 E2E_THREADS=16/4 timeout 600 python scripts/e2e_big.py 8 4375000 > $O/e2e_stage_host.log 2>&1 ) ; grep -E "best|io_threads" $O/e2e_stage_host.log | cut -c1-250
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], json.dumps(d.get("e2e")), (d.get("parity_check") or {}).get("identical"))'; tail -3 $O/bench_default.err
