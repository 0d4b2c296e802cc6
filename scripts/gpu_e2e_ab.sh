#!/bin/bash
# file -> text runs of the 1-Gbase input with the device's BGZF decoder off / on (timings only), then byte parity with the oracle
cd $GRAFT_REPO_ROOT
for g in ${GI:-0 1}; do
  echo "== STA_GPU_INFLATE=$g"
  STA_GPU_INFLATE=$g E2E_NO_ORACLE=1 E2E_THREADS="${THR:-16/4}" timeout 300 python scripts/e2e_big.py 2>&1 | grep -v "^input\|^engine start" | cut -c1-330
done
if [ -n "$PARITY" ]; then E2E_THREADS="16/4" timeout 600 python scripts/e2e_big.py 2>&1 | grep "parity\|oracle" ; fi
