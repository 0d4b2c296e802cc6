#!/bin/bash
# file -> text timings of the 1-Gbase input under a list of environment settings (one per line of $SETS, ';' separated)
cd $GRAFT_REPO_ROOT
IFS=';' read -ra S <<< "$SETS"
for e in "${S[@]}"; do
  echo "== $e"
  env $e E2E_NO_ORACLE=1 E2E_CMD_TIMEOUT=60 E2E_THREADS="${THR:-16/4}" timeout 300 python scripts/e2e_big.py 2>&1 | grep "io_threads" | cut -c1-330
done
