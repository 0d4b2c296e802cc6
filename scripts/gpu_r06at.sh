#!/bin/bash
# Round 6, GPU session AT: a third batch of parity hunts on the device build, other seeds (hunt4's larger inputs among them).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06at; mkdir -p $O
timeout 900 python scripts/hunt9.py 940 950 > $O/hunt9.log 2>&1; tail -1 $O/hunt9.log
HUNT8_CASES=40 timeout 1200 python scripts/hunt8.py 9821 9822 > $O/hunt8.log 2>&1; tail -1 $O/hunt8.log
HUNT6_CASES=40 timeout 1200 python scripts/hunt6.py 9621 9622 > $O/hunt6.log 2>&1; tail -1 $O/hunt6.log
HUNT5_CASES=40 timeout 1200 python scripts/hunt5.py 9521 9522 > $O/hunt5.log 2>&1; tail -1 $O/hunt5.log
timeout 1200 python scripts/hunt4.py 9421 > $O/hunt4.log 2>&1; tail -2 $O/hunt4.log
