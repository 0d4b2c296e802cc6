#!/bin/bash
# Round 6, GPU session AG: the parity hunts on the DEVICE build (they usually run on the CPU emulation, which runs workgroups one after the other):
# hunt9 (rich-CIGAR mate overlaps) 8 seeds, hunt8 / hunt6 / hunt5 one fresh seed each.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ag; mkdir -p $O
timeout 600 python scripts/hunt9.py 900 908 > $O/hunt9.log 2>&1; tail -2 $O/hunt9.log
HUNT8_CASES=40 timeout 900 python scripts/hunt8.py 9801 > $O/hunt8.log 2>&1; tail -1 $O/hunt8.log
HUNT6_CASES=40 timeout 900 python scripts/hunt6.py 9601 > $O/hunt6.log 2>&1; tail -1 $O/hunt6.log
HUNT5_CASES=40 timeout 900 python scripts/hunt5.py 9501 > $O/hunt5.log 2>&1; tail -1 $O/hunt5.log
grep -h -v "^ok\|^skip\|^seed" $O/hunt*.log | head -20
