#!/bin/bash
# Round 6, GPU session AE: kernel + copy timeline of the 16 M-column mpileup30 step (where are the 3 ms outside k_baq7s?) and the rocprofv3 kernel
# stats of the default workload for profiles/.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r06ae; mkdir -p $R/$O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-e2e --workload mpileup30 > $R/$O/prof.log 2>&1
cd $R; D=$(dirname $(ls $O/prof/*/*kernel_trace.csv $O/prof/*kernel_trace.csv 2>/dev/null | head -1)); python scripts/step_timeline.py $D | tee $O/timeline.txt
rm -f $D/*memory_copy_trace.csv.bak
