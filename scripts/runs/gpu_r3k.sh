cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
timeout 300 python bench.py --workload mpileup300_B --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_mpileup300_B_pmc.json 2> $O/err.log; python -c "
import json; d=json.loads(open('$O/bench_mpileup300_B_pmc.json').read()); print(d['roofline'])"
bash scripts/gpu_sq.sh mpileup300_B r3k/sq 2>&1 | grep -i "deep\|len_fast"
rm -rf gpurun_out/r3k/sq/p1 gpurun_out/r3k/sq/p2
