cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { python bench.py --workload mpileup30 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k: round(v,3) for k,v in list(d['kernels_ms_per_step'].items())[:4]})"; }
run nontemporal
STA_BAQ_PLAIN_MEM=1 run plain
run nontemporal2
