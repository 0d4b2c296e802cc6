cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
for rep in 1 2; do for np in "" 1; do
  STA_BENCH_NOPROF=$np python bench.py --steps 20 --warmup 3 --workload mpileup30 --no-cpu-baseline --no-pmc --no-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpileup30 noprof=$np', round(d['ms_per_step'],3))"
done; done 2>&1 | tee $O/noprof.log
for np in "" 1; do
  STA_BENCH_NOPROF=$np python bench.py --steps 20 --warmup 3 --workload mpileup30_B --no-cpu-baseline --no-pmc --no-e2e 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpileup30_B noprof=$np', round(d['ms_per_step'],3))"
done 2>&1 | tee -a $O/noprof.log
