cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -n 16 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log)
tail -n 4 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -n 2 $O/bench_default.err
for w in mpileup30_B depth30 consensus30 consensus30_simple mpileup300; do
timeout 400 python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err; done
python - <<PY
import json
for w in ("default","mpileup30_B","depth30","consensus30","consensus30_simple","mpileup300"):
    try:
        d=json.loads(open("$O/bench_%s.json"%w).read()); print(w, round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in list(d["kernels_ms_per_step"].items())[:6]}, "frac", round(d["roofline"]["frac"],4), "parity", (d.get("parity_check") or {}).get("identical"))
    except Exception as e: print(w, "FAILED", e)
PY
for w in mpileup30 depth30; do
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > /dev/null 2>> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; find $O/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -5 {}'
done
