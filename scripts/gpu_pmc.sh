# HBM traffic per kernel from the PMC counters (separate passes, MI355X_MICROARCH.md "HBM"): usage: TAG=r01 bash scripts/gpu_pmc.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; WL=${1:-mpileup30}; TAG=${TAG:-r01}
mkdir -p $R/gpurun_out/pmc_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$c -o $WL -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $R/gpurun_out/pmc_$TAG/$c.log 2>&1
  ls $R/gpurun_out/pmc_$TAG/$c | head
done
python - <<PY
import csv, glob, collections, json, os
R = os.environ.get("GRAFT_REPO_ROOT"); tag = "$TAG"; wl = "$WL"
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("%s/gpurun_out/pmc_%s/%s/*counter_collection.csv" % (R, tag, c))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c: continue
            k = row["Kernel_Name"].split("(")[0]
            agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[c] = {"launches": n, "sum": v, "per_launch": v / max(n, 1)}
json.dump(out, open("%s/gpurun_out/pmc_%s/%s_traffic_raw.json" % (R, tag, wl), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -sum(x["sum"] for x in kv[1].values()))[:12]:
    print(k, {c: round(x["per_launch"]) for c, x in v.items()})
PY
