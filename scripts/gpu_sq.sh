# SQ counters per kernel: usage: bash scripts/gpu_sq.sh <workload> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; WL=${1:-mpileup30_B}; TAG=${2:-sq}
mkdir -p $R/gpurun_out/$TAG
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/p$i -o x -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --workload $WL > $R/gpurun_out/$TAG/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
R="$R"; tag="$TAG"
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0,0.0]))
for f in glob.glob("%s/gpurun_out/%s/p*/*counter_collection.csv" % (R, tag)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        a = agg[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k in agg:
    if "k_" in k and "scan" not in k:
        print(k, {c: round(v[1]/v[0]) for c, v in sorted(agg[k].items())})
PY
