#!/bin/bash
# Round 5, GPU session B: the whole -m gpu suite on the new code, A/B of the band-word forms inside one box, the shader clock under k_baq7s
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
cp samtools_amd/lib/libsamtools_amd.so /tmp/lib_keep.so
( time timeout 1200 python -m pytest tests -m gpu -q -x -o timeout=600 -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
run() { # lib workload
  cp samtools_amd/lib/lib$1.so samtools_amd/lib/libsamtools_amd.so
  python bench.py --steps 10 --warmup 3 --workload $2 --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:4]})"
}
for rep in 1 2 3; do for v in prev threebit twobit; do run $v mpileup30; done; done 2>&1 | tee $O/ab_baq.log
for v in prev twobit; do run $v mpileup30_indel; run $v mpileup300; done 2>&1 | tee -a $O/ab_baq.log
cp /tmp/lib_keep.so samtools_amd/lib/libsamtools_amd.so
# shader clock: GRBM_GUI_ACTIVE and SQ_BUSY_CYCLES of the kernel against its duration in the same pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/$O/clk -o x -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-e2e --workload mpileup30 > $R/$O/clk.log 2>&1
python - <<PY
import csv, glob, collections
R="$R/$O/clk"
dur = {}
for f in glob.glob(R + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(R + "/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        a = agg[k][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for k in agg:
    if "k_baq7s" in k or "emit_tile" in k:
        c = {n: v[1] / v[0] for n, v in agg[k].items()}
        ms = sum(dur[k]) / len(dur[k])
        print(k, "avg ms", round(ms, 3), {n: round(x) for n, x in sorted(c.items())})
        if "GRBM_GUI_ACTIVE" in c: print("   GRBM_GUI_ACTIVE / duration = %.3f GHz" % (c["GRBM_GUI_ACTIVE"] / ms / 1e6))
PY
true
