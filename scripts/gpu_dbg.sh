cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
( timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py tests/test_gpu_plp_api.py -m gpu -q -x -o timeout=150 2>&1 | tail -4 | cut -c1-300 )
python - <<'PY'
import sys, os, subprocess, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
inp = bench.synth_inputs("mpileup30_B_hotspot", 4 << 20)
from bamio import sam_to_bam
bam = sam_to_bam(inp["sam"], inp["dir"] + "/s.bam")
for args in (["mpileup", "-B", "-f", inp["fa"], bam],):
    t = time.perf_counter(); p = subprocess.run(["samtools_amd/bin/samtools-amd"] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, STA_DRIVER_TIMING="1", STA_WINDOW_COLS=str(4 << 20))); dt = time.perf_counter() - t
    print("default -d 8000 on the hotspot input (one 4 M-column window):", round(dt, 2), "s", p.stderr.decode().strip().split("\n")[-1][:200])
PY
