"""The `e2e` object of bench.py's default run, taken apart: the same input (the CPU-baseline sample of 2 M columns at 30x replicated onto eight
contigs, BAM level 1) through `samtools-amd depth -a / mpileup -B -f / mpileup -f` with STA_DRIVER_TIMING=1, so that the wall time of each
command is split into process start, the window pipeline's phases and what is left (exit: freeing page-locked pools, closing the runtime).
    python scripts/e2e_bench_shape.py [sample_cols] [copies]"""
import os, subprocess, sys, time, shutil
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from samtools_amd import _capi
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inp = bench.synth_inputs("mpileup30", cols)
d = inp["dir"]
names = ["chrS%d" % k for k in range(copies)]
body = open(inp["sam"]).read()
reads = body[body.index("\n", body.index("@SQ")) + 1:]
big = os.path.join(d, "e2e.sam")
with open(big, "w") as fh:
    fh.write("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, cols) for nm in names))
    for nm in names:
        fh.write(reads.replace("\tchrS\t", "\t%s\t" % nm))
fa_txt = open(inp["fa"]).read()
big_fa = os.path.join(d, "e2e.fa")
with open(big_fa, "w") as fh:
    for nm in names:
        fh.write(fa_txt.replace(">chrS\n", ">%s\n" % nm, 1))
bam = os.path.join(d, "e2e.bam")
_capi.io_write_bam(big, bam, 1)
exe = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
mb = reads.count("\n") * copies * 150 / 1e6
print("input: %d contigs x %d columns, %.0f Mbases, BAM %.0f MB" % (copies, cols, mb, os.path.getsize(bam) / 1e6))
# as a user runs it (no timing lines: the process ends as soon as its text is out, driver_capture.cpp driver_exit_now_if_asked), best of three
big = os.environ.get("STA_E2E_BIG") is not None      # a multi-Gbase input: page-locked staging (the default there) against plain memory, nothing else
for args in (["depth", "-a", bam], ["mpileup", "-B", "-f", big_fa, bam], ["mpileup", "-f", big_fa, bam]):
    variants = ({}, {"STA_PIN": "0"}, {"STA_PIN": "0", "STA_PIPE_SLOTS": "3"}) if big else ({}, {"STA_NO_FAST_EXIT": "1"})
    if os.environ.get("STA_E2E_VARIANTS"):        # ad-hoc A/B: ';'-separated K=V,K=V sets ("-" = nothing set)
        variants = tuple(dict(kv.split("=", 1) for kv in v.split(",") if "=" in kv) for v in os.environ["STA_E2E_VARIANTS"].split(";"))
    for env_extra in variants:
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            p = subprocess.run([exe] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env_extra))
            ts.append(time.perf_counter() - t0)
        print("plain run:", " ".join(args[:3])[:24], env_extra or "", " ".join("%.3f" % t for t in ts), "s wall; best = %.0f Mbases/s" % (mb / min(ts)), "rc", p.returncode)
for env_extra in (({},) if big else ({}, {"STA_FAST_EXIT": "1"}, {"STA_GPU_INFLATE": "1"})):
    for args in (["depth", "-a", bam], ["mpileup", "-B", "-f", big_fa, bam], ["mpileup", "-f", big_fa, bam]):
        for rep in range(2):
            env = dict(os.environ, STA_DRIVER_TIMING=os.environ.get("STA_E2E_TIMING", "1"), **env_extra)
            t0 = time.perf_counter()
            p = subprocess.run([exe] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t0
            tl = [l for l in p.stderr.decode().split("\n") if "driver timing" in l or "driver threads" in l]
            print(" ".join(args[:3])[:24], env_extra or "", "%.3f s wall = %.0f Mbases/s |" % (dt, mb / dt), " ".join(tl)[:900])
            if rep == 1:
                for l in p.stderr.decode().split("\n"):
                    if l.startswith("[timeline]") or l.startswith("[window"): print("     ", l)
shutil.rmtree(d, ignore_errors=True)
