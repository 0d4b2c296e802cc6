"""Where an end-to-end `samtools-amd mpileup` run spends its time (STA_DRIVER_TIMING=1), SAM and BAM input, both input lanes."""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth import write_synth_sam
from bamio import sam_to_bam
out = "/tmp/e2e"; os.makedirs(out, exist_ok=True)
n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
sam, fa = write_synth_sam(out, n_ref=n_ref, depth=30, read_len=150, seed=5, paired=False)
bam = sam_to_bam(sam, os.path.join(out, "synth.bam"), level=1)
for args in (["mpileup", "-B", "-f", fa], ["mpileup", "-f", fa]):
    for path in (sam, bam):
        for lane in ("chunk", "rec"):
            for pin in ("1", "0"):
                env = dict(os.environ, STA_DRIVER_TIMING="1", STA_IO_THREADS="4")
                if lane == "rec": env["STA_IO_LANE"] = "rec"
                if pin == "0": env["STA_NO_PINNED"] = "1"
                t0 = time.perf_counter()
                p = subprocess.run(["samtools_amd/bin/samtools-amd"] + args + [path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
                dt = time.perf_counter() - t0
                tl = [l for l in p.stderr.decode().split("\n") if "driver timing" in l]
                print(" ".join(args[:2]), os.path.basename(path), lane, "pinned" if pin == "1" else "pageable", "%.3f s total" % dt, tl[0] if tl else "")
