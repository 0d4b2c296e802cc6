"""End-to-end CLI timing (file -> text) of samtools-amd vs the CPU oracle on one synthetic input, SAM text and BAM:
the host side (decode, staging, PCIe, fwrite) is included, unlike bench.py's HBM-resident number.
usage: python scripts/e2e_cli.py [n_ref_columns]"""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth import write_synth_sam
from bamio import sam_to_bam
out = "/tmp/e2e"; os.makedirs(out, exist_ok=True)      # (large files: not under gpurun_out/, which is copied back)
n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
sam, fa = write_synth_sam(out, n_ref=n_ref, depth=30, read_len=150, seed=5, paired=False)
bam = sam_to_bam(sam, os.path.join(out, "synth.bam"), level=1)
n_reads = 30 * n_ref // 150
mb = n_reads * 150 / 1e6


def best_of(cmd, env=None, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as dn:
            subprocess.run(cmd, stdout=dn, stderr=dn, check=True, env=env)
        best = min(best, time.perf_counter() - t0)
    return best


# process start + HIP context creation, paid once per run whatever the input size
t_start = best_of(["samtools_amd/bin/samtools-amd", "depth", os.path.join("tests", "golden", "mpileup", "mp_D.sam")])
print("engine start-up (tiny input): %.2f s" % t_start)
for name, args in (("mpileup -f", ["mpileup", "-f", fa]), ("mpileup -B -f", ["mpileup", "-B", "-f", fa]), ("depth -a", ["depth", "-a"])):
    t_or = best_of(["oracle/_build/oracle_samtools"] + args + [sam], reps=1)
    line = "%-14s %d reads (%.0f Mbases): oracle %.2f s (%.1f Mb/s)" % (name, n_reads, mb, t_or, mb / t_or)
    for label, path, thr in (("sam", sam, "4"), ("bam io=1", bam, "1"), ("bam io=4", bam, "4")):
        t = best_of(["samtools_amd/bin/samtools-amd"] + args + [path], env=dict(os.environ, STA_IO_THREADS=thr))
        line += " | %s %.2f s (%.0f Mb/s, %.0f net of start-up)" % (label, t, mb / t, mb / max(t - t_start, 1e-3))
    print(line)
