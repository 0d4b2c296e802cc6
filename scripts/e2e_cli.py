"""End-to-end CLI timing (file -> text) of samtools-amd vs the CPU oracle on one synthetic SAM: the host side
(single-thread SAM decode, staging, PCIe, fwrite) is included, unlike bench.py's HBM-resident number."""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth import write_synth_sam
out = "gpurun_out/e2e"; os.makedirs(out, exist_ok=True)
n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sam, fa = write_synth_sam(out, n_ref=n_ref, depth=30, read_len=150, seed=5, paired=False)
n_reads = 30 * n_ref // 150
for name, args in (("mpileup -f", ["mpileup", "-f", fa, sam]), ("mpileup -B -f", ["mpileup", "-B", "-f", fa, sam]), ("depth -a", ["depth", "-a", sam])):
    res = {}
    for who, exe in (("oracle", "oracle/_build/oracle_samtools"), ("engine", "samtools_amd/bin/samtools-amd")):
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                subprocess.run([exe] + args, stdout=dn, stderr=dn, check=True)
            best = min(best, time.perf_counter() - t0)
        res[who] = best
    mb = n_reads * 150 / 1e6
    print("%-14s %d reads: oracle %.2f s (%.1f Mbases/s)  engine %.2f s (%.1f Mbases/s)  x%.1f" % (
        name, n_reads, res["oracle"], mb / res["oracle"], res["engine"], mb / res["engine"], res["oracle"] / res["engine"]))
