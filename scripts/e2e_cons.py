"""End-to-end CLI timing of `consensus` (BAM file -> FASTQ text): samtools-amd vs the CPU oracle on synthetic 30x 150 bp reads.
Host decode, staging, PCIe and the writer are included -- this is NOT bench.py's HBM-resident number.

    python scripts/e2e_cons.py [contigs=2] [columns_per_contig=4375000] [outdir=/dev/shm/sta_e2e_cons]

Uses scripts/e2e_big.py's generator (no MD tags on the generated reads: the Bayesian mode then skips the mismatch costs)."""
import hashlib
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_contigs = sys.argv[1] if len(sys.argv) > 1 else "2"
cols = sys.argv[2] if len(sys.argv) > 2 else "4375000"
out = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm/sta_e2e_cons"
ENG = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
ORA = os.path.join(REPO, "oracle", "_build", "oracle_samtools")


def timed(cmd, env=None):
    t0 = time.perf_counter()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    h = hashlib.sha256(); n = 0
    while True:
        b = p.stdout.read(1 << 22)
        if not b:
            break
        h.update(b); n += len(b)
    err = p.stderr.read().decode()
    if p.wait() != 0:
        raise SystemExit("%s failed: %s" % (" ".join(cmd[:3]), err[-400:]))
    return time.perf_counter() - t0, h.hexdigest(), n


def main():
    bam = os.path.join(out, "big.bam")
    if not os.path.exists(bam):
        # e2e_big.py builds the input (and then runs its own commands; only the files are wanted here)
        env = dict(os.environ, E2E_THREADS="8")
        subprocess.run([sys.executable, os.path.join(REPO, "scripts", "e2e_big.py"), n_contigs, cols, out], env=env, stdout=subprocess.DEVNULL, check=True)
    n_reads = int(open(os.path.join(out, "n_reads")).read())
    mb = n_reads * 150 / 1e6
    print("input: %s contigs x %s columns, %d reads = %.0f Mbases piled, host cores %d" % (n_contigs, cols, n_reads, mb, os.cpu_count()))
    for name, args in (("consensus -m simple -f fastq", ["consensus", "-m", "simple", "-f", "fastq", bam]),
                       ("consensus -f fastq (Bayesian)", ["consensus", "-f", "fastq", bam]),
                       ("consensus -f pileup (Bayesian)", ["consensus", "-f", "pileup", bam])):
        best = None
        for thr in ("8", "16"):
            dt, sha, n = timed([ENG] + args, env=dict(os.environ, STA_IO_THREADS=thr))
            if best is None or dt < best[0]:
                best = (dt, sha, n, thr)
        odt, osha, on = timed([ORA] + args)
        print("%-32s engine %.2f s = %.0f Mbases/s (io_threads=%s)   oracle %.2f s = %.1f Mbases/s   x%.1f   %d bytes: %s"
              % (name, best[0], mb / best[0], best[3], odt, mb / odt, odt / best[0], n, "IDENTICAL" if (best[1], best[2]) == (osha, on) else "DIFFERENT"))


if __name__ == "__main__":
    main()
