"""End-to-end CLI timing (BAM file -> text) on >= 1 Gbase of synthetic 30x 150 bp reads (SURVEY.md 8d config 3 shape:
several contigs), samtools-amd vs the CPU oracle.  Host decode, staging, PCIe and the output stream are all included --
this is NOT bench.py's HBM-resident number.

    python scripts/e2e_big.py [contigs=8] [columns_per_contig=4375000] [outdir=/dev/shm/sta_e2e]
    E2E_DEPTH=1.5 E2E_GENOME=1 python scripts/e2e_big.py 24 100000000      # genome-sized: 24 contigs, 2.4 Gbp (> 2^31 linear columns)
E2E_GENOME=1: one thread setting, the oracle only for `depth -a` of the whole file and `mpileup -f` of the last 2 Mbp of the last
contig (xxhash of the streams), plus -- E2E_SHARDS=N -- the N blocks of a sharded `depth -a` (STA_SHARD=r/N, run one after the
other) whose concatenation must hash like the unsharded text.

The input is generated in parallel (one process per contig), BGZF level 1.  Output goes to /dev/null for the timings; one
extra run per command writes to a file and is compared (sha256) with the oracle's text (`mpileup -f` on one contig only:
the oracle needs ~20 s per 125 Mbases there).  STA_DRIVER_TIMING=1 gives the per-thread phase table of the window pipeline."""
import hashlib
import multiprocessing as mp
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4375000
out = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm/sta_e2e"
os.makedirs(out, exist_ok=True)
ENG = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
ORA = os.path.join(REPO, "oracle", "_build", "oracle_samtools")


def make_contig(i):
    import numpy as np
    from synth import synth_ref, synth_reads, cigar_str
    from bamio import bam_record_bytes, bgzf_compress
    name = "chr%d" % (i + 1)
    ref = synth_ref(cols, seed=1 + i)
    rd = synth_reads(ref, depth=float(os.environ.get("E2E_DEPTH", "30")), read_len=150, seed=42 + i)
    tid = {"chr%d" % (k + 1): k for k in range(n_contigs)}
    names = rd["names"].tobytes().split(b"\0")
    recs = []
    for r in range(rd["n"]):
        line = "%s_%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s" % (
            names[r].decode(), i, int(rd["flag"][r]), name, int(rd["_abs_pos"][r]) + 1, int(rd["mapq"][r]), cigar_str(rd, r),
            rd["_bases"][r].tobytes().decode(), (rd["_quals"][r] + 33).astype(np.uint8).tobytes().decode())
        recs.append(bam_record_bytes(line, tid))
    comp = bgzf_compress(b"".join(recs), level=1)
    part = os.path.join(out, "part%d.bgzf" % i)
    with open(part, "wb") as fh:
        fh.write(comp)
    s = ref.tobytes().decode()
    fa = ">%s\n" % name + "\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + "\n"
    with open(os.path.join(out, "part%d.fa" % i), "w") as fh:
        fh.write(fa)
    return int(rd["n"])


def timed(cmd, env=None, stdout=None):
    t0 = time.perf_counter()
    with open(os.devnull, "wb") as dn:
        p = subprocess.run(cmd, stdout=stdout or dn, stderr=subprocess.PIPE, env=env, timeout=float(os.environ.get("E2E_CMD_TIMEOUT", "600")))
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise SystemExit("%s failed: %s" % (" ".join(cmd[:3]), p.stderr.decode()[-400:]))
    return dt, p.stderr.decode()


def sha_of(cmd, env=None, h=None):
    """(hex digest, bytes) of the command's stdout; h: a running hash object to continue (sharded blocks)"""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
    if h is None:
        h = hashlib.sha256()
    n = 0
    while True:
        b = p.stdout.read(1 << 22)
        if not b:
            break
        h.update(b); n += len(b)
    if p.wait() != 0:
        raise SystemExit("failed: " + " ".join(cmd[:3]))
    return h.hexdigest(), n


def main():
    from bamio import bam_header_bytes, bgzf_compress, _EOF
    bam, fa = os.path.join(out, "big.bam"), os.path.join(out, "big.fa")
    t0 = time.perf_counter()
    if not os.path.exists(bam):
        with mp.Pool(min(n_contigs, 24)) as pool:
            counts = pool.map(make_contig, range(n_contigs))
        names = ["chr%d" % (k + 1) for k in range(n_contigs)]
        hdr = ["@HD\tVN:1.6\tSO:coordinate"] + ["@SQ\tSN:%s\tLN:%d" % (n, cols) for n in names]
        with open(bam, "wb") as fo:
            fo.write(bgzf_compress(bam_header_bytes(hdr, names, [cols] * n_contigs)))
            for i in range(n_contigs):
                fo.write(open(os.path.join(out, "part%d.bgzf" % i), "rb").read())
                os.unlink(os.path.join(out, "part%d.bgzf" % i))
            fo.write(_EOF)
        with open(fa, "w") as fo:
            for i in range(n_contigs):
                fo.write(open(os.path.join(out, "part%d.fa" % i)).read())
                os.unlink(os.path.join(out, "part%d.fa" % i))
        n_reads = sum(counts)
        open(os.path.join(out, "n_reads"), "w").write(str(n_reads))
    n_reads = int(open(os.path.join(out, "n_reads")).read())
    mb = n_reads * 150 / 1e6
    print("input: %d contigs x %d columns, %d reads = %.0f Mbases piled, BAM %.0f MB (generated in %.0f s), host cores %d"
          % (n_contigs, cols, n_reads, mb, os.path.getsize(bam) / 1e6, time.perf_counter() - t0, os.cpu_count()))
    t_start, _ = timed([ENG, "depth", os.path.join(REPO, "tests", "golden", "mpileup", "mp_D.sam")])
    print("engine start-up (tiny input): %.2f s" % t_start)
    cmds = [("depth -a", ["depth", "-a", bam]), ("mpileup -B -f", ["mpileup", "-B", "-f", fa, bam]), ("mpileup -f", ["mpileup", "-f", fa, bam])]
    for name, args in ([] if os.environ.get("E2E_ONLY_SHARDS") else cmds):
        best, best_err = 1e9, ""
        # E2E_THREADS: decode threads per input, optionally "decode/stage" (STA_STAGE_THREADS: threads copying a window's slices)
        for thr in (os.environ.get("E2E_THREADS", "16/4" if os.environ.get("E2E_GENOME") else "8,16,24").split(",")):
            env = dict(os.environ, STA_IO_THREADS=thr.split("/")[0], STA_DRIVER_TIMING="1")
            if "/" in thr:
                env["STA_STAGE_THREADS"] = thr.split("/")[1]
            dt, err = timed([ENG] + args, env=env)
            line = [l for l in err.split("\n") if l.startswith("[driver timing]")]
            print("  %-14s io_threads=%-5s %.2f s  %.0f Mbases/s   %s" % (name, thr, dt, mb / dt, line[0] if line else ""))
            if dt < best:
                best, best_err = dt, thr
        print("%-14s best %.2f s = %.0f Mbases/s (io_threads=%s; %.0f net of start-up)" % (name, best, mb / best, best_err, mb / max(best - t_start, 1e-3)))
    if os.environ.get("E2E_GENOME"):
        import xxhash
    if os.environ.get("E2E_GENOME") and not os.environ.get("E2E_ONLY_SHARDS") and not os.environ.get("E2E_TIMING_ONLY"):
        t1 = time.perf_counter()
        a, na = sha_of([ENG, "depth", "-a", bam], h=xxhash.xxh3_128()); t2 = time.perf_counter()
        b, nb = sha_of([ORA, "depth", "-a", bam], h=xxhash.xxh3_128()); t3 = time.perf_counter()
        print("parity depth -a (whole file) engine %d bytes in %.1f s, oracle %d bytes in %.1f s: %s" % (na, t2 - t1, nb, t3 - t2, "IDENTICAL" if a == b else "DIFFERENT"))
        reg = "chr%d:%d-%d" % (n_contigs, cols - 2000000 + 1, cols)
        for args in (["mpileup", "-f", fa, "-r", reg, bam], ["mpileup", "-B", "-a", "-f", fa, "-r", "chr%d:1-3000000" % (n_contigs // 2), bam]):
            a, na = sha_of([ENG] + args, h=xxhash.xxh3_128()); b, nb = sha_of([ORA] + args, h=xxhash.xxh3_128())
            print("parity %s engine %d bytes, oracle %d bytes: %s" % (" ".join(args[:-1]).replace(fa, "ref.fa"), na, nb, "IDENTICAL" if a == b else "DIFFERENT"))
    if os.environ.get("E2E_GENOME"):
        ns = int(os.environ.get("E2E_SHARDS", "0"))
        if ns:
            whole, nw = sha_of([ENG, "depth", "-aa", bam], h=xxhash.xxh3_128())          # (a sharded run refuses single -a: whether a contig prints depends on every block)
            h = xxhash.xxh3_128(); tot = 0; times = []
            for r in range(ns):
                t1 = time.perf_counter()
                _, nb_ = sha_of([ENG, "depth", "-aa", bam], env=dict(os.environ, STA_SHARD="%d/%d" % (r, ns)), h=h)
                times.append(time.perf_counter() - t1); tot += nb_
            print("sharded depth -aa, %d blocks one after the other (%d linear columns): %d bytes vs %d unsharded: %s; seconds per block: %s"
                  % (ns, n_contigs * cols, tot, nw, "IDENTICAL" if h.hexdigest() == whole else "DIFFERENT", " ".join("%.1f" % x for x in times)))
        return
    if os.environ.get("E2E_NO_ORACLE"):          # timings only (thread sweeps)
        return
    if os.environ.get("E2E_QUICK"):
        # a short run (GPU minutes): the oracle only for depth -a; mpileup text compared between one and several staging threads
        a, na = sha_of([ENG, "depth", "-a", bam]); b, nb = sha_of([ORA, "depth", "-a", bam])
        print("parity depth -a engine %d bytes, oracle %d bytes: %s" % (na, nb, "IDENTICAL" if a == b else "DIFFERENT"))
        a, na = sha_of([ENG, "mpileup", "-B", "-f", fa, bam], env=dict(os.environ, STA_STAGE_THREADS="1"))
        b, nb = sha_of([ENG, "mpileup", "-B", "-f", fa, bam], env=dict(os.environ, STA_STAGE_THREADS="6"))
        print("mpileup -B -f  1 vs 6 staging threads: %d / %d bytes: %s" % (na, nb, "IDENTICAL" if a == b else "DIFFERENT"))
        return
    # the oracle on the same file (single thread); mpileup -f only on the first contig (x n_contigs = the whole file)
    for name, args, scale in (("depth -a", ["depth", "-a", bam], 1), ("mpileup -B -f", ["mpileup", "-B", "-f", fa, bam], 1),
                              ("mpileup -f (chr1 only)", ["mpileup", "-f", fa, "-r", "chr1", bam], n_contigs)):
        dt, _ = timed([ORA] + args)
        print("oracle %-24s %.2f s -> %.1f Mbases/s single thread" % (name, dt, mb / scale / dt))
    # byte parity of what was just timed
    for name, args in (("depth -a", ["depth", "-a", bam]), ("mpileup -B -f", ["mpileup", "-B", "-f", fa, bam]), ("mpileup -f -r chr2", ["mpileup", "-f", fa, "-r", "chr2", bam])):
        a, na = sha_of([ENG] + args)
        b, nb = sha_of([ORA] + args)
        print("parity %-20s engine %d bytes, oracle %d bytes: %s" % (name, na, nb, "IDENTICAL" if a == b else "DIFFERENT"))


if __name__ == "__main__":
    main()
