"""Eighth hunt (round 5): templates with MORE THAN TWO records under tiny windows -- the class that produced findings 4, 7 and 9 of
profiles/r05_hipemu_findings.md.  The messy generator's pairs get extra records of the same name for a third of the proper pairs:
supplementary / secondary alignments of either mate placed upstream (0-60 columns in front), over the first mate, between the mates,
over the second mate or downstream, with the proper-pair bit and the mate position drawn at random, sometimes two of them.  mpileup
(overlap resolution on) and depth -s, windows of 2-13 reads or 37-300 columns, both input lanes, SAM and BAM, engine vs oracle:
    STA_EXE=tests/cpu/hipemu/_build/plain/samtools_amd/bin/samtools-amd python scripts/hunt8.py <seed> [<seed> ...]"""
import os, random, subprocess, sys
sys.path.insert(0, "tests")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth_rich import write_rich_sam
from bamio import sam_to_bam
import hunt5

EXE = os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")
ORACLE = "oracle/_build/oracle_samtools"
N_CASES = int(os.environ.get("HUNT8_CASES", "40"))


def add_records(rnd, src, dst, fa):
    refs, name = {}, None
    for line in open(fa):
        if line.startswith(">"): name = line[1:].split()[0]; refs[name] = []
        else: refs[name].append(line.strip())
    refs = {k: "".join(v) for k, v in refs.items()}
    head, recs, contigs = [], [], []
    for line in open(src):
        if line.startswith("@"):
            head.append(line)
            if line.startswith("@SQ"): contigs.append(line.split("\t")[1][3:])
        else:
            recs.append(line.rstrip("\n").split("\t"))
    by_name = {}
    for f in recs:
        if f[2] != "*": by_name.setdefault(f[0], []).append(f)
    extra = []
    for nm, fs in by_name.items():
        if len(fs) != 2 or not (int(fs[0][1]) & 1) or fs[0][2] != fs[1][2] or rnd.random() > 0.33:
            continue
        a, b = sorted(fs, key=lambda f: int(f[3]))
        pa, pb = int(a[3]), int(b[3])
        for _ in range(1 if rnd.random() < 0.8 else 2):
            L = rnd.randint(25, 90)
            where = rnd.randrange(5)
            if where == 0: pos = pa - L - rnd.randint(0, 60)                        # upstream, ending 0-60 columns in front of the first mate
            elif where == 1: pos = pa + rnd.randint(-L // 2, 40)                    # over the first mate
            elif where == 2: pos = (pa + pb) // 2 + rnd.randint(-20, 20)            # between / over both
            elif where == 3: pos = pb + rnd.randint(-L // 2, 40)                    # over the second mate
            else: pos = pb + rnd.randint(60, 300)                                   # downstream
            ref = refs[a[2]]
            pos = max(1, min(pos, len(ref) - L - 1))
            of = rnd.choice((a, b))
            flag = (int(of[1]) & ~(2 | 256 | 2048)) | rnd.choice((2048, 2048, 256)) | (2 if rnd.random() < 0.7 else 0)
            seq = "".join(c if rnd.random() > 0.03 else rnd.choice("ACGT") for c in ref[pos - 1:pos - 1 + L])
            cig = "%dM" % L
            if rnd.random() < 0.25 and L > 30:
                k = rnd.randint(8, L - 12); d = rnd.randint(1, 6)
                cig = "%dM%dD%dM" % (k, d, L - k); seq = seq[:k] + ref[pos - 1 + k + d:pos - 1 + d + L]
            qual = "".join(chr(33 + rnd.choice((2, 11, 25, 37, 40))) for _ in seq)
            mate = rnd.choice((a, b))
            extra.append([nm, str(flag), a[2], str(pos), a[4], cig, "=", mate[3], "0", seq, qual, "RG:Z:g1"])
    allr = [(contigs.index(f[2]) if f[2] != "*" else 1 << 30, int(f[3]), i, f) for i, f in enumerate(recs)]
    allr += [(contigs.index(f[2]), int(f[3]), len(recs) + i, f) for i, f in enumerate(extra)]
    allr.sort(key=lambda t: (t[0], t[1], t[2]))
    with open(dst, "w") as fh:
        fh.writelines(head)
        for t in allr: fh.write("\t".join(t[3]) + "\n")
    return len(extra)


def main():
    bad = total = 0
    for seed in hunt5.seeds:
        rnd = random.Random(seed * 32452843 + 8)
        out = "/tmp/hunt8_%d" % seed; os.makedirs(out, exist_ok=True)
        sam0, fa = write_rich_sam(out, seed=7000 + seed, n_templates=rnd.choice([500, 1000, 1800]))
        sam = os.path.join(out, "m.sam")
        n = add_records(rnd, sam0, sam, fa)
        bam = sam_to_bam(sam, os.path.join(out, "m.bam"), level=1, block=rnd.choice([3000, 20000]))
        bed = os.path.join(out, "r.bed")
        with open(bed, "w") as f:
            f.write("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
        print("seed %d: %d extra records" % (seed, n), flush=True)
        for case in range(N_CASES):
            k = rnd.random()
            if os.environ.get("HUNT8_NO_DEPTH") and 0.6 <= k < 0.9: k = 0.0          # (depth -s: see DESIGN.md section 8-7, a known gap)
            if k < 0.6:
                args = ["mpileup"] + rnd.choice([[], ["-B"], ["-B", "-Q", "0"], ["-A", "-B"], ["-Q", "0"], ["-B", "-q", "20"], ["-B", "--rf", "PAIRED"],
                                                 ["-B", "--ff", "UNMAP,SECONDARY,QCFAIL,DUP,SUPPLEMENTARY"], ["-B", "-l", bed], ["-B", "-d", "60"], ["-B", "-s", "-O"]]) + ["-f", fa]
            elif k < 0.9:
                args = ["depth", "-s"] + rnd.choice([[], ["-aa"], ["-J"], ["-Q", "20"], ["-g", "SECONDARY"], ["-G", "0x800"], ["-b", bed], ["-l", "40"]])
            else:
                args = rnd.choice([["coverage"], ["bedcov", bed], ["stats", "-p"], ["plpdump"], ["consensus", "-f", "pileup"]])
            env = {}
            if rnd.random() < 0.7: env["STA_WINDOW_READS"] = str(rnd.choice([2, 3, 4, 5, 6, 8, 13]))
            if rnd.random() < 0.4: env["STA_WINDOW_COLS"] = str(rnd.choice([37, 100, 300]))
            if rnd.random() < 0.25: env["STA_IO_LANE"] = "rec"
            if rnd.random() < 0.2: env["STA_PLP_BATCH"] = "64"
            inp = bam if rnd.random() < 0.5 else sam
            bedcov = args[0] == "bedcov"
            o = subprocess.run([ORACLE] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            try:
                p = subprocess.run([EXE] + args + [inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=900)
                rc, got, err = p.returncode, p.stdout, p.stderr
            except subprocess.TimeoutExpired:
                rc, got, err = -999, b"", b"timeout"
            total += 1
            ok = rc == o.returncode and got == o.stdout
            print("%s seed %d case %d %s %s rc=%d/%d bytes %d/%d" % ("ok  " if ok else "FAIL", seed, case, env, " ".join(a if len(a) < 30 else "~" + os.path.basename(a) for a in args + [inp]), rc, o.returncode, len(got), len(o.stdout)), flush=True)
            if not ok:
                bad += 1
                g, w = got.split(b"\n"), o.stdout.split(b"\n")
                for i, (x, y) in enumerate(zip(g, w)):
                    if x != y:
                        print("   line", i + 1, "\n   got ", x[:300], "\n   want", y[:300]); break
                if rc != o.returncode: print("   stderr engine:", err.decode(errors="replace")[-300:].replace("\n", " | "))
    print("hunt8: %d failures in %d runs" % (bad, total))


if __name__ == "__main__":
    main()
