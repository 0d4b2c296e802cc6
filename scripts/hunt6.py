"""Sixth bug hunt (round 5, third session, no GPU minutes left): hunt5's commands over inputs enriched with what the messy generator
does not make -- base-modification tags (MM:Z / ML:B: several codes per base, '?' / '.' forms, both strands), precomputed BAQ tags
(BQ:Z), long indel-rich reads (300-1200 bp: the BAQ list kernels), pile-ups of 80-300 reads on one start (the deep emit and the -d
cap), a FASTA with lower case, IUPAC codes and N runs -- and the options hunt5 does not draw (-M / --output-mods, -R, -G, -b lists,
depth -f lists / -d, stats and consensus option sets).  Engine vs oracle, sized for the CPU emulation of the kernels:
    STA_EXE=tests/cpu/hipemu/_build/plain/samtools_amd/bin/samtools-amd python scripts/hunt6.py <seed> [<seed> ...]"""
import os, random, subprocess, sys
sys.path.insert(0, "tests")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth_rich import write_rich_sam
from bamio import sam_to_bam
import hunt5

EXE = os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")
ORACLE = "oracle/_build/oracle_samtools"
N_CASES = int(os.environ.get("HUNT6_CASES", "40"))
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def mm_tags(rnd, seq, flag):
    """MM:Z / ML:B for one record (SAM tags spec 1.7): positions counted on the ORIGINAL read (reverse-complemented back when the
    record is stored reversed), deltas = number of skipped occurrences of the canonical base."""
    orig = "".join(COMP.get(c, "N") for c in reversed(seq)) if flag & 16 else seq
    mm, ml = [], []
    for spec in rnd.sample(["C+m", "C+h", "C+mh", "A+a", "G-m", "N+n", "T+472552", "C+76792", "A-a"], rnd.randint(1, 3)):
        base = spec[0]
        occ = [i for i, c in enumerate(orig) if base == "N" or c == base]
        if not occ:
            continue
        k = rnd.randint(0, min(len(occ), 9))
        pick = sorted(rnd.sample(range(len(occ)), k))
        deltas, prev = [], -1
        for p in pick:
            deltas.append(p - prev - 1); prev = p
        n_codes = len(spec) - 2 if not spec[2:].isdigit() else 1
        mm.append(spec + rnd.choice(["", "", "?", "."]) + "".join(",%d" % d for d in deltas))
        ml += [rnd.choice((0, 3, 127, 128, 200, 255)) for _ in range(k * n_codes)]
    if not mm:
        return []
    out = ["MM:Z:" + ";".join(mm) + ";"]
    if ml or rnd.random() < 0.5:
        out.append("ML:B:C" + "".join(",%d" % v for v in ml))
    return out


def long_read(rnd, ref, cname, clen):
    """one 300-1200 bp read with an insertion or a deletion every ~40-150 bases"""
    L = rnd.randint(300, 1200)
    pos = rnd.randint(0, max(0, clen - 2 * L))
    ops, seq, x, left = [], [], pos, L
    if rnd.random() < 0.3:
        s = rnd.randint(1, 30); ops.append((s, "S")); seq.append("".join(rnd.choice("ACGT") for _ in range(s))); left -= s
    while left > 0:
        m = min(left, rnd.randint(40, 150))
        seg = ref[x:x + m]
        seg = "".join(c if rnd.random() > 0.01 else rnd.choice("ACGT") for c in seg) + "N" * (m - len(seg))
        ops.append((m, "M")); seq.append(seg); x += m; left -= m
        if left <= 0: break
        if rnd.random() < 0.5:
            k = min(left, rnd.randint(1, 14)); ops.append((k, "I")); seq.append("".join(rnd.choice("ACGT") for _ in range(k))); left -= k
            if left <= 0:                      # a read may not end on an insertion here: close with a match
                ops.append((1, "M")); seq.append("A"); x += 1
                break
        else:
            k = rnd.randint(1, 20); ops.append((k, "D")); x += k
    s = "".join(seq)
    q = "".join(chr(33 + rnd.choice((2, 11, 25, 37, 40))) for _ in range(len(s)))
    return pos, "".join("%d%s" % o for o in ops), s, q


def enrich(rnd, sam, fa, out_sam, out_fa, mods=True):
    contigs, refs, order = [], {}, []
    name = None
    for line in open(fa):
        if line.startswith(">"):
            name = line[1:].split()[0]; refs[name] = []; order.append(name)
        else:
            refs[name].append(line.rstrip("\n"))
    refs = {k: "".join(v) for k, v in refs.items()}
    head, recs = [], []
    for line in open(sam):
        if line.startswith("@"):
            head.append(line)
            if line.startswith("@SQ"):
                contigs.append(line.split("\t")[1][3:])
            continue
        f = line.rstrip("\n").split("\t")
        flag = int(f[1])
        if f[2] == "*":
            recs.append((1 << 30, 0, len(recs), f)); continue
        if f[9] != "*" and not (flag & 4):
            if mods and rnd.random() < 0.25:
                f += mm_tags(rnd, f[9], flag)
            if rnd.random() < 0.12:
                f.append("BQ:Z:" + "".join(chr(64 + (rnd.randint(0, 40) if rnd.random() < 0.2 else 0)) for _ in f[9]))
        recs.append((contigs.index(f[2]), int(f[3]) - 1, len(recs), f))
    tid = {c: i for i, c in enumerate(contigs)}
    n = len(recs)
    for k in range(rnd.randint(3, 10)):
        c = rnd.choice(contigs)
        pos, cig, s, q = long_read(rnd, refs[c], c, len(refs[c]))
        recs.append((tid[c], pos, n, ["long%d" % k, str(rnd.choice((0, 16))), c, str(pos + 1), "60", cig, "*", "0", "0", s, q, "RG:Z:g1"])); n += 1
    for k in range(rnd.randint(1, 2)):
        c = rnd.choice(contigs)
        L = rnd.randint(60, 120)
        pos = rnd.randint(0, len(refs[c]) - 2 * L)
        for j in range(rnd.randint(80, 300)):
            p = pos + (rnd.randint(0, 3) if rnd.random() < 0.3 else 0)
            s = "".join(ch if rnd.random() > 0.03 else rnd.choice("ACGT") for ch in refs[c][p:p + L])
            cig = "%dM" % L
            if rnd.random() < 0.1 and L > 40:
                a = rnd.randint(10, L - 20); d = rnd.randint(1, 5)
                cig = "%dM%dD%dM" % (a, d, L - a); s = s[:a] + refs[c][p + a + d:p + d + L]
            q = "".join(chr(33 + rnd.choice((2, 11, 25, 37))) for _ in s)
            recs.append((tid[c], p, n, ["hot%d_%d" % (k, j), str(rnd.choice((0, 16, 1024))), c, str(p + 1), str(rnd.choice((60, 60, 20, 0))), cig, "*", "0", "0", s, q, "RG:Z:g2"])); n += 1
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    with open(out_sam, "w") as fh:
        fh.writelines(head)
        for r in recs:
            fh.write("\t".join(r[3]) + "\n")
    # the FASTA the commands are given: case, IUPAC codes, N runs (the reads were drawn from the plain sequence: mismatches there)
    with open(out_fa, "w") as fh, open(out_fa + ".full.fa", "w") as fh_full:
        for c in order:
            s = list(refs[c])
            for _ in range(rnd.randint(2, 8)):
                a = rnd.randint(0, len(s) - 1); b = min(len(s), a + rnd.randint(1, 400))
                kind = rnd.random()
                for i in range(a, b):
                    if kind < 0.5: s[i] = s[i].lower()
                    elif kind < 0.75: s[i] = rnd.choice("RYMKSWHBVDNn") if rnd.random() < 0.3 else s[i]
                    else: s[i] = "N" if b - a < 60 else s[i]
            s = "".join(s)
            fh_full.write(">%s\n" % c)             # (consensus -T reads c->ref[pos] without looking at its length, bam_consensus.c:2368,2502: it gets this one)
            for i in range(0, len(s), 60):
                fh_full.write(s[i:i + 60] + "\n")
            # a contig shorter in the FASTA than its @SQ LN says (reads beyond its end: bam_plcmd.c:440-445), or absent from it
            k = rnd.random()
            if k < 0.2: s = s[:rnd.randint(len(s) // 2, len(s) - 1)]
            elif k < 0.3 and c != order[0]: continue
            fh.write(">%s\n" % c)
            for i in range(0, len(s), 60):
                fh.write(s[i:i + 60] + "\n")
    return out_sam, out_fa


def draw_mpileup6(rnd, fa, bed, rgfile):
    a = hunt5.draw_mpileup(rnd, fa, bed)
    if rnd.random() < 0.3: a.insert(1, rnd.choice(["-M", "--output-mods"]))
    if rnd.random() < 0.1 and ("-M" in a or "--output-mods" in a): a.insert(1, "--no-output-ins-mods")
    if rnd.random() < 0.1: a.insert(1, "-R")
    if rnd.random() < 0.15: a[1:1] = ["-G", rgfile]
    if rnd.random() < 0.1: a[1:1] = ["--output-QNAME"]
    return a


def draw_stats(rnd, tgt):
    o = ["stats"]
    if rnd.random() < 0.4: o += ["-c", rnd.choice(["1,100,1", "1,50,5", "2,300,10", "1,8,1"])]
    if rnd.random() < 0.2: o += ["-d"]
    if rnd.random() < 0.2: o += ["-f", rnd.choice(["PAIRED", "0x2"])]
    if rnd.random() < 0.2: o += ["-F", rnd.choice(["0x800", "SECONDARY,QCFAIL"])]
    if rnd.random() < 0.2: o += ["-l", str(rnd.choice([60, 100, 150]))]
    if rnd.random() < 0.2: o += ["-I", rnd.choice(["g1", "s2", "nope"])]
    if rnd.random() < 0.2: o += ["-t", tgt]
    if rnd.random() < 0.2: o += ["-p"]
    return o


def draw_consensus6(rnd, fa):
    o = hunt5.draw_consensus(rnd)
    if rnd.random() < 0.2: o += ["-l", str(rnd.choice([1, 50, 70, 0]))]
    if rnd.random() < 0.2: o += ["-C", str(rnd.choice([0, 10, 30]))]
    if rnd.random() < 0.15: o += ["-5"]
    if rnd.random() < 0.15: o += ["--het-only"]
    if rnd.random() < 0.15: o += ["--mark-ins"]
    if rnd.random() < 0.15: o += ["--no-adj-qual"]
    if rnd.random() < 0.15: o += ["--no-use-MQ"]
    if rnd.random() < 0.15: o += ["--no-adj-MQ"]
    if rnd.random() < 0.15: o += ["--low-MQ", "10", "--high-MQ", "40"]
    if rnd.random() < 0.15: o += ["--scale-MQ", "0.7"]
    if rnd.random() < 0.15: o += ["--NM-halo", "20", "--SC-cost", "30"]
    if rnd.random() < 0.15: o += ["--P-het", "0.01"]
    if rnd.random() < 0.15: o += ["--P-indel", "0.001"]
    if rnd.random() < 0.15: o += ["--het-scale", "0.5"]
    if rnd.random() < 0.15: o += ["-p"]
    if rnd.random() < 0.2: o += ["-X", rnd.choice(["hiseq", "hifi", "r10.4_sup", "r10.4_dup", "ultima"])]
    if rnd.random() < 0.15: o += ["-t", rnd.choice(["hiseq", "hifi", "flat"])]
    if rnd.random() < 0.15: o += ["--default-qual", "20"]
    if rnd.random() < 0.2: o += ["-T", fa]
    if rnd.random() < 0.1: o += ["--ff", "UNMAP,DUP"]
    return o


def main():
    bad = total = 0
    for seed in hunt5.seeds:
        rnd = random.Random(seed * 104729 + 6)
        out = "/tmp/hunt6_%d" % seed; os.makedirs(out, exist_ok=True)
        nt = rnd.choice([600, 1200, 2000])
        sam0, fa0 = write_rich_sam(out, seed=3000 + seed, n_templates=nt)
        sam, fa = enrich(rnd, sam0, fa0, os.path.join(out, "e.sam"), os.path.join(out, "e.fa"))
        d2 = os.path.join(out, "b"); os.makedirs(d2, exist_ok=True)
        sam2, _ = write_rich_sam(d2, seed=4000 + seed, n_templates=nt // 3)
        bam = sam_to_bam(sam, os.path.join(out, "e.bam"), level=1, block=rnd.choice([3000, 20000, 0xff00]))
        bed = os.path.join(out, "r.bed")
        with open(bed, "w") as f:
            f.write("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
        tgt = os.path.join(out, "t.txt")
        with open(tgt, "w") as f:
            f.write("# targets\nc1\t100\t9000\nc1\t9500\t9600\nzz\t1\t5\nc3\t20000\t44000\n")
        rgfile = os.path.join(out, "rg.txt")
        with open(rgfile, "w") as f:
            f.write(rnd.choice(["g1\n", "g2\n", "g1\ng2\n", "nope\n"]))
        blist = os.path.join(out, "bams.txt")
        with open(blist, "w") as f:
            f.write(sam + "\n" + sam2 + "\n")
        for case in range(N_CASES):
            k = rnd.random()
            files = None
            if k < 0.45:
                args, nf = draw_mpileup6(rnd, fa, bed, rgfile), 2
                if rnd.random() < 0.1: args[1:1] = ["-b", blist]; files = []
            elif k < 0.6:
                args, nf = hunt5.draw_depth(rnd, bed), 2
                if rnd.random() < 0.2: args += ["-d", str(rnd.choice([0, 5, 100]))]
                if rnd.random() < 0.15: args += ["-f", blist]; files = []
            elif k < 0.75: args, nf = draw_consensus6(rnd, fa + ".full.fa"), 1
            elif k < 0.83: args, nf = hunt5.draw_calmd(rnd), 1
            elif k < 0.9: args, nf = draw_stats(rnd, tgt), 1
            else: args, nf = hunt5.draw_other(rnd, fa, bed)
            if files is None:
                files = [sam, sam2] if (nf > 1 and rnd.random() < 0.35) else [sam]
            if args[0] == "calmd": files = [sam, fa]
            env = {}
            if rnd.random() < 0.6: env["STA_WINDOW_COLS"] = str(rnd.choice([37, 300, 900, 3000, 10000]))
            if rnd.random() < 0.3: env["STA_WINDOW_READS"] = str(rnd.choice([5, 50, 700]))
            if rnd.random() < 0.3: env["STA_PLP_BATCH"] = str(rnd.choice([64, 700]))
            if args[0] == "mpileup" and rnd.random() < 0.2: env["STA_EMIT_DEEP"] = rnd.choice(["0", "1"])
            use_bam = rnd.random() < 0.5 and "-H" not in args
            o = subprocess.run([ORACLE] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            eargs = args + [bam if (use_bam and a == sam) else a for a in files]
            # SURVEY.md 8e through the CLI: the command cut into W blocks of reference columns (STA_SHARD=r/W, random cuts), the blocks'
            # texts concatenated in rank order = the unsharded text.  (A single -a and a -d cap that can trigger are refused: skipped.)
            world = rnd.choice([2, 3, 5, 8]) if (args[0] in ("mpileup", "depth") and rnd.random() < 0.3 and "-H" not in args) else 1
            if world > 1 and "-r" not in args and rnd.random() < 0.7:
                env["STA_SHARD_CUTS"] = ",".join(str(c) for c in sorted(rnd.randint(0, 84000) for _ in range(world - 1)))
            try:
                rc, got, err = 0, b"", b""
                for r in range(world):
                    if world > 1: env["STA_SHARD"] = "%d/%d" % (r, world)
                    p = subprocess.run([EXE] + eargs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=1800)
                    rc, got, err = max(rc, p.returncode), got + p.stdout, err + p.stderr
            except subprocess.TimeoutExpired:
                rc, got, err = -999, b"", b"timeout"
            if world > 1 and rc == 1 and (b"supports no single -a" in err or b"-d depth cap can trigger" in err):
                print("skip seed %d case %d (refused in a sharded run) %s" % (seed, case, " ".join(args[:8])), flush=True)
                continue
            total += 1
            ok = rc == o.returncode and got == o.stdout
            print("%s seed %d case %d %s %s rc=%d/%d bytes %d/%d" % ("ok  " if ok else "FAIL", seed, case, env, " ".join(a if len(a) < 30 else "~" + os.path.basename(a) for a in eargs), rc, o.returncode, len(got), len(o.stdout)), flush=True)
            if not ok:
                bad += 1
                g, w = got.split(b"\n"), o.stdout.split(b"\n")
                for i, (x, y) in enumerate(zip(g, w)):
                    if x != y:
                        print("   line", i + 1, "\n   got ", x[:300], "\n   want", y[:300]); break
                if rc != o.returncode or not got: print("   stderr engine:", err.decode(errors="replace")[-300:].replace("\n", " | "), "\n   stderr oracle:", o.stderr.decode(errors="replace")[-200:].replace("\n", " | "))
    print("hunt6: %d failures in %d runs" % (bad, total))


if __name__ == "__main__":
    main()
