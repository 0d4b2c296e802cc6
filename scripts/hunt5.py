"""Fifth bug hunt (round 5, after the GPU minutes ran out): randomly drawn commands, option sets, window sizes and input forms over messy
multi-contig inputs (tests/synth_rich.py), engine vs oracle -- sized for the CPU emulation of the kernels (tests/cpu/hipemu):
    STA_EXE=tests/cpu/hipemu/_build/plain/samtools_amd/bin/samtools-amd python scripts/hunt5.py <seed> [<seed> ...]
Every (seed, case) is reproducible: the command line and the environment of a failing run are printed."""
import os, random, subprocess, sys
sys.path.insert(0, "tests")
from synth_rich import write_rich_sam
from bamio import sam_to_bam

EXE = os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")
ORACLE = "oracle/_build/oracle_samtools"
seeds = [int(x) for x in sys.argv[1:]] or [1]
N_CASES = int(os.environ.get("HUNT5_CASES", "40"))
FLAGCOLS = ["QNAME", "FLAG", "POS", "MAPQ", "RNAME", "RNEXT", "PNEXT", "RLEN"]
TAGS = ["NM", "RG", "MD", "AS", "XS", "ZZ"]


def draw_mpileup(rnd, fa, bed):
    o = []
    if rnd.random() < 0.45: o += ["-B"]
    elif rnd.random() < 0.3: o += ["-E"]
    if rnd.random() < 0.25: o += ["-A"]
    if rnd.random() < 0.2: o += ["-x"]
    if rnd.random() < 0.35: o += ["-Q", str(rnd.choice([0, 5, 13, 20, 35]))]
    if rnd.random() < 0.3: o += ["-q", str(rnd.choice([0, 1, 10, 30]))]
    if rnd.random() < 0.25: o += ["-a"] * rnd.randint(1, 2)
    if rnd.random() < 0.2: o += ["-d", str(rnd.choice([3, 8, 15, 60]))]
    if rnd.random() < 0.15: o += ["-C", str(rnd.choice([20, 50]))]
    if rnd.random() < 0.15: o += ["-6"]
    if rnd.random() < 0.15: o += ["--ff", rnd.choice(["UNMAP", "UNMAP,SECONDARY,QCFAIL,DUP", "0x400"])]
    if rnd.random() < 0.1: o += ["--rf", rnd.choice(["PAIRED", "0x1"])]
    if rnd.random() < 0.15: o += ["-l", bed]
    if rnd.random() < 0.2: o += ["-r", rnd.choice(["c1", "c2:100-5000", "c3:1000-30000", "c1:29000-30000"])]
    if rnd.random() < 0.15: o += ["--reverse-del"]
    if rnd.random() < 0.1: o += ["--no-output-ins"] * rnd.randint(1, 2)
    if rnd.random() < 0.1: o += ["--no-output-del"] * rnd.randint(1, 2)
    if rnd.random() < 0.1: o += ["--no-output-ends"]
    if rnd.random() < 0.3:
        if rnd.random() < 0.6: o += ["-s"]
        if rnd.random() < 0.5: o += ["-O"]
        if rnd.random() < 0.3: o += ["--output-BP-5"]
        cols = rnd.sample(FLAGCOLS, rnd.randint(0, 4)) + rnd.sample(TAGS, rnd.randint(0, 3))
        rnd.shuffle(cols)
        if cols: o += ["--output-extra", ",".join(cols)]
        if rnd.random() < 0.3: o += ["--output-sep", ";"]
        if rnd.random() < 0.3: o += ["--output-empty", "?"]
    if "-B" not in o or rnd.random() < 0.7: o += ["-f", fa]
    return ["mpileup"] + o


def draw_depth(rnd, bed):
    o = []
    if rnd.random() < 0.5: o += [rnd.choice(["-a", "-aa"])]
    if rnd.random() < 0.3: o += ["-s"]
    if rnd.random() < 0.3: o += ["-J"]
    if rnd.random() < 0.3: o += ["-Q", str(rnd.choice([0, 5, 20]))]
    if rnd.random() < 0.2: o += ["-l", str(rnd.choice([30, 60, 90]))]
    if rnd.random() < 0.2: o += ["-b", bed]
    if rnd.random() < 0.2: o += ["-r", rnd.choice(["c2", "c3:5000-20000", "c1:1-200"])]
    if rnd.random() < 0.2: o += ["-g", rnd.choice(["0x400", "UNMAP", "SECONDARY"])]
    if rnd.random() < 0.15: o += ["-G", rnd.choice(["16", "0x800"])]
    if rnd.random() < 0.2: o += ["-H"]
    return ["depth"] + o


def draw_consensus(rnd):
    o = ["-f", rnd.choice(["fasta", "fastq", "pileup"])]
    if rnd.random() < 0.4: o += ["-m", "simple"]
    if rnd.random() < 0.3: o += ["-a"]
    if rnd.random() < 0.3: o += ["--show-del", "yes"]
    if rnd.random() < 0.3: o += ["--show-ins", "no"]
    if rnd.random() < 0.3: o += ["-A"]
    if rnd.random() < 0.3: o += ["-d", str(rnd.choice([1, 3, 8]))]
    if rnd.random() < 0.3: o += ["-H", str(rnd.choice([0.3, 0.6]))]
    if rnd.random() < 0.3: o += ["-c", str(rnd.choice([0.5, 0.75]))]
    if rnd.random() < 0.2: o += ["--min-MQ", str(rnd.choice([5, 20]))]
    if rnd.random() < 0.2: o += ["--min-BQ", str(rnd.choice([5, 20]))]
    if rnd.random() < 0.2: o += ["-r", rnd.choice(["c2", "c3:5000-20000"])]
    return ["consensus"] + o


def draw_calmd(rnd):
    o = ["--no-PG"]
    if rnd.random() < 0.3: o += ["-e"]
    if rnd.random() < 0.5: o += ["-r"]
    if rnd.random() < 0.3: o += ["-E"]
    if rnd.random() < 0.3: o += ["-A"]
    if rnd.random() < 0.2: o += ["-q"]
    if rnd.random() < 0.2: o += ["-n", str(rnd.choice([1, 3]))]
    if rnd.random() < 0.2: o += ["-C", str(rnd.choice([20, 50]))]
    if rnd.random() < 0.2: o += ["-d"]
    if rnd.random() < 0.15: o += ["-N"]
    return ["calmd", "-Q"] + o


def draw_other(rnd, fa, bed):
    k = rnd.random()
    if k < 0.25:
        o = ["coverage"]
        if rnd.random() < 0.4: o += ["-Q", "10"]
        if rnd.random() < 0.4: o += ["-q", "5"]
        if rnd.random() < 0.3: o += ["-r", "c3:5000-20000"]
        if rnd.random() < 0.3: o += ["-l", "60"]
        if rnd.random() < 0.25: o += rnd.choice([["-m"], ["-m", "-A"], ["-D"], ["-m", "-w", "60"]])
        return o, 2
    if k < 0.5:
        o = ["bedcov"]
        if rnd.random() < 0.4: o += ["-j"]
        if rnd.random() < 0.4: o += ["-d", str(rnd.choice([2, 8]))]
        if rnd.random() < 0.4: o += ["-c"]
        if rnd.random() < 0.3: o += ["-Q", "20"]
        return o + [bed], 2
    if k < 0.7:
        o = ["plpdump"]
        if rnd.random() < 0.4: o += ["-x"]
        if rnd.random() < 0.3: o += ["-d", str(rnd.choice([5, 12]))]
        if rnd.random() < 0.3: o += ["-p"]
        return o, 2
    if k < 0.85:
        o = ["glf"]
        if rnd.random() < 0.5: o += ["-f", fa]
        if rnd.random() < 0.4: o += ["-Q", "3"]
        return o, 1
    return ["stats"], 1


def main():
    bad = total = 0
    for seed in seeds:
        rnd = random.Random(seed * 7919 + 5)
        out = "/tmp/hunt5_%d" % seed; os.makedirs(out, exist_ok=True)
        nt = rnd.choice([800, 1500, 2500])
        sam, fa = write_rich_sam(out, seed=1000 + seed, n_templates=nt)
        d2 = os.path.join(out, "b"); os.makedirs(d2, exist_ok=True)
        sam2, _ = write_rich_sam(d2, seed=2000 + seed, n_templates=nt // 3)
        bam = sam_to_bam(sam, os.path.join(out, "rich.bam"), level=1, block=rnd.choice([3000, 20000, 0xff00]))
        bed = os.path.join(out, "r.bed")
        with open(bed, "w") as f:
            f.write("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
        for case in range(N_CASES):
            k = rnd.random()
            if k < 0.5: args, nf = draw_mpileup(rnd, fa, bed), 2
            elif k < 0.7: args, nf = draw_depth(rnd, bed), 2
            elif k < 0.8: args, nf = draw_consensus(rnd), 1
            elif k < 0.88: args, nf = draw_calmd(rnd), 1
            else: args, nf = draw_other(rnd, fa, bed)
            files = [sam, sam2] if (nf > 1 and rnd.random() < 0.35) else [sam]
            if args[0] == "calmd": files = [sam, fa]
            env = {}
            if rnd.random() < 0.6: env["STA_WINDOW_COLS"] = str(rnd.choice([37, 300, 900, 3000, 10000]))
            if rnd.random() < 0.3: env["STA_WINDOW_READS"] = str(rnd.choice([5, 50, 700]))
            if rnd.random() < 0.3: env["STA_PLP_BATCH"] = str(rnd.choice([64, 700]))
            if args[0] == "mpileup" and rnd.random() < 0.2: env["STA_EMIT_DEEP"] = rnd.choice(["0", "1"])
            use_bam = rnd.random() < 0.5 and "-H" not in args          # (depth -H prints the file names)
            o = subprocess.run([ORACLE] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            eargs = args + [bam if (use_bam and a == sam) else a for a in files]
            try:
                p = subprocess.run([EXE] + eargs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=1800)
                rc, got, err = p.returncode, p.stdout, p.stderr
            except subprocess.TimeoutExpired:
                rc, got, err = -999, b"", b"timeout"
            total += 1
            ok = rc == o.returncode and got == o.stdout
            print("%s seed %d case %d %s %s rc=%d/%d bytes %d/%d" % ("ok  " if ok else "FAIL", seed, case, env, " ".join(a if len(a) < 30 else "~" + os.path.basename(a) for a in eargs), rc, o.returncode, len(got), len(o.stdout)), flush=True)
            if not ok:
                bad += 1
                g, w = got.split(b"\n"), o.stdout.split(b"\n")
                for i, (x, y) in enumerate(zip(g, w)):
                    if x != y:
                        print("   line", i + 1, "\n   got ", x[:300], "\n   want", y[:300]); break
                if rc != o.returncode: print("   stderr engine:", err.decode(errors="replace")[-300:].replace("\n", " | "), "\n   stderr oracle:", o.stderr.decode(errors="replace")[-200:].replace("\n", " | "))
    print("hunt5: %d failures in %d runs" % (bad, total))


if __name__ == "__main__":
    main()
