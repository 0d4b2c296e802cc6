"""Fourth bug hunt (round 5): the generic walker's emit forms (single-walk with cursors per string, per-column walks, LDS slice / eight-byte
stores) on messy multi-contig inputs (tests/synth_rich.py) under randomly drawn sets of extra columns, against the oracle.
    python scripts/hunt4.py [seed ...]"""
import os, random, subprocess, sys
sys.path.insert(0, "tests")
from synth_rich import write_rich_sam
from bamio import sam_to_bam
out = "/tmp/hunt4"; os.makedirs(out, exist_ok=True)
seeds = [int(x) for x in sys.argv[1:]] or [1, 2]
FLAGCOLS = ["QNAME", "FLAG", "POS", "MAPQ", "RNAME", "RNEXT", "PNEXT", "RLEN"]
TAGS = ["NM", "RG", "MD", "AS", "XS", "ZZ"]
bad = 0
for seed in seeds:
    rnd = random.Random(seed)
    sam, fa = write_rich_sam(out, seed=seed, n_templates=5000)
    sam2, _ = write_rich_sam(out, seed=seed + 100, n_templates=1500)
    bam = sam_to_bam(sam, os.path.join(out, "rich_%d.bam" % seed), level=1, block=20000)
    for case in range(10):
        opts = ["-B"] if rnd.random() < 0.7 else []
        if rnd.random() < 0.5: opts += ["-s"]
        if rnd.random() < 0.5: opts += ["-O"]
        if rnd.random() < 0.3: opts += ["--output-BP-5"]
        if rnd.random() < 0.3: opts += ["--output-MQ"]
        if rnd.random() < 0.3: opts += ["--output-QNAME"]
        cols = rnd.sample(FLAGCOLS, rnd.randint(0, 5)) + rnd.sample(TAGS, rnd.randint(0, 4))
        rnd.shuffle(cols)
        if cols: opts += ["--output-extra", ",".join(cols)]
        if rnd.random() < 0.3: opts += ["--output-sep", ";"]
        if rnd.random() < 0.3: opts += ["--output-empty", "?"]
        if rnd.random() < 0.4: opts += ["-a"] * rnd.randint(1, 2)
        if rnd.random() < 0.4: opts += ["-Q", str(rnd.choice([0, 20, 35]))]
        if rnd.random() < 0.2: opts += ["--reverse-del"]
        if rnd.random() < 0.2: opts += ["-d", "12"]
        if not any(o in opts for o in ("-O", "--output-BP-5", "--output-MQ", "--output-QNAME", "--output-extra")): opts += ["-O"]
        files = [sam, sam2] if rnd.random() < 0.4 else [sam]
        args = ["mpileup"] + opts + ["-f", fa] + files
        o = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        for envx in ({}, {"STA_GENERIC_PASSES": "1"}, {"STA_GENERIC_LDS_CAP": "1024", "STA_WINDOW_COLS": "900"}, {"STA_WINDOW_COLS": "3000", "STA_PLP_BATCH": "700"}):
            eargs = [bam if (a == sam and "STA_PLP_BATCH" in envx) else a for a in args]
            p = subprocess.run([os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")] + eargs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx))
            ok = p.returncode == o.returncode and p.stdout == o.stdout
            print("%s seed %d case %d %s %s rc=%d/%d bytes %d/%d" % ("ok  " if ok else "FAIL", seed, case, envx, " ".join(opts), p.returncode, o.returncode, len(p.stdout), len(o.stdout)), flush=True)
            if not ok:
                bad += 1
                g, w = p.stdout.split(b"\n"), o.stdout.split(b"\n")
                for i, (x, y) in enumerate(zip(g, w)):
                    if x != y:
                        print("   line", i + 1, "\n   got ", x[:400], "\n   want", y[:400]); break
                if p.returncode != o.returncode: print("   stderr engine:", p.stderr.decode()[-300:].replace("\n", " | "))
print("hunt4: %d failures" % bad)
