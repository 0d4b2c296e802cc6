#!/usr/bin/env python3
"""Copy the reference's own golden vectors for the mpileup/depth path into tests/golden/.

/root/reference does not exist on the GPU box, so the small input fixtures and the
expected outputs of every reproducible case in tests/regcases.py are committed under
tests/golden/ (data files only -- no reference source code).  Expected outputs larger
than 16 KiB are stored gzip-compressed.  Run from the repo root in the build container:

    python tests/gen_golden_fixtures.py

Sources: /root/reference/test/bedcov/* and test/coverage/* (+ test/dat/sample.sam) copied whole, /root/reference/test/mpileup/{*.sam,*.bam,*.fa,regions,xx.bed*,expected/*.out},
/root/reference/test/dat/{mpileup.*,view.001.sam}, /root/reference/test/large_pos/*, /root/reference/examples/{ex1.sam.gz,ex1.fa},
/root/reference/test/consensus/{*.sam,*.fa,*.fai,*.bed,expected/*.out},
/root/reference/test/stat/{inputs of regcases.STATS_COV; of the expected files only the COV section}.
"""
import gzip
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import regcases  # noqa: E402

REF = "/root/reference/test"
OUT = os.path.join(HERE, "golden")


def copy(src, dst, gz_over=None):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    if gz_over is not None and os.path.getsize(src) > gz_over:
        with open(src, "rb") as fi, gzip.GzipFile(dst + ".gz", "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
    else:
        shutil.copyfile(src, dst)


def tokens_to_files(argstr, workdir):
    out = []
    for a in argstr.split():
        for part in re.split(r"[,:]", a):
            p = os.path.normpath(os.path.join(workdir, part))
            if os.path.isfile(p):
                out.append(p)
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are already committed")
    for f in os.listdir(os.path.join(REF, "bedcov")):
        copy(os.path.join(REF, "bedcov", f), os.path.join(OUT, "bedcov", f))
    for f in os.listdir(os.path.join(REF, "coverage")):
        copy(os.path.join(REF, "coverage", f), os.path.join(OUT, "coverage", f))
    copy(os.path.join(REF, "dat", "sample.sam"), os.path.join(OUT, "coverage", "sample.sam"))
    inputs = set()
    for exp, args, post in regcases.MPILEUP + regcases.DEPTH + regcases.EXPECTED_FAIL:
        inputs.update(tokens_to_files(args, os.path.join(REF, "mpileup")))
        src = os.path.join(REF, "mpileup", "expected", exp)
        if post == "gz1":
            copy(src + ".f3-6.gz", os.path.join(OUT, "mpileup", "expected", exp + ".f3-6.gz"))
        else:
            copy(src, os.path.join(OUT, "mpileup", "expected", exp), gz_over=16384)
    for exp, args, post in regcases.TESTPL:
        inputs.update(tokens_to_files(args, REF))
        copy(os.path.join(REF, exp), os.path.join(OUT, exp), gz_over=16384)
    copy(os.path.join(REF, "dat", "mpileup.err.1"), os.path.join(OUT, "dat", "mpileup.err.1"))
    for p in sorted(inputs):
        rel = os.path.relpath(p, REF)
        copy(p, os.path.join(OUT, rel))
    # SAM twin of a BAM the reference ships: pins tests/bamio.py (our BAM writer) byte for byte (tests/test_host_io.py)
    copy(os.path.join(REF, "mpileup", "ce#5b.sam"), os.path.join(OUT, "mpileup", "ce#5b.sam"))
    # the only reference test output that depends on bcf_call_glfgen / errmod_cal (row a14): tview's consensus line
    copy(os.path.join(REF, "large_pos", "tview.expected.out"), os.path.join(OUT, "large_pos", "tview.expected.out"))
    # BASELINE.json configs[0]: examples/ex1.sam.gz (headerless SAM, @SQ comes from the FASTA index) + ex1.fa
    for f in ("ex1.sam.gz", "ex1.fa"):
        copy(os.path.join(os.path.dirname(REF), "examples", f), os.path.join(OUT, "examples", f))
    # consensus: test/consensus inputs + every expected file consensus.reg names
    for f in os.listdir(os.path.join(REF, "consensus")):
        if f.endswith((".sam", ".fa", ".fai", ".bed")):
            copy(os.path.join(REF, "consensus", f), os.path.join(OUT, "consensus", f))
    for exp, args, post in regcases.CONSENSUS:
        copy(os.path.join(REF, "consensus", "expected", exp), os.path.join(OUT, "consensus", "expected", exp))
    # mpileup.reg:89 -- read groups of mpileup.1.bam except ERR013140
    d = gzip.open(os.path.join(REF, "mpileup", "mpileup.1.bam")).read()
    rgs = sorted(set(m.group(1).decode() for m in re.finditer(rb"RGZ([A-Z0-9]+)", d)))
    with open(os.path.join(OUT, "mpileup", "35.rg.txt"), "w") as fh:
        for r in rgs:
            if r != "ERR013140":
                fh.write(r + "\n")
    # `stats`, coverage distribution: inputs + the COV section of each expected file (comment line and COV lines)
    for case in regcases.STATS_COV:
        exp, opts, inp = case[:3]
        copy(os.path.join(REF, "stat", inp), os.path.join(OUT, "stat", inp))
        for tok in opts.split():
            if tok.startswith("{G}/"):
                copy(os.path.join(REF, "stat", tok[4:]), os.path.join(OUT, "stat", tok[4:]))
        sec = [l for l in open(os.path.join(REF, "stat", exp)) if l.startswith("COV\t") or l.startswith("# Coverage distribution")]
        with open(os.path.join(OUT, "stat", exp + ".cov"), "w") as fh:
            fh.writelines(sec)
    total = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(OUT) for f in fs)
    print("golden fixtures: %.1f MiB under %s" % (total / 2 ** 20, OUT))


if __name__ == "__main__":
    main()
