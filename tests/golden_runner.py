"""Run a samtools-compatible CLI (oracle or product) over the reference's golden cases.

Used by tests/test_oracle_goldens.py (against /root/reference fixtures, only
when that tree exists -- i.e. in the build container) and by
tests/gen_golden_fixtures.py which copies the small fixtures + expected outputs
into tests/golden/ so the same cases travel to the GPU box.
"""
import gzip
import os
import subprocess
import tempfile

from product_paths import product_exe

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def expand_args(argstr, workdir, tmpdir):
    """Expand the @LIST:/@RG35/@HDRONLY: pseudo arguments of regcases."""
    out = []
    for a in argstr.split():
        if a.startswith("@LIST:"):
            p = os.path.join(tmpdir, "list_%d.txt" % len(out))
            with open(p, "w") as fh:
                for f in a[6:].split(","):
                    fh.write(os.path.join(workdir, f) + "\n")
            out.append(p)
        elif a == "@RG35":
            # mpileup.reg:89 -- all RG ids of mpileup.1.bam except ERR013140
            p = os.path.join(workdir, "35.rg.txt")
            out.append(p)
        elif a.startswith("@HDRONLY:"):
            src = os.path.join(workdir, a[9:])
            p = os.path.join(tmpdir, "hdronly.sam")
            with open(src) as fi, open(p, "w") as fo:
                for line in fi:
                    if line.startswith("@"):
                        fo.write(line)
            out.append(p)
        else:
            out.append(a)
    return out


def postprocess(text, post):
    if post is None:
        return text
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    if post == "col4":
        return "".join(l.split("\t")[3] + "\n" for l in lines)
    if post == "depth1":
        return "".join("\t".join([f[0], f[1], f[3]]) + "\n" for f in (l.split("\t") for l in lines))
    if post == "depth2":
        return "".join("\t".join([f[0], f[1], f[3], f[6]]) + "\n" for f in (l.split("\t") for l in lines))
    if post == "gz1":
        # mpileup.reg:29 -- expected/1.out is stored as columns 3-6 only
        return "".join("\t".join(l.split("\t")[2:6]) + "\n" for l in lines)
    if post.startswith("grep:"):
        return "".join(l + "\n" for l in lines if post[5:] in l)
    raise ValueError(post)


def read_expected(path, post):
    if post == "gz1":
        with gzip.open(path + ".f3-6.gz", "rt") as fh:
            return fh.read()
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        with gzip.open(path + ".gz", "rt", encoding="latin1") as fh:
            return fh.read()
    with open(path, encoding="latin1") as fh:
        return fh.read()


GOLDEN = os.path.join(HERE, "golden")
ORACLE = os.path.join(REPO, "oracle", "_build", "oracle_samtools")
PRODUCT = product_exe()


def case_paths(group, exp):
    """(workdir, expected_path) for a regcases entry of the given group."""
    if group == "testpl":
        return GOLDEN, os.path.join(GOLDEN, exp)
    if group == "consensus":
        return os.path.join(GOLDEN, "consensus"), os.path.join(GOLDEN, "consensus", "expected", exp)
    return os.path.join(GOLDEN, "mpileup"), os.path.join(GOLDEN, "mpileup", "expected", exp)


def run_case(binary, workdir, expected_path, argstr, post, env=None, timeout=600):
    """Returns (ok, got, want, stderr)."""
    with tempfile.TemporaryDirectory() as tmp:
        argv = expand_args(argstr, workdir, tmp)
        cmd = binary if isinstance(binary, list) else [binary]
        if post == "outfile":
            # consensus.reg:62-63 -- `-o cons.tmp; cat cons.tmp`: the -o / --output argument is redirected into tmp
            argv = [os.path.join(tmp, a) if a == "cons.tmp" else a for a in argv]
        p = subprocess.run(cmd + argv, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=env, timeout=timeout)
        out = p.stdout
        if post == "outfile":
            with open(os.path.join(tmp, "cons.tmp"), "rb") as fh:
                out = out + fh.read()
            post = None
    got = postprocess(out.decode("latin1"), post)
    want = read_expected(expected_path, post)
    return got == want, got, want, p.stderr.decode("latin1")


def first_diff(got, want):
    g, w = got.split("\n"), want.split("\n")
    for i, (a, b) in enumerate(zip(g, w)):
        if a != b:
            return "line %d\n  got : %s\n  want: %s" % (i + 1, a[:300], b[:300])
    return "line count got=%d want=%d" % (len(g), len(w))
