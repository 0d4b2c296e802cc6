"""Pins the CPU oracle against the reference's own golden vectors.

Every reproducible `P` line of test/mpileup/mpileup.reg and depth.reg, the
test.pl mpileup cases, the large-position depth cases and every `P` line of
test/consensus/consensus.reg must match byte for byte.  Runs on CPU (no GPU needed)."""
import os

import pytest

import regcases
from golden_runner import case_paths, first_diff, run_case

CASES = ([("reg", c) for c in regcases.MPILEUP + regcases.DEPTH] + [("testpl", c) for c in regcases.TESTPL]
         + [("consensus", c) for c in regcases.CONSENSUS])


@pytest.mark.parametrize("group,case", CASES, ids=["%s::%s" % (c[0], c[1][:60]) for _, c in CASES])
def test_oracle_matches_reference_golden(oracle_bin, group, case):
    exp, args, post = case
    workdir, exp_path = case_paths(group, exp)
    ok, got, want, err = run_case(oracle_bin, workdir, exp_path, args, post)
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-400:])


def test_oracle_mandatory_stderr_line(oracle_bin):
    # test/test.pl:957 compares stderr with test/dat/mpileup.err.1
    import os, subprocess, tempfile
    from golden_runner import GOLDEN, expand_args
    with tempfile.TemporaryDirectory() as tmp:
        argv = expand_args(regcases.TESTPL[0][1], GOLDEN, tmp)
        p = subprocess.run([oracle_bin] + argv, cwd=GOLDEN, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.stderr.decode() == open(os.path.join(GOLDEN, "dat", "mpileup.err.1")).read()


def tview_consensus_from_glf(glf_text, tview_expected):
    """Rebuild tview's consensus row (third line of `samtools tview -d T -p CHROMOSOME_I:10000000000`, test/test.pl:2910)
    from `glf` output: reference-row '*' marks an insertion column (blank in the consensus row), other columns are
    consecutive positions starting at the requested one; uncovered positions stay blank."""
    calls = {}
    for line in glf_text.split("\n"):
        if line:
            f = line.split("\t")
            calls[int(f[1])] = f[7]
    rows = tview_expected.split("\n")
    ref_row, want = rows[1], rows[2]
    pos, got = 10000000000, ""
    for ch in ref_row:
        if ch == "*":
            got += " "
        else:
            got += calls.get(pos, " ")
            pos += 1
    return got, want.ljust(len(ref_row))


def test_oracle_glfgen_reproduces_tview_consensus_line(oracle_bin):
    """Row a14 (bcf_call_glfgen + errmod_cal + tview's call): the only reference golden that depends on it is the consensus
    line of test/large_pos/tview.expected.out -- 78 characters, one of them a heterozygous call."""
    import subprocess
    gold = os.path.join(os.path.dirname(__file__), "golden", "large_pos")
    out = subprocess.run([oracle_bin, "glf", os.path.join(gold, "longref.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    got, want = tview_consensus_from_glf(out.stdout.decode(), open(os.path.join(gold, "tview.expected.out")).read())
    assert got == want
    assert "K" in want


@pytest.mark.parametrize("n", ["1", "2", "3"])
def test_oracle_calmd_recomputes_the_md_and_nm_tags_the_reference_inputs_carry(oracle_bin, n):
    """8(f) row 3 (bam_fillmd1_core + the record writer): test/dat/mpileup.{1,2,3}.sam carry MD:Z / NM:i written by the aligner
    against test/dat/mpileup.ref.fa, the very pair the reference's calmd test runs on (test/test.pl:3652-3661, which only checks
    the container magic).  bam_md.c:156-193 leaves a tag alone when the recomputed value equals the stored one (MD compared case-
    insensitively), so `calmd in.sam ref.fa` must hand the input back byte for byte and stay silent; a wrong stored tag is
    replaced at the END of the record and reported."""
    import subprocess
    dat = os.path.join(os.path.dirname(__file__), "golden", "dat")
    sam = os.path.join(dat, "mpileup.%s.sam" % n)
    ref = os.path.join(dat, "mpileup.ref.fa")
    out = subprocess.run([oracle_bin, "calmd", sam, ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    src = open(sam, "rb").read()
    assert out.stdout == src and out.stderr == b""
    recs = [l for l in src.decode().splitlines() if not l.startswith("@")]
    assert sum("\tMD:Z:" in l and "\tNM:i:" in l for l in recs) >= 230


def test_oracle_calmd_replaces_wrong_tags_and_appends_missing_ones(oracle_bin, tmp_path):
    """hand-made damage on the reference input: a wrong NM, a wrong MD, both tags stripped"""
    import subprocess
    dat = os.path.join(os.path.dirname(__file__), "golden", "dat")
    ref = os.path.join(dat, "mpileup.ref.fa")
    lines = open(os.path.join(dat, "mpileup.1.sam")).read().splitlines()
    hdr = [l for l in lines if l.startswith("@")]
    recs = [l.split("\t") for l in lines if not l.startswith("@")]
    pick = [r for r in recs if any(t.startswith("MD:Z:") for t in r[11:]) and any(t.startswith("NM:i:") for t in r[11:])][:3]
    good = ["\t".join(r) for r in pick]
    a, b, c = [list(r) for r in pick]
    nm_a = [t for t in a if t.startswith("NM:i:")][0]; md_b = [t for t in b if t.startswith("MD:Z:")][0]
    a[a.index(nm_a)] = "NM:i:77"
    b[b.index(md_b)] = "MD:Z:1A1"
    nm_c = [t for t in c if t.startswith("NM:i:")][0]; md_c = [t for t in c if t.startswith("MD:Z:")][0]
    c = [t for t in c if t not in (nm_c, md_c)]
    sam = tmp_path / "bad.sam"
    sam.write_text("\n".join(hdr + ["\t".join(a), "\t".join(b), "\t".join(c)]) + "\n")
    out = subprocess.run([oracle_bin, "calmd", str(sam), ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    got = [l for l in out.stdout.decode().splitlines() if not l.startswith("@")]
    # the replaced tag moves to the end; everything else stays where it was
    wa = [t for t in good[0].split("\t") if t != nm_a] + [nm_a]
    wb = [t for t in good[1].split("\t") if t != md_b] + [md_b]
    wc = c + [nm_c, md_c]
    assert got == ["\t".join(wa), "\t".join(wb), "\t".join(wc)]
    err = out.stderr.decode()
    assert "[bam_fillmd1] different NM for read '%s': 77 -> %s" % (a[0], nm_a[5:]) in err
    assert "[bam_fillmd1] different MD for read '%s': '1A1' -> '%s'" % (b[0], md_b[5:]) in err
    quiet = subprocess.run([oracle_bin, "calmd", "-Q", str(sam), ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert quiet.stdout == out.stdout and quiet.stderr == b""


@pytest.mark.parametrize("exp,require,exclude", [("44.out", 16, 0), ("46.out", 0, 16)])
def test_oracle_reads_a_filtered_bam_from_stdin(oracle_bin, tmp_path, exp, require, exclude):
    """mpileup.reg:102,104 (43.out, 45.out): `samtools view -h -f 16 | samtools mpileup -x -`.  The expected files are byte
    for byte 44.out / 46.out; `view` is emulated by tests/bamio.py (flag filter on the BAM records), the pipe is real."""
    import gzip
    import subprocess
    from bamio import bam_filter
    gold = os.path.join(os.path.dirname(__file__), "golden", "mpileup")
    src = bam_filter(os.path.join(gold, "mpileup.1.bam"), str(tmp_path / "f.bam"), require, exclude)
    want = gzip.open(os.path.join(gold, "expected", exp + ".gz")).read()
    with open(src, "rb") as fh:
        out = subprocess.run([oracle_bin, "mpileup", "-x", "-"], stdin=fh, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert out.stdout == want


@pytest.mark.skipif(not os.path.exists("/root/reference/bam_consensus_tab.h"), reason="reference tree only exists in the build container")
def test_consensus_tables_are_the_documented_formulas():
    """oracle/o_consensus.c and the product generate q2p[] / mqual_pow_1m[] with pow() from the formulas the reference header
    documents (bam_consensus_tab.h:27-37) instead of carrying its 357 literals: the doubles must be the same bit for bit."""
    import math
    import re
    s = open("/root/reference/bam_consensus_tab.h").read()

    def table(name):
        a = s.index(name)
        return [float(x) for x in re.findall(r"[-+0-9.e]+", s[s.index("{", a) + 1:s.index("};", a)])]
    q2p, mq = table("static double q2p[101]"), table("static double mqual_pow_1m[256]")
    assert len(q2p) == 101 and len(mq) == 256
    assert all(q2p[i] == math.pow(10, -i / 10.0) for i in range(101))
    assert all(mq[i] == math.pow(10, -(i * .9) / 10.0) for i in range(255)) and mq[255] == mq[10]
