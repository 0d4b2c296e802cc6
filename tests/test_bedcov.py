"""`bedcov` (a consumer of the pileup iterator) against the reference's own goldens
(test/bedcov/*.expected, test/test.pl:3817-3868; fixtures copied to tests/golden/bedcov).
CPU: the oracle restatement (oracle/o_bedcov.c on the restated HTSlib iterator) is pinned on them.
GPU: `samtools-amd bedcov` = bedcov.c's column loop on the engine's bam_mplp_* surface must reproduce them too."""
import os
import subprocess

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bedcov")
CASES = [
    ("bedcov.expected", ["bedcov.bed", "bedcov.bam"]),
    ("bedcov_j.expected", ["-j", "bedcov.bed", "bedcov.bam"]),
    ("bedcov_gG.expected", ["-g512", "-G2048", "bedcov_gG.bed", "bedcov.bam"]),
    ("bedcov_c.expected", ["-c", "bedcov_gG.bed", "bedcov.bam"]),
]


def run(exe, args, cwd):
    p = subprocess.run([exe, "bedcov"] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    return p.stdout


def header_cases(tmp_path):
    """The four -H cases of test.pl:3826-3868 (expected text built the way the Perl test builds it)."""
    bam = os.path.join(G, "bedcov.bam")
    out = []
    exp = open(os.path.join(G, "bedcov.expected"), "rb").read()
    out.append((["-H", os.path.join(G, "bedcov.bed"), bam], b"#chrom\tchromStart\tchromEnd\t" + bam.encode() + b"_cov\n" + exp))
    b2 = tmp_path / "h2.bed"; b2.write_bytes(b"#chrom\tchromStart\tchromEnd\tT1\nchr1\t12209228\t12209246\t10\n")
    out.append((["-H", str(b2), bam], b"#chrom\tchromStart\tchromEnd\tT1\t" + bam.encode() + b"_cov\nchr1\t12209228\t12209246\t10\t24\n"))
    b3 = tmp_path / "h3.bed"; b3.write_bytes(b"#chrom\tchromStart\tchromEnd\t\nchr1\t12209228\t12209246\t10\n")
    out.append((["-H", str(b3), bam], b"#chrom\tchromStart\tchromEnd\t\t" + bam.encode() + b"_cov\nchr1\t12209228\t12209246\t10\t24\n"))
    b4 = tmp_path / "h4.bed"; b4.write_bytes(b"chr1\t12209228\t12209246\t4\t5\t6\t7\t8\t9\t10\t11\t12\t13\t14\n")
    out.append((["-H", str(b4), bam],
                b"#chrom\tchromStart\tchromEnd\tname\tscore\tstrand\tthickStart\tthickEnd\titemRgb\tblockCount\tblockSizes\tblockStarts\t.\t.\t"
                + bam.encode() + b"_cov\nchr1\t12209228\t12209246\t4\t5\t6\t7\t8\t9\t10\t11\t12\t13\t14\t24\n"))
    return out


@pytest.mark.parametrize("exp,args", CASES, ids=[c[0] for c in CASES])
def test_oracle_bedcov_matches_reference_golden(oracle_bin, exp, args):
    assert run(oracle_bin, args, G) == open(os.path.join(G, exp), "rb").read()


def test_oracle_bedcov_header_cases(oracle_bin, tmp_path):
    for args, want in header_cases(tmp_path):
        assert run(oracle_bin, args, G) == want


@pytest.mark.gpu
@pytest.mark.parametrize("exp,args", CASES, ids=[c[0] for c in CASES])
def test_engine_bedcov_matches_reference_golden(product_bin, exp, args):
    assert run(product_bin, args, G) == open(os.path.join(G, exp), "rb").read()


@pytest.mark.gpu
def test_engine_bedcov_header_and_depth_columns(product_bin, oracle_bin, tmp_path):
    for args, want in header_cases(tmp_path):
        assert run(product_bin, args, G) == want
    # options without a reference golden: engine vs oracle
    for args in (["-d", "20", "-c", "bedcov_gG.bed", "bedcov.bam"], ["-Q", "30", "-j", "-d", "5", "bedcov_gG.bed", "bedcov.bam", "bedcov.bam"]):
        assert run(product_bin, args, G) == run(oracle_bin, args, G)
