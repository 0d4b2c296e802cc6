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


def run(exe, args, cwd, iterator=False):
    env = dict(os.environ, STA_COV_ITERATOR="1") if iterator else None      # reference loop on the bam_mplp_* surface instead of k_cov_cols
    p = subprocess.run([exe, "bedcov"] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
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
@pytest.mark.parametrize("iterator", [False, True], ids=["device_reduction", "iterator_loop"])
@pytest.mark.parametrize("exp,args", CASES, ids=[c[0] for c in CASES])
def test_engine_bedcov_matches_reference_golden(product_bin, exp, args, iterator):
    assert run(product_bin, args, G, iterator) == open(os.path.join(G, exp), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("iterator", [False, True], ids=["device_reduction", "iterator_loop"])
def test_engine_bedcov_header_and_depth_columns(product_bin, oracle_bin, tmp_path, iterator):
    for args, want in header_cases(tmp_path):
        assert run(product_bin, args, G, iterator) == want
    # options without a reference golden: engine vs oracle
    big = os.path.join(os.path.dirname(G), "mpileup", "mpileup.1.bam")
    bed = tmp_path / "b.bed"; bed.write_text("17\t100\t2000\tx\n17\t3000\t3000\n17\t150\t160\n17\t4000\t9000\n")
    for args in (["-d", "20", "-c", "bedcov_gG.bed", "bedcov.bam"], ["-Q", "30", "-j", "-d", "5", "bedcov_gG.bed", "bedcov.bam", "bedcov.bam"],
                 ["-c", "-d", "0", str(bed), big], ["-j", "-Q", "20", "-g", "1024", str(bed), big, big],
                 ["-j", "-d", "20", "-c", str(bed), big, os.path.join(os.path.dirname(G), "mpileup", "mpileup.2.bam"), big]):
        assert run(product_bin, args, G, iterator) == run(oracle_bin, args, G), args
