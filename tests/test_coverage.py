"""`coverage` (tabular mode) against the reference's goldens test/coverage/{1..5}.expected
(test/test.pl:4143-4161; fixtures under tests/golden/coverage, input = test/dat/sample.sam).
CPU: oracle/o_coverage.c on the restated HTSlib iterator.  GPU: coverage.c's loop on the engine's bam_mplp_* surface."""
import os
import subprocess

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coverage")


def cases(tmp_path):
    s = os.path.join(G, "sample.sam")
    s1 = str(tmp_path / "sample1.sam")          # test.pl: sed '/A1/d'
    with open(s) as fi, open(s1, "w") as fo:
        fo.writelines(l for l in fi if "A1" not in l)
    return [("1.expected", [s]), ("1.expected", ["--min-depth", "1", s]), ("2.expected", ["--min-depth", "2", s]),
            ("3.expected", ["--min-depth", "2", "-Q", "8", "-q", "45", s]), ("4.expected", ["--min-depth", "1", s, s1]),
            ("5.expected", ["--min-depth", "4", s, s1])]


def run(exe, args, iterator=False):
    env = dict(os.environ, STA_COV_ITERATOR="1") if iterator else None      # reference loop on the bam_mplp_* surface instead of k_cov_cols
    p = subprocess.run([exe, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    return p.stdout


def test_oracle_coverage_matches_reference_goldens(oracle_bin, tmp_path):
    for exp, args in cases(tmp_path):
        assert run(oracle_bin, args) == open(os.path.join(G, exp), "rb").read(), exp


@pytest.mark.gpu
@pytest.mark.parametrize("iterator", [False, True], ids=["device_reduction", "iterator_loop"])
def test_engine_coverage_matches_reference_goldens(product_bin, oracle_bin, tmp_path, iterator):
    for exp, args in cases(tmp_path):
        assert run(product_bin, args, iterator) == open(os.path.join(G, exp), "rb").read(), exp
    # options without a golden: engine vs oracle (region, flags, read length, header off) on a bigger file
    big = os.path.join(os.path.dirname(G), "dat", "mpileup.1.sam")
    three = [os.path.join(os.path.dirname(G), "dat", "mpileup.%d.sam" % k) for k in (1, 2, 3)]
    for args in (["-r", "17:200-900", big], ["-H", "--ff", "UNMAP,DUP", "-l", "50", "-Q", "20", big], ["-q", "30", "-d", "10", big],
                 ["--min-depth", "40", "-Q", "30"] + three, ["-r", "17:4100-4200", big]):
        assert run(product_bin, args, iterator) == run(oracle_bin, args), args
