"""Pins the CPU oracle against the reference's own golden vectors.

Every reproducible `P` line of test/mpileup/mpileup.reg and depth.reg, the
test.pl mpileup cases and the large-position depth cases must match byte for
byte.  Runs on CPU (no GPU needed)."""
import pytest

import regcases
from golden_runner import case_paths, first_diff, run_case

CASES = [("reg", c) for c in regcases.MPILEUP + regcases.DEPTH] + [("testpl", c) for c in regcases.TESTPL]


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
