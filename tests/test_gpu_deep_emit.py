"""The read-major emit kernel for deep windows (k_mplp_emit_deep, kernels_plp.hip) is chosen by the engine at a mean depth of 100
and above; here it is forced (STA_EMIT_DEEP=1) onto the reference's mpileup goldens -- shallow columns, indels, clips, ref skips,
pads, several files, -a / -aa, regions -- so that every token kind goes through its "plain inside the strip" and its
dealt-out (read, column) paths, with default and with tiny windows.  Needs a real MI355X: -m gpu."""
import os

import pytest

import regcases
from golden_runner import case_paths, first_diff, run_case

pytestmark = pytest.mark.gpu

CASES = [("reg", c) for c in regcases.MPILEUP] + [("testpl", c) for c in regcases.TESTPL]
IDS = ["%s::%s" % (c[0], c[1][:60]) for _, c in CASES]


@pytest.mark.parametrize("window_cols", [None, 37], ids=["default_windows", "tiny_windows"])
@pytest.mark.parametrize("group,case", CASES, ids=IDS)
def test_deep_emit_kernel_matches_reference_golden(product_bin, group, case, window_cols):
    exp, args, post = case
    if window_cols and exp == "1.out":
        pytest.skip("large single-contig case; not with 37-column windows")
    workdir, exp_path = case_paths(group, exp)
    env = dict(os.environ, STA_EMIT_DEEP="1")
    if window_cols:
        env["STA_WINDOW_COLS"] = str(window_cols); env["STA_WINDOW_READS"] = "5"
    ok, got, want, err = run_case(product_bin, workdir, exp_path, args, post, env=env)
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-600:])
