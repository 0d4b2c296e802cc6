"""The consensus path's device logic, run on the CPU.  samtools_amd/csrc/cons_core.h + cons_window.h hold the
per-read / per-column step functions the HIP kernels execute; tests/cpu/cons_emul.cpp runs those same functions in loops
behind the product's own command driver (driver_consensus.cpp).  Here that harness must reproduce (a) every `P` line of the
reference's test/consensus/consensus.reg byte for byte and (b) the oracle on synthetic inputs, with the windows cut every
1 Mi, 997 and 64 columns.  The device run of the same cases is tests/test_gpu_consensus.py."""
import os
import subprocess

import pytest

import regcases
from cons_cases import LARGE_POS, OPTION_SETS, make_inputs
from golden_runner import case_paths, first_diff, run_case

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul_bin(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cons_emul") / "cons_emul")
    subprocess.run([os.path.join(REPO, "scripts", "build_cons_emul.sh"), exe], check=True)
    return exe


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return make_inputs(tmp_path_factory.mktemp("cons_inputs"))


@pytest.mark.parametrize("case", regcases.CONSENSUS, ids=["%s::%s" % (c[0], c[1][:50]) for c in regcases.CONSENSUS])
def test_harness_matches_reference_golden(emul_bin, case):
    exp, args, post = case
    workdir, exp_path = case_paths("consensus", exp)
    ok, got, want, err = run_case(emul_bin, workdir, exp_path, args, post, env=dict(os.environ, STA_NO_PINNED="1"))
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-400:])


def _run(binary, args, env=None):
    p = subprocess.run([binary, "consensus"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_NO_PINNED="1", **(env or {})))
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda o: "_".join(a for a in o if not a.startswith("{"))[:60])
def test_harness_matches_oracle_on_synthetic_inputs(emul_bin, oracle_bin, inputs, opts):
    for sam, fa in inputs:
        args = [a.format(fa=fa) for a in opts] + [sam]
        rc, want, err = _run(oracle_bin, args)
        assert rc == 0, err.decode()[-300:]
        assert len(want) > 500
        for wc in ("1048576", "997", "64"):
            rc2, got, err2 = _run(emul_bin, args, {"STA_WINDOW_COLS": wc})
            assert rc2 == 0, err2.decode()[-300:]
            assert got == want, "%s window %s: %s" % (os.path.basename(sam), wc, first_diff(got.decode("latin1"), want.decode("latin1")))


def test_harness_positions_beyond_32_bits(emul_bin, oracle_bin):
    for opts in (["-m", "simple", "-f", "pileup"], ["-f", "fastq"], ["-f", "pileup", "-r", "CHROMOSOME_I:10000000000-10000000050"]):
        rc, want, err = _run(oracle_bin, opts + [LARGE_POS])
        rc2, got, err2 = _run(emul_bin, opts + [LARGE_POS])
        assert rc == 0 and rc2 == 0, (err, err2)
        assert got == want and len(want) > 100, opts
