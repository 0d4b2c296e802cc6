"""The engine's kernels in the CPU suite: a slice of the `-m gpu` parity cases run against the CPU emulation build of the library's
own sources (tests/cpu/hipemu -- test infrastructure, never a fallback; the whole `-m gpu` suite runs on it with STA_HIPEMU=1 set by
hand).  Here: the reference's own golden outputs for a spread of mpileup / depth / consensus options (BAQ, mate overlaps, multiple
files, -6, -E, the generic column walkers, regions, BED lists, deep data with -d), and one synthetic BAQ window against the oracle.
The emulated library is built on demand (clang++, host code only, ~1 min on eight cores); when that build is not possible the test
skips -- the kernels' parity proper is the `-m gpu` suite on the MI355X."""
import fcntl
import os
import subprocess

import pytest

import regcases
from golden_runner import case_paths, first_diff, run_case
from synth import write_synth_sam

pytestmark = pytest.mark.xdist_group("hipemu_smoke")       # one worker runs them all, behind one build

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "cpu", "hipemu")
EXE = os.path.join(EMU, "_build", "plain", "samtools_amd", "bin", "samtools-amd")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

PICK = {"2.out", "8.out", "14.out", "16.out", "20.out", "21.out", "23.out", "25.out", "27.out", "34.out", "39.out", "41.out", "47.out", "76.out", "79.out",
        "mp_DI.out", "a5.out", "d2_12r.out", "d3_12r2a.out", "d5_b3aa.out", "d6_wdel.out", "d8_PROSUP.out", "mp2.out"}


@pytest.fixture(scope="module")
def emu_bin():
    if not os.path.exists(CLANG):
        pytest.skip("needs the ROCm clang++ (host compile)")
    jobs = str(max(1, min(8, os.cpu_count() or 1)))
    with open(os.path.join(EMU, ".build.lock"), "w") as lock:      # (xdist workers: one build at a time)
        fcntl.flock(lock, fcntl.LOCK_EX)
        p = subprocess.run(["make", "-C", EMU, "-j" + jobs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    if p.returncode != 0 or not os.path.exists(EXE):
        pytest.skip("the emulation build failed here: " + p.stdout.decode()[-400:])
    return EXE


def _cases():
    out = []
    for group, table in (("reg", regcases.MPILEUP + regcases.DEPTH), ("testpl", regcases.TESTPL), ("consensus", regcases.CONSENSUS[:6])):
        for c in table:
            if group == "consensus" or c[0] in PICK:
                out.append((group, c))
    return out


@pytest.mark.parametrize("group,case", _cases(), ids=["%s::%s" % (c[0], c[1][:50]) for _, c in _cases()])
def test_emulated_kernels_reproduce_the_reference_goldens(emu_bin, group, case):
    exp, args, post = case
    workdir, exp_path = case_paths(group, exp)
    env = {k: v for k, v in os.environ.items() if not k.startswith("STA_")}
    ok, got, want, err = run_case(emu_bin, workdir, exp_path, args, post, env=env, timeout=900)
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-600:])


def test_emulated_baq_and_pileup_equal_the_oracle_on_a_synthetic_window(emu_bin, oracle_bin, tmp_path):
    sam, fa = write_synth_sam(str(tmp_path), n_ref=4000, depth=30, read_len=150, seed=44, paired=True, indel_rate=0.05)
    for args in (["mpileup", "-f", fa, sam], ["mpileup", "-E", "-A", "-s", "-O", "--output-extra", "QNAME,NM", "-f", fa, sam], ["depth", "-a", "-s", "-J", sam]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got = subprocess.run([emu_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert got.returncode == 0, got.stderr.decode()[-400:]
        assert got.stdout == want and len(want) > 50000, args
