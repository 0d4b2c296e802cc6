"""The lane functions of the class-S BAQ kernel (samtools_amd/csrc/baq_band7s.h: what every lane of k_baq7s executes) run on the
CPU by tests/cpu/baq_emul.cpp against the oracle's sam_prob_realn() (oracle/o_baq.c; HTSlib realn.c / probaln.c, call site
bam_plcmd.c:451) on generated reads: lengths 16..256, soft / hard clips, substitution stretches and shifted halves (the MAP path
leaves the diagonal), ambiguous bases in the read and in the reference, qualities 0..93, extended and per-base mode.
Bit-exact: the resulting quality bytes of every read must be the oracle's.  Test infrastructure: the product has no CPU path."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ (the header uses ext_vector_type)")
    d = tmp_path_factory.mktemp("baq_emul")
    objs = []
    for f in ("o_baq", "o_io"):
        o = str(d / (f + ".o"))
        subprocess.run(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-c", os.path.join(REPO, "oracle", f + ".c"), "-o", o], check=True)
        objs.append(o)
    exe = str(d / "baq_emul")
    subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(REPO, "samtools_amd", "csrc"),
                    os.path.join(HERE, "cpu", "baq_emul.cpp")] + objs + ["-lz", "-lm", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("mode", [16, 0], ids=["table", "formula"])
@pytest.mark.parametrize("seed,force_edge", [(1, 0), (2, 0), (3, 1), (4, 0)])
def test_lane_functions_equal_the_oracle(emul, seed, force_edge, mode):
    """mode = the kernel's STA_BAQ7S_MODE: 16 (default build) takes the MAP quality from the exact threshold table, 0 from the formula"""
    p = subprocess.run([emul, "6000", str(seed), str(force_edge), str(mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out + p.stderr.decode()[-2000:]
    f = out.split()
    n, changed, amb, bad = int(f[1]), int(f[3]), int(f[5]), int(f[7])
    assert n > 5000 and bad == 0
    assert changed > n * 0.9          # BAQ lowers something in nearly every read: the comparison is not vacuous
    assert amb > 100                  # windows with ambiguous reference bases (the all-tests code path) were among them


def test_log_threshold_table_is_exact(emul):
    """k = (int)(-4.343 * log(1 - max) + .499) without the logarithm (baq_band7s.h map_quality): the 101 thresholds are found by bisection
    with the host's own log(), F must be a clean single step over 2 x 10^5 doubles either side of each (else the engine keeps the formula),
    and table == formula on 4 million posteriors: uniform, near 1 over the whole range of 1 - max, within 1e-9 of every threshold, on the
    300 neighbouring doubles of every threshold, and on the special values (posterior 1, 0, 0 / 0)."""
    p = subprocess.run([emul, "logtab", "4000000", "7"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()[-2000:]
    f = p.stdout.decode().split()
    assert int(f[2]) > 4000000 and int(f[4]) == 0
