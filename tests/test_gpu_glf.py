"""Row a14 on the GPU: `samtools-amd glf` (k_glf_cols: bcf_call_glfgen + errmod_cal per column) against the oracle's restatement,
bit for bit on the float outputs, and against the reference's only golden that depends on it (tview's consensus line)."""
import os
import subprocess

import pytest

from synth import write_synth_sam
from synth_rich import write_rich_sam
from test_oracle_goldens import tview_consensus_from_glf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_both(oracle_bin, product_bin, args, env=None):
    want = subprocess.run([oracle_bin, "glf"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    got = subprocess.run([product_bin, "glf"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == 0, got.stderr.decode()[-300:]
    return got.stdout, want


def test_tview_consensus_line_of_the_reference(oracle_bin, product_bin):
    sam = os.path.join(GOLD, "large_pos", "longref.sam")
    got, want = run_both(oracle_bin, product_bin, [sam])
    assert got == want
    line, expected = tview_consensus_from_glf(got.decode(), open(os.path.join(GOLD, "large_pos", "tview.expected.out")).read())
    assert line == expected


@pytest.mark.parametrize("window", [None, "777"])
def test_glf_equals_oracle_on_synthetic_and_messy_input(tmp_path, oracle_bin, product_bin, window):
    env = dict(os.environ)
    if window:
        env["STA_WINDOW_COLS"] = window
    sam, fa = write_synth_sam(str(tmp_path), n_ref=20000, depth=40, read_len=150, seed=31, paired=True, indel_rate=0.05)
    for args in ([sam], ["-f", fa, sam], ["-Q", "0", "-t", "0.7", "-f", fa, sam]):
        got, want = run_both(oracle_bin, product_bin, args, env)
        assert got == want, args
    rich, rfa = write_rich_sam(str(tmp_path), seed=5, n_templates=1500)
    got, want = run_both(oracle_bin, product_bin, ["-f", rfa, rich], env)
    assert got == want
    assert len(want.split(b"\n")) > 50000


def test_glf_deep_columns_are_flagged_and_still_agree(tmp_path, oracle_bin, product_bin):
    """More than 255 counted bases: HTSlib subsamples with its global random stream; engine and oracle both keep the first
    255 in pileup order and raise the flag."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=3000, depth=400, read_len=100, seed=32, paired=False)
    got, want = run_both(oracle_bin, product_bin, ["-f", fa, sam])
    assert got == want
    assert any(l.split(b"\t")[4] == b"1" for l in want.split(b"\n") if l)


@pytest.mark.parametrize("slots", ["auto", "64", "32", "16", "0"])
def test_glf_columns_with_many_distinct_qualities(tmp_path, oracle_bin, product_bin, slots):
    """The kernel keeps a column's counters (errmod_cal's sorted multiset of quality / strand / base) in a few LDS slots per column and sends a
    group of columns whose keys do not fit through the one-byte-per-key form in a second launch (kernels_glf.hip GlfCnt): qualities drawn
    from 0..60 at depth 90 give most columns more distinct keys than 32 or 64 slots hold; at depth 12 everything fits.  STA_GLF_SLOTS=0 is the
    second launch's form for every column, 32 / 64 force the first form."""
    import random
    rnd = random.Random(77)
    for depth, tag in ((90, "deep"), (12, "shallow")):
        sam, fa = write_synth_sam(str(tmp_path), n_ref=2500, depth=depth, read_len=100, seed=33 + depth, paired=False, name="c" + tag)
        lines = []
        for l in open(sam):
            if not l.startswith("@"):
                f = l.rstrip("\n").split("\t")
                f[10] = "".join(chr(33 + rnd.randint(0, 60)) for _ in f[9])
                l = "\t".join(f) + "\n"
            lines.append(l)
        open(sam, "w").writelines(lines)
        env = dict(os.environ)
        env.pop("STA_GLF_SLOTS", None)
        if slots != "auto": env["STA_GLF_SLOTS"] = slots            # (auto: k_glf_tier picks the first form from a sample of the qualities)
        got, want = run_both(oracle_bin, product_bin, ["-Q", "0", "-f", fa, sam], env)
        assert got == want, (tag, slots)
        assert want.count(b"\n") > 2000
