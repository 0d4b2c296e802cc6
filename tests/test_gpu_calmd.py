"""8(f) row 3 on the GPU: `samtools-amd calmd` (k_md_len / k_md_emit / k_calmd_tag + the BAQ kernels in plain and extended
mode) against the oracle's restatement of bam_fillmd1_core and the BAQ tag writer, field for field."""
import os
import subprocess

import pytest

from synth import write_synth_sam
from synth_rich import write_rich_sam
from bamio import sam_to_bam

pytestmark = pytest.mark.gpu
DAT = os.path.join(os.path.dirname(__file__), "golden", "dat")
OPTS = [[], ["-e"], ["-r"], ["-r", "-E"], ["-r", "-A"], ["-r", "-A", "-E", "-e"], ["-q"], ["-n", "2"], ["-e", "-n", "3", "-q", "-r"]]


def run_both(oracle_bin, product_bin, args, env=None, product_args=None):
    want = subprocess.run([oracle_bin, "calmd"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = subprocess.run([product_bin, "calmd"] + (product_args or args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == want.returncode, got.stderr.decode()[-300:]
    return got.stdout, want.stdout


@pytest.mark.parametrize("n", ["1", "2", "3"])
def test_calmd_on_the_reference_inputs(oracle_bin, product_bin, n):
    sam, fa = os.path.join(DAT, "mpileup.%s.sam" % n), os.path.join(DAT, "mpileup.ref.fa")
    for opts in OPTS:
        got, want = run_both(oracle_bin, product_bin, opts + [sam, fa])
        assert got == want, opts
    # the stored aligner tags are reproduced by the device path too
    got, _ = run_both(oracle_bin, product_bin, [sam, fa])
    recs = [l.rstrip("\n").split("\t") for l in open(sam) if not l.startswith("@")]
    for line, rec in zip(got.decode().split("\n"), recs):
        f = line.split("\t")
        tags = {t[:2]: t[5:] for t in rec[11:]}
        if "MD" in tags and f[6] != "*":
            assert f[6].upper() == tags["MD"].upper() and int(f[5]) == int(tags["NM"])


@pytest.mark.parametrize("batch", [None, "97"])
def test_calmd_equals_oracle_on_synthetic_and_messy_input(tmp_path, oracle_bin, product_bin, batch):
    env = dict(os.environ)
    if batch:
        env["STA_CALMD_BATCH"] = batch
    sam, fa = write_synth_sam(str(tmp_path), n_ref=30000, depth=20, read_len=150, seed=41, paired=True, indel_rate=0.08, max_indel=12)
    rich, rfa = write_rich_sam(str(tmp_path), seed=6, n_templates=1200)
    for src, ref in ((sam, fa), (rich, rfa)):
        bam = sam_to_bam(src, src[:-4] + ".bam", level=1, block=30000)
        for opts in OPTS:
            got, want = run_both(oracle_bin, product_bin, opts + [src, ref], env, opts + [bam, ref])
            assert got == want, (os.path.basename(src), opts)
