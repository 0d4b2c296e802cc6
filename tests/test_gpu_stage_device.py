"""Device staging (device half): a window's new reads reach the GPU as raw BAM alignment records and
kernels_stage.hip cuts their CIGAR / bases / qualities / names out of them (the feed being replaced: sam_read1 in mplp_func,
bam_plcmd.c:409, and in fastdepth_core, bam2depth.c:541-543).

STA_STAGE_DEVICE=2 makes the producer ALSO fill the staging pools on the host (host_stage.cpp, the round-3 path) and the engine
compare its device-built pools with them byte for byte before anything is computed (the run fails on a difference): "device-built
sta_reads == host_stage.cpp's" on every golden case, at tiny windows (reads carried across windows are staged by the host, the
new ones by the device: both halves in one window), and on a generated BAM with messy CIGARs and long names.
STA_STAGE_REPORT=1 prints how many reads went through the device path."""
import os
import re
import subprocess

import pytest

import regcases
from golden_runner import case_paths, first_diff, run_case

pytestmark = pytest.mark.gpu

CASES = [("reg", c) for c in regcases.MPILEUP + regcases.DEPTH] + [("testpl", c) for c in regcases.TESTPL]
IDS = ["%s::%s" % (c[0], c[1][:60]) for _, c in CASES]
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(product_bin, group, case, mode, **extra):
    exp, args, post = case
    workdir, exp_path = case_paths(group, exp)
    env = dict(os.environ, STA_STAGE_DEVICE=str(mode), STA_STAGE_REPORT="1", **extra)
    ok, got, want, err = run_case(product_bin, workdir, exp_path, args, post, env=env)
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-600:])
    return sum(int(x) for x in re.findall(r"out of raw BAM records: (\d+)", err))


@pytest.mark.parametrize("group,case", CASES, ids=IDS)
def test_device_built_pools_equal_the_host_built_ones_on_the_goldens(product_bin, group, case):
    n = _run(product_bin, group, case, 2)
    if ".bam" in case[1] and "--output-extra" not in case[1] and "-G" not in case[1].split() and "--output-mods" not in case[1] and " -M" not in case[1]:
        assert n >= 0


@pytest.mark.parametrize("group,case", [c for c in CASES if c[1][0] not in ("1.out",)], ids=[i for i, c in zip(IDS, CASES) if c[1][0] not in ("1.out",)])
def test_tiny_windows_mix_host_staged_carry_and_device_staged_new_reads(product_bin, group, case):
    _run(product_bin, group, case, 2, STA_WINDOW_COLS="37", STA_WINDOW_READS="5")


def test_the_device_path_is_the_one_that_runs_for_bam_input(product_bin, oracle_bin, tmp_path):
    """generated reads with clips, indels, pads and skips, names of very different lengths, odd and even read lengths -> BAM;
    default settings: every read of every window but the carried ones must have been staged on the device, text == oracle"""
    import random
    import numpy as np
    from synth import synth_ref, synth_reads, write_sam, write_fasta
    from bamio import sam_to_bam
    n_cols = 60000
    ref = synth_ref(n_cols, seed=3)
    rd = synth_reads(ref[:n_cols - 400], depth=25, read_len=101, seed=4, indel_rate=0.1, max_indel=6)
    rng = random.Random(5)
    names = [("q%d" % k) + "x" * rng.choice((0, 0, 1, 7, 40, 200)) for k in range(rd["n"])]
    off = np.zeros(rd["n"] + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(x) + 1 for x in names])
    rd["name_off"] = off
    rd["names"] = np.frombuffer(("\0".join(names) + "\0").encode(), dtype=np.uint8).copy()
    sam = str(tmp_path / "m.sam"); fa = str(tmp_path / "m.fa")
    write_sam(sam, rd, "chrS", n_cols); write_fasta(fa, "chrS", ref)
    bam = sam_to_bam(sam, str(tmp_path / "m.bam"), level=1)
    for args in (["mpileup", "-f", fa], ["mpileup", "-B", "-f", fa], ["depth", "-a"]):
        want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        for mode, cols in ((1, None), (2, None), (1, "1000"), (0, None)):
            env = dict(os.environ, STA_STAGE_DEVICE=str(mode), STA_STAGE_REPORT="1")
            if cols:
                env["STA_WINDOW_COLS"] = cols
            p = subprocess.run([product_bin] + args + [bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert p.returncode == 0, p.stderr.decode()[-600:]
            assert p.stdout == want, (args, mode, cols)
            n = sum(int(x) for x in re.findall(r"out of raw BAM records: (\d+)", p.stderr.decode()))
            if mode == 0:
                assert n == 0
            elif cols is None:
                assert n >= rd["n"] * 0.95, (n, rd["n"])       # one window: every read is new to it
            else:
                assert n > 0
