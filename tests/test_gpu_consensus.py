"""`samtools-amd consensus` on the device: every `P` line of the reference's test/consensus/consensus.reg byte
for byte, and the oracle on synthetic inputs (30x pairs with many indels, with and without MD tags; the messy multi-contig
set) for every caller mode and writer, with the windows cut every 1 Mi, 997 and 64 columns.  The same cases run through the
CPU harness in tests/test_consensus_emul.py.  -m gpu."""
import os
import subprocess

import pytest

import regcases
from cons_cases import LARGE_POS, OPTION_SETS, make_inputs
from golden_runner import case_paths, first_diff, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    return make_inputs(tmp_path_factory.mktemp("cons_inputs"))


@pytest.mark.parametrize("case", regcases.CONSENSUS, ids=["%s::%s" % (c[0], c[1][:50]) for c in regcases.CONSENSUS])
def test_device_matches_reference_golden(product_bin, case):
    exp, args, post = case
    workdir, exp_path = case_paths("consensus", exp)
    ok, got, want, err = run_case(product_bin, workdir, exp_path, args, post)
    assert ok, "%s\n%s\nstderr: %s" % (args, first_diff(got, want), err[-400:])


def _run(binary, args, env=None):
    p = subprocess.run([binary, "consensus"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda o: "_".join(a for a in o if not a.startswith("{"))[:60])
def test_device_matches_oracle_on_synthetic_inputs(product_bin, oracle_bin, inputs, opts):
    for sam, fa in inputs:
        args = [a.format(fa=fa) for a in opts] + [sam]
        rc, want, err = _run(oracle_bin, args)
        assert rc == 0, err.decode()[-300:]
        for wc in ("1048576", "997", "64"):
            rc2, got, err2 = _run(product_bin, args, {"STA_WINDOW_COLS": wc})
            assert rc2 == 0, err2.decode()[-300:]
            assert got == want, "%s window %s: %s" % (os.path.basename(sam), wc, first_diff(got.decode("latin1"), want.decode("latin1")))


def test_device_positions_beyond_32_bits(product_bin, oracle_bin):
    for opts in (["-m", "simple", "-f", "pileup"], ["-f", "fastq"], ["-f", "pileup", "-r", "CHROMOSOME_I:10000000000-10000000050"]):
        rc, want, err = _run(oracle_bin, opts + [LARGE_POS])
        rc2, got, err2 = _run(product_bin, opts + [LARGE_POS])
        assert rc == 0 and rc2 == 0, (err, err2)
        assert got == want and len(want) > 100, opts


@pytest.mark.parametrize("mode", ["simple", "bayesian_no_mq", "bayesian"])
def test_bulk_entry_through_the_c_abi(oracle_bin, mode):
    """sta_consensus_run / sta_fetch_consensus through ctypes on device-resident arrays (bench.py's generator): every column of
    one 262 144-column window against the oracle's `-f pileup` rows.  In the Bayesian mode with mapping qualities the MD text
    travels as text column 0 ("*": the generated reads carry no tag)."""
    import shutil
    import numpy as np
    import torch
    import samtools_amd as sa
    import bench
    n_cols = 1 << 18
    inp = bench.synth_inputs("consensus30", n_cols)
    try:
        dev = torch.device("cuda", 0)
        eng = sa.Engine(0, torch.cuda.current_stream().cuda_stream)
        w, keep, _ = bench.build_window(torch, np, sa, inp["rd"], n_cols, dev)
        n = inp["rd"]["n"]
        if mode == "bayesian":
            xo = torch.arange(n + 1, dtype=torch.int32, device=dev); xt = torch.full((n + 1,), ord("*"), dtype=torch.uint8, device=dev)
            keep += [xo, xt]
            w.files[0].n_xcols = 1; w.files[0].xcol_off = xo.data_ptr(); w.files[0].xcol_text = xt.data_ptr(); w.files[0].n_xcol_bytes = n
        eng.stage_window(w)
        p = sa.ConsParams.defaults(want_pileup=1)
        args = ["-f", "pileup"]
        if mode == "simple":
            p.mode = 0; args += ["-m", "simple"]
        elif mode == "bayesian_no_mq":
            p.use_mqual = 0; args += ["--no-use-MQ"]
        info = eng.consensus_run(p)
        ins, cols, off, sq, ql = eng.fetch_consensus(n_cols, info, want_text=True)
        eng.close()
        rc, want, err = _run(oracle_bin, args + [inp["sam"]])
        assert rc == 0, err.decode()[-300:]
    finally:
        shutil.rmtree(inp["dir"], ignore_errors=True)
    rows = want.decode().split("\n")[:-1]
    sqr, qlr = sq.raw, ql.raw
    got = []
    c = 0
    for pos in range(n_cols):
        for k in range(ins[pos] + 1):
            col = cols[c]
            if col.depth > 0 and col.base != ord("*"):
                a, b = off[c], off[c + 1]
                got.append("chrS\t%d\t%d\t%d\t%c\t%d\t%s\t%s" % (pos + 1, k, col.depth, col.base, col.qual, sqr[a:b].decode(), qlr[a:b].decode()))
            c += 1
    assert c == info.n_cols and info.n_kept_reads == n
    assert len(got) == len(rows)
    assert got == rows


def test_window_of_reads_without_seq(tmp_path, product_bin, oracle_bin):
    """A window whose records hold no base at all (SEQ "*" throughout): the per-base preparation has nothing to launch -- a grid of 0 workgroups is
    hipErrorInvalidConfiguration and failed the whole run (found on the device by scripts/hunt5.py / hunt6.py with 5-read windows in round 6; the CPU
    emulation of the kernels now refuses such a launch as the HIP runtime does)."""
    import random
    rnd = random.Random(3)
    ref = "".join(rnd.choice("ACGT") for _ in range(600))
    sam = str(tmp_path / "s.sam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:600\n")
        for i in range(12):
            f.write("a%d\t0\tc1\t%d\t40\t50M\t*\t0\t0\t%s\t%s\n" % (i, 10 + 3 * i, ref[9 + 3 * i:59 + 3 * i], "F" * 50))
        for i in range(9):
            f.write("b%d\t0\tc1\t%d\t40\t40M\t*\t0\t0\t*\t*\n" % (i, 250 + 2 * i))
        for i in range(12):
            f.write("c%d\t16\tc1\t%d\t40\t50M\t*\t0\t0\t%s\t%s\n" % (i, 430 + 3 * i, ref[429 + 3 * i:479 + 3 * i], "F" * 50))
    for opts in (["-f", "pileup", "-d", "3"], ["-f", "fastq", "-a", "--min-MQ", "20", "-C", "30"], ["-m", "simple", "-f", "pileup"]):
        rc, want, err = _run(oracle_bin, opts + [sam])
        assert rc == 0, err
        for env in ({"STA_WINDOW_COLS": "37"}, {"STA_WINDOW_READS": "3", "STA_WINDOW_COLS": "37"}, {"STA_WINDOW_COLS": "100"}, {}):
            rc2, got, err2 = _run(product_bin, opts + [sam], env)
            assert rc2 == 0, (opts, env, err2[-300:])
            assert got == want, (opts, env)
