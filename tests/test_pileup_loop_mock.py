"""sta_pileup_loop() host logic on the CPU: samtools_amd/csrc/cons_loop_api.cpp (record batches, window cuts, look-back reads,
seq_init / seq_column / seq_free protocol) is linked with a MOCK engine (tests/cpu/cons_mock_engine.cpp: the five engine entry
points it calls, implemented with the shared step functions of cons_window.h) instead of the device library, and driven by the
external C99 client tests/cabi/cons_client.c.  Every pileup_t field of every column must equal the oracle's `consensus -f dump`,
for record batches of 40 / 3000 / 65536, also under AddressSanitizer + UBSan.  The device run is tests/test_gpu_cabi_client.py."""
import os
import subprocess

import pytest

from synth import write_synth_sam
from synth_rich import write_rich_sam

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
G = os.path.join(HERE, "golden", "consensus")


def _build(outdir, sanitize):
    d = str(outdir)
    flags = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if sanitize else []
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared"] + flags + ["-o", os.path.join(d, "libsamtools_amd.so"),
                    os.path.join(HERE, "cpu", "cons_mock_engine.cpp"), os.path.join(REPO, "samtools_amd", "csrc", "cons_loop_api.cpp")], check=True)
    exe = os.path.join(d, "cons_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-g", "-DSTA_CONS_DROPIN"] + flags +
                   ["-I", os.path.join(REPO, "include"), os.path.join(HERE, "cabi", "cons_client.c"), "-L", d, "-lsamtools_amd", "-Wl,-rpath," + d, "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module", params=[False, True], ids=["plain", "asan_ubsan"])
def client(request, tmp_path_factory):
    return _build(tmp_path_factory.mktemp("cons_mock"), request.param)


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("cons_mock_in")
    sam, _ = write_synth_sam(str(d), n_ref=8000, depth=25, read_len=150, seed=75, paired=True, indel_rate=0.3, max_indel=6)
    os.makedirs(str(d / "rich"), exist_ok=True)
    rich, _ = write_rich_sam(str(d / "rich"), seed=13, n_templates=800)
    return [os.path.join(G, n + ".sam") for n in ("consen1", "consen1b", "consen1c", "consen2", "consen3", "consen4")] + [sam, rich]


def _run(client, args, env=None):
    p = subprocess.run([client] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-800:]
    tally = [l for l in p.stderr.decode().split("\n") if l.startswith("# init")][0].split()
    assert tally[2] == tally[4]                      # every seq_init is matched by a seq_free
    return p.stdout


def test_pileup_loop_columns_equal_the_oracle(client, oracle_bin, inputs):
    for sam in inputs:
        want = subprocess.run([oracle_bin, "consensus", "-m", "simple", "-f", "dump", sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert len(want) > 100
        for batch in ("40", "3000", "65536"):
            got = _run(client, [sam], {"STA_PLP_BATCH": batch})
            assert got == want, "%s batch %s" % (os.path.basename(sam), batch)


def test_pileup_loop_early_abort_frees_every_read(client, oracle_bin, inputs):
    sam = inputs[-2]
    want = subprocess.run([oracle_bin, "consensus", "-m", "simple", "-f", "dump", sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got = _run(client, ["-s", "77", sam], {"STA_PLP_BATCH": "50"})
    assert got == b"\n".join(want.split(b"\n")[:77]) + b"\n"
