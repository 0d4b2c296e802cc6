"""The drop-in boundary exercised from OUTSIDE the library: tests/cabi/plp_client.c is plain C99, built
with `gcc -std=c99 -DSTA_PLP_DROPIN -Iinclude ... -lsamtools_amd` (no private headers), written like the reference's own
small pileup clients (bedcov.c:303-335 pull loop, bam_plbuf.c:40-69 push loop, bam_plcmd.c:119 bam_plp_insertion_mod).
Every bam_pileup1_t field of every column it is handed must equal the oracle's restated HTSlib iterator.  -m gpu."""
import os
import subprocess

import pytest

from cabi_client import build_client, build_cons_client
from product_paths import REPO, lib_dir
from synth import write_synth_sam

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMS = {
    "mpileup1": [os.path.join(G, "dat", "mpileup.1.sam")],
    "three_files": [os.path.join(G, "dat", "mpileup.%d.sam" % i) for i in (1, 2, 3)],
    "overlap50": [os.path.join(G, "mpileup", "overlap50.sam")],
    "mp_D": [os.path.join(G, "mpileup", "mp_D.sam")], "mp_DI": [os.path.join(G, "mpileup", "mp_DI.sam")],
    "mp_I": [os.path.join(G, "mpileup", "mp_I.sam")], "mp_ID": [os.path.join(G, "mpileup", "mp_ID.sam")],
    "mp_N2": [os.path.join(G, "mpileup", "mp_N2.sam")], "mp_P": [os.path.join(G, "mpileup", "mp_P.sam")],
    "depth3": [os.path.join(G, "mpileup", "xx#depth3.sam")],
}


@pytest.fixture(scope="module")
def client(tmp_path_factory):
    return build_client(tmp_path_factory.mktemp("cabi"))


def _diff(oracle_bin, client, args, env=None):
    want = subprocess.run([oracle_bin, "plpdump"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got = subprocess.run([client] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    assert got.returncode == 0, got.stderr.decode()[-500:]
    if got.stdout != want:
        for i, (a, b) in enumerate(zip(got.stdout.split(b"\n"), want.split(b"\n"))):
            assert a == b, "column record %d differs\n got: %r\nwant: %r" % (i + 1, a[:300], b[:300])
        assert len(got.stdout) == len(want)
    assert len(want) > 0


@pytest.mark.parametrize("mode", [[], ["-x"], ["-p"]], ids=["pull", "pull_no_overlaps", "push_plbuf"])
@pytest.mark.parametrize("name", sorted(SAMS))
def test_external_client_sees_htslib_columns(oracle_bin, client, name, mode):
    files = SAMS[name][:1] if "-p" in mode else SAMS[name]
    _diff(oracle_bin, client, mode + files)


def test_external_client_depth_cap_and_window_sizes(tmp_path, oracle_bin, client):
    _diff(oracle_bin, client, ["-x", "-d", "20", os.path.join(G, "dat", "mpileup.1.sam")])
    sam, _ = write_synth_sam(str(tmp_path), n_ref=15000, depth=30, read_len=150, seed=72, paired=True, indel_rate=0.1, max_indel=6)
    for batch in ("40", "3000", None):
        _diff(oracle_bin, client, [sam], {"STA_PLP_BATCH": batch} if batch else None)


@pytest.mark.parametrize("flag,golden", [("-M", "mp2.out"), ("-N", "mp2-noins.out")], ids=["output_mods", "no_output_ins_mods"])
def test_external_client_prints_the_reference_mods_goldens(client, flag, golden):
    """`mpileup -x -Q0 --output-mods [--no-output-ins-mods] mod1.sam` (mpileup.reg: mp2.out, mp2-noins.out) reproduced by the EXTERNAL
    client through the drop-in names alone: a modification state per read from the constructor hook (hts_base_mod_state_alloc +
    bam_parse_basemod, bam_plcmd.c:356-369), bam_mods_at_qpos behind every base (:86-109), bam_plp_insertion_mod with the live state
    (:119).  The expected text is the reference's own file."""
    want = open(os.path.join(G, "mpileup", "expected", golden), "rb").read()
    got = subprocess.run([client, flag, "-x", os.path.join(G, "mpileup", "mod1.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()[-500:]
    if got.stdout != want:
        for i, (a, b) in enumerate(zip(got.stdout.split(b"\n"), want.split(b"\n"))):
            assert a == b, "line %d differs\n got: %r\nwant: %r" % (i + 1, a[:300], b[:300])
    assert got.stdout == want
    assert b"[+m" in want and b"+3" in want


def test_insertion_mod_refuses_a_state_that_was_never_parsed(tmp_path):
    """a state straight from hts_base_mod_state_alloc (never handed to bam_parse_basemod) must not read as "no modifications": < 0"""
    src = tmp_path / "t.c"
    src.write_text("""
#include <stdio.h>
#include <string.h>
#include "samtools_amd_plp.h"
int main(void) {
    hts_base_mod_state *m = hts_base_mod_state_alloc();
    bam1_t b; bam_pileup1_t p; kstring_t ks = {0, 0, NULL}; int dl = 0;
    unsigned char data[32];
    memset(&b, 0, sizeof b); memset(&p, 0, sizeof p); memset(data, 0, sizeof data);
    b.data = data; b.l_data = 16; b.core.l_qname = 4; b.core.n_cigar = 2; b.core.l_qseq = 2;
    ((uint32_t *)(data + 4))[0] = 1u << 4 | 0; ((uint32_t *)(data + 4))[1] = 1u << 4 | 1;      /* 1M1I */
    p.b = &b; p.indel = 1; p.cigar_ind = 0;
    if (!m) return 2;
    printf("%d\\n", bam_plp_insertion_mod(&p, m, &ks, &dl) < 0 ? 1 : 0);
    if (bam_parse_basemod(&b, m) != 0) return 3;
    printf("%d\\n", bam_plp_insertion_mod(&p, m, &ks, &dl));
    hts_base_mod_state_free(m);
    return 0;
}
""")
    exe = str(tmp_path / "t")
    lib = lib_dir()
    subprocess.run(["gcc", "-std=c99", "-DSTA_PLP_DROPIN", "-I", os.path.join(REPO, "include"), str(src), "-L", lib, "-lsamtools_amd",
                    "-Wl,-rpath," + lib, "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.split() == [b"1", b"1"], p.stdout            # refused; after parsing: one inserted base
    assert b"bam_parse_basemod" in p.stderr


# ---- the consensus iterator: pileup_loop() (include/samtools_amd_cons.h) from an external C99 client ----

@pytest.fixture(scope="module")
def cons_client(tmp_path_factory):
    return build_cons_client(tmp_path_factory.mktemp("cabi_cons"))


def _cons_diff(oracle_bin, client, sam, env=None, extra=()):
    want = subprocess.run([oracle_bin, "consensus", "-m", "simple", "-f", "dump", sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got = subprocess.run([client] + list(extra) + [sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    assert got.returncode == 0, got.stderr.decode()[-500:]
    tally = [l for l in got.stderr.decode().split("\n") if l.startswith("# init")]
    assert tally, got.stderr.decode()[-300:]
    f = tally[0].split()
    assert f[2] == f[4] and int(f[2]) > 0, tally          # every seq_init is matched by a seq_free
    return got.stdout, want


@pytest.mark.parametrize("name", ["consen1", "consen1b", "consen1c", "consen2", "consen3", "consen4"])
def test_pileup_loop_client_sees_the_reference_columns(oracle_bin, cons_client, name):
    """every pileup_t field of every column (insertion columns, pads, deletions) of the reference's consensus test inputs"""
    got, want = _cons_diff(oracle_bin, cons_client, os.path.join(G, "consensus", name + ".sam"))
    assert got == want


def test_pileup_loop_client_on_synthetic_reads_and_small_batches(tmp_path, oracle_bin, cons_client):
    from synth_rich import write_rich_sam
    sam, _ = write_synth_sam(str(tmp_path), n_ref=15000, depth=30, read_len=150, seed=73, paired=True, indel_rate=0.3, max_indel=6)
    os.makedirs(str(tmp_path / "rich"), exist_ok=True)
    rich, _ = write_rich_sam(str(tmp_path / "rich"), seed=12, n_templates=1500)
    for path in (sam, rich):
        for batch in ("40", "3000", None):
            got, want = _cons_diff(oracle_bin, cons_client, path, {"STA_PLP_BATCH": batch} if batch else None)
            if got != want:
                for i, (a, b) in enumerate(zip(got.split(b"\n"), want.split(b"\n"))):
                    assert a == b, "%s batch %s: column record %d differs\n got: %r\nwant: %r" % (os.path.basename(path), batch, i + 1, a[:300], b[:300])
                assert len(got) == len(want)
            assert len(want) > 100000


def test_pileup_loop_early_abort(tmp_path, oracle_bin, cons_client):
    """seq_column returning 1 stops the loop (consensus_pileup.c:421-422): exactly the first 100 columns, every read freed"""
    sam, _ = write_synth_sam(str(tmp_path), n_ref=5000, depth=20, read_len=100, seed=74, paired=False)
    got, want = _cons_diff(oracle_bin, cons_client, sam, extra=("-s", "100"))
    assert got == b"\n".join(want.split(b"\n")[:100]) + b"\n"
