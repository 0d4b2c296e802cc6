"""The bam_plp_* / bam_mplp_* / bam_plbuf_* surface (include/samtools_amd_plp.h) against the CPU oracle's
restated HTSlib iterator: every bam_pileup1_t field of every column must match.  `plpdump` is a small
client of the callback surface on both sides (samtools_amd/csrc/driver_plpdump.cpp, oracle/o_plpdump.c).
Needs a GPU: -m gpu."""
import os
import subprocess

import pytest

from synth import write_synth_sam

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = {
    "mpileup1": [os.path.join(G, "dat", "mpileup.1.sam")],
    "three_files": [os.path.join(G, "dat", "mpileup.%d.sam" % i) for i in (1, 2, 3)],
    "overlap50": [os.path.join(G, "mpileup", "overlap50.sam")],
    "overlap_bam": [os.path.join(G, "mpileup", "overlap.bam")],
    "clip_bam": [os.path.join(G, "mpileup", "c1#clip.bam")],
    "pad_bam": [os.path.join(G, "mpileup", "c1#pad2.bam")],
    "mp_D": [os.path.join(G, "mpileup", "mp_D.sam")],
    "mp_DI": [os.path.join(G, "mpileup", "mp_DI.sam")],
    "mp_I": [os.path.join(G, "mpileup", "mp_I.sam")],
    "mp_ID": [os.path.join(G, "mpileup", "mp_ID.sam")],
    "mp_N": [os.path.join(G, "mpileup", "mp_N.sam")],
    "mp_N2": [os.path.join(G, "mpileup", "mp_N2.sam")],
    "mp_P": [os.path.join(G, "mpileup", "mp_P.sam")],
    "depth3": [os.path.join(G, "mpileup", "xx#depth3.sam")],
    "unmap_bam": [os.path.join(G, "mpileup", "ce#unmap1.bam")],
}
MODES = {
    "auto": ([], {}),
    "no_overlaps": (["-x"], {}),
    "push_plbuf": (["-p"], {}),
    "tiny_windows": ([], {"STA_PLP_BATCH": "3"}),
    "small_windows_push": (["-p"], {"STA_PLP_BATCH": "17"}),
}


def run_both(oracle_bin, product_bin, args, env_extra):
    want = subprocess.run([oracle_bin, "plpdump"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    env = dict(os.environ); env.update(env_extra)
    got = subprocess.run([product_bin, "plpdump"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == 0, got.stderr.decode()[-500:]
    if got.stdout != want:
        g, w = got.stdout.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(g, w)):
            if a != b:
                pytest.fail("column record %d differs\n got: %r\nwant: %r" % (i + 1, a[:400], b[:400]))
        pytest.fail("record count differs: got %d want %d" % (len(g), len(w)))
    return len(want)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("name", sorted(FILES))
def test_plp_surface_equals_oracle_iterator(oracle_bin, product_bin, name, mode):
    flags, env = MODES[mode]
    files = FILES[name]
    if "-p" in flags:
        files = files[:1]
    run_both(oracle_bin, product_bin, flags + files, env)


@pytest.mark.parametrize("batch", ["50", "4000", None])
def test_plp_surface_on_synthetic_pairs(tmp_path, oracle_bin, product_bin, batch):
    """30x paired 150 bp reads with indels: mate-overlap adjusted qualities and every column entry, across window sizes."""
    sam, _ = write_synth_sam(str(tmp_path), n_ref=15000, depth=30, read_len=150, seed=71, paired=True, indel_rate=0.1, max_indel=6)
    run_both(oracle_bin, product_bin, [sam], {"STA_PLP_BATCH": batch} if batch else {})


def test_plp_surface_max_depth_cap(oracle_bin, product_bin):
    """bam_plp_set_maxcnt: reads arriving at a start position once more than maxcnt are live are dropped (47.out rule)."""
    run_both(oracle_bin, product_bin, ["-d", "8500", os.path.join(G, "mpileup", "deep.sam")], {})
    run_both(oracle_bin, product_bin, ["-x", "-d", "20", os.path.join(G, "dat", "mpileup.1.sam")], {})


def test_plp_surface_overlap_visibility_at_scale(tmp_path, oracle_bin, product_bin):
    """150 000 columns of 30x pairs with frequent indels: b->qual[] of every entry must be what HTSlib would show AT THAT
    COLUMN (a pair is resolved when its second mate is pushed; deletion placeholders before the mate's start can tell)."""
    sam, _ = write_synth_sam(str(tmp_path), n_ref=150000, depth=30, read_len=150, seed=106, paired=True, indel_rate=0.1, max_indel=7)
    run_both(oracle_bin, product_bin, [sam], {})
    run_both(oracle_bin, product_bin, [sam], {"STA_PLP_BATCH": "3000"})
