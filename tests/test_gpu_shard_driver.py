"""The product drivers sharded over reference columns: every rank runs sta_main_mpileup /
sta_main_depth restricted to its block (STA_SHARD=rank/world) and the concatenation of the ranks' text in rank order must be the
unsharded output, byte for byte -- with the block cuts swept through mate overlaps, long ref skips, contig boundaries and
regions.  Part 1 runs the ranks one after the other (the cuts are what is under test); part 2 is the real thing: two processes
on the one GPU of the test box under torch.distributed.run (gloo), text gathered by samtools_amd.shard.  -m gpu."""
import os
import subprocess
import sys

import pytest

from synth import write_synth_sam
from synth_rich import write_rich_sam

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(product_bin, args, env=None):
    p = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr.decode()[-500:]
    return p.stdout


def _sharded(product_bin, args, world, cuts=None, env=None):
    out = b""
    for r in range(world):
        e = dict(env or {}, STA_SHARD="%d/%d" % (r, world))
        if cuts is not None:
            e["STA_SHARD_CUTS"] = ",".join(str(c) for c in cuts)
        out += _run(product_bin, args, e)
    return out


@pytest.fixture(scope="module")
def pairs(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard_pairs")
    return write_synth_sam(str(d), n_ref=60000, depth=30, read_len=150, seed=23, paired=True, indel_rate=0.1, max_indel=7)


@pytest.fixture(scope="module")
def rich(tmp_path_factory):
    d = tmp_path_factory.mktemp("shard_rich")
    return write_rich_sam(str(d), seed=11, n_templates=6000)


@pytest.mark.parametrize("cmd", [["mpileup", "-f", "{fa}"], ["mpileup", "-E", "-A", "-f", "{fa}"], ["mpileup", "-B", "-aa", "-f", "{fa}"], ["mpileup", "-x", "-B"],
                                 ["depth"], ["depth", "-aa"], ["depth", "-s", "-J", "-q", "20"]], ids=lambda c: "_".join(x for x in c if not x.startswith("{")))
def test_equal_blocks_reassemble_the_unsharded_output(product_bin, pairs, cmd):
    sam, fa = pairs
    args = [a.format(fa=fa) for a in cmd] + [sam]
    want = _run(product_bin, args)
    assert len(want) > 100000
    for world in (2, 3, 8):
        assert _sharded(product_bin, args, world) == want, "world %d" % world
    # small windows inside the blocks as well
    assert _sharded(product_bin, args, 3, env={"STA_WINDOW_COLS": "7000"}) == want


def test_cuts_swept_through_mate_overlaps(product_bin, pairs):
    """every cut position in a stretch of 400 columns (pairs overlap there in every phase), overlap resolution + BAQ on"""
    sam, fa = pairs
    args = ["mpileup", "-f", fa, sam]
    want = _run(product_bin, args)
    for cut in list(range(30000, 30400, 7)) + [1, 149, 150, 59999]:
        assert _sharded(product_bin, args, 2, cuts=[cut]) == want, "cut %d" % cut


@pytest.mark.parametrize("cmd", [["mpileup", "-f", "{fa}"], ["mpileup", "-aa", "-B", "-f", "{fa}"], ["depth", "-aa"], ["depth"]],
                         ids=lambda c: "_".join(x for x in c if not x.startswith("{")))
def test_messy_multi_contig_input(product_bin, rich, cmd):
    """several contigs, long ref skips, clips, pads, every flag: blocks cut inside and between contigs"""
    sam, fa = rich
    args = [a.format(fa=fa) for a in cmd] + [sam]
    want = _run(product_bin, args)
    for world in (2, 5):
        assert _sharded(product_bin, args, world) == want
    for cuts in ([29999], [30000], [30001], [39000], [38999, 39001], [100, 83990]):
        assert _sharded(product_bin, args, len(cuts) + 1, cuts=cuts) == want, cuts
    # a region is its own coordinate space
    rargs = args[:-1] + ["-r", "c3:5000-30000", sam]
    rwant = _run(product_bin, rargs)
    for world in (2, 3):
        assert _sharded(product_bin, rargs, world) == rwant


@pytest.mark.parametrize("cmd", [["mpileup", "-C", "50", "-f", "{fa}"], ["mpileup", "-C", "20", "-aa", "-E", "-f", "{fa}"]], ids=["C50", "C20_aa_E"])
def test_capq_with_overlap_detection_runs_the_windows_in_front_of_a_block_on_the_device(product_bin, pairs, rich, cmd):
    """-C: sam_cap_mapq reads BAQ-adjusted qualities, so only the device knows who reaches the overlap hash; a rank plans the windows
    in front of its block (text discarded) instead of passing them over on the host -- refused until round 6 (found by scripts/hunt6.py 612)"""
    for sam, fa in (pairs, rich):
        args = [a.format(fa=fa) for a in cmd] + [sam]
        want = _run(product_bin, args)
        assert len(want) > 100000
        for world, env in ((2, None), (5, {"STA_WINDOW_COLS": "300", "STA_WINDOW_READS": "5"})):
            assert _sharded(product_bin, args, world, env=env) == want, (world, env)
        for cuts in ([30173], [13199], [100, 39001, 83990]):
            assert _sharded(product_bin, args, len(cuts) + 1, cuts=cuts, env={"STA_WINDOW_COLS": "10000"}) == want, cuts


def test_state_that_crosses_blocks_is_refused(product_bin, pairs):
    sam, fa = pairs
    for args, msg in ((["mpileup", "-a", "-f", fa, sam], b"single -a"), (["depth", "-a", sam], b"single -a"), (["mpileup", "-d", "10", "-B", sam], b"depth cap")):
        p = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_SHARD="1/2"))
        assert p.returncode != 0 and msg in p.stderr, (args, p.stderr[-200:])


@pytest.mark.parametrize("cmd", [["mpileup", "-f", "{fa}"], ["depth", "-aa"]], ids=["mpileup", "depth"])
def test_two_processes_one_gather(tmp_path, product_bin, pairs, cmd):
    """python -m torch.distributed.run --nproc-per-node 2 -m samtools_amd.shard <command>: one process per rank (both on the
    box's single GPU), text collected on rank 0 by the variable-size gather."""
    sam, fa = pairs
    args = [a.format(fa=fa) for a in cmd] + [sam]
    want = _run(product_bin, args)
    outp = str(tmp_path / "sharded.txt")
    env = dict(os.environ, STA_SHARD_BACKEND="gloo", STA_SHARD_ONE_DEVICE="1", PYTHONPATH=REPO)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533" if cmd[0] == "mpileup" else "29534", "-m", "samtools_amd.shard"] + args + ["-o", outp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert open(outp, "rb").read() == want


@pytest.mark.parametrize("cmd", [["mpileup", "-B", "-f", "{fa}"], ["depth", "-aa"]], ids=["mpileup_B", "depth"])
def test_three_processes_write_their_blocks_in_place(tmp_path, product_bin, pairs, cmd):
    """STA_SHARD_PWRITE=1 with -o FILE: no gather -- the ranks exchange their byte counts and each writes its block at its offset of the file
    (the shapes without BAQ produce text seven times faster than one xGMI link carries it: DESIGN.md section 6)."""
    sam, fa = pairs
    args = [a.format(fa=fa) for a in cmd] + [sam]
    want = _run(product_bin, args)
    outp = str(tmp_path / "inplace.txt")
    open(outp, "wb").write(b"x" * (len(want) + 1000))            # (an older, longer file: it is truncated to the new size)
    env = dict(os.environ, STA_SHARD_BACKEND="gloo", STA_SHARD_ONE_DEVICE="1", STA_SHARD_PWRITE="1", STA_SHARD_TIMING="1", PYTHONPATH=REPO)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29541" if cmd[0] == "mpileup" else "29542", "-m", "samtools_amd.shard"] + args + ["-o", outp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stderr.count(b"written in place") == 3
    assert open(outp, "rb").read() == want


@pytest.mark.parametrize("cmd", [["mpileup", "-f", "{fa}"], ["depth", "-aa"]], ids=["mpileup", "depth"])
def test_one_process_over_rccl_with_device_capture(tmp_path, product_bin, pairs, cmd):
    """The RCCL form of the product launcher on the hardware there is: `torch.distributed.run --nproc-per-node 1 -m samtools_amd.shard`
    with the default backend (nccl = RCCL).  One GPU cannot hold two RCCL ranks, so the gather's send / receive pair is not reached,
    but the communicator, the exit-status reduction and the size all-gather on device tensors, and the device capture feeding the
    gather's tensor (sta_main_capture_device -> sta_capture_device_take) all execute; the text is the unsharded text."""
    sam, fa = pairs
    args = [a.format(fa=fa) for a in cmd] + [sam]
    want = _run(product_bin, args)
    outp = str(tmp_path / "rccl1.txt")
    env = dict(os.environ, PYTHONPATH=REPO, STA_SHARD_TIMING="1")
    env.pop("STA_SHARD_BACKEND", None); env.pop("STA_SHARD_HOST_CAPTURE", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29535" if cmd[0] == "mpileup" else "29536", "-m", "samtools_amd.shard"] + args + ["-o", outp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert b"(device capture)" in p.stderr, p.stderr.decode()[-400:]
    assert open(outp, "rb").read() == want


def test_four_processes_unequal_blocks_against_the_oracle(tmp_path, oracle_bin, pairs):
    """world size 4 under torch.distributed.run (gloo, all on the box's one GPU), unequal blocks (STA_SHARD_CUTS: one cut inside a
    mate overlap, one a column behind it, one near the end), BAM input with a .bai beside it -- every rank starts from the
    index -- and the gathered text is compared with the ORACLE's, not with the unsharded engine."""
    from bamio import sam_to_bam, write_bai
    sam, fa = pairs
    bam = sam_to_bam(sam, str(tmp_path / "p.bam"), block=20000)
    write_bai(bam)
    for cmd, port in ((["mpileup", "-f", fa], "29541"), (["depth", "-aa"], "29542")):
        want = subprocess.run([oracle_bin] + cmd + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        outp = str(tmp_path / ("sharded_%s.txt" % cmd[0]))
        env = dict(os.environ, STA_SHARD_BACKEND="gloo", STA_SHARD_ONE_DEVICE="1", PYTHONPATH=REPO, STA_SHARD_CUTS="30100,30101,58000")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                            "--master-port", port, "-m", "samtools_amd.shard"] + cmd + [bam, "-o", outp],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=REPO)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert open(outp, "rb").read() == want, cmd


def test_indexed_and_unindexed_sharded_and_region_runs_agree(tmp_path, product_bin, oracle_bin, rich):
    """a .bai beside the BAM changes where the readers start (linear index offset instead of the file's first record) and where
    they stop, never the text: blocks, regions and both together, against the same runs with STA_NO_INDEX=1 and the oracle."""
    from bamio import sam_to_bam, write_bai
    sam, fa = rich
    bam = sam_to_bam(sam, str(tmp_path / "r.bam"), block=9000)
    write_bai(bam)
    for cmd in (["mpileup", "-f", fa], ["depth", "-aa"]):
        want = _run(product_bin, cmd + [sam])
        assert subprocess.run([oracle_bin] + cmd + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout == want
        for world in (3, 7):
            assert _sharded(product_bin, cmd + [bam], world) == want
            assert _sharded(product_bin, cmd + [bam], world, env={"STA_NO_INDEX": "1"}) == want
        for reg in ("c1:100-20000", "c2", "c3:20000-44999", "c3:44000"):
            rwant = subprocess.run([oracle_bin] + cmd + ["-r", reg, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            assert _run(product_bin, cmd + ["-r", reg, bam]) == rwant, (cmd, reg)
            assert _run(product_bin, cmd + ["-r", reg, bam], {"STA_NO_INDEX": "1"}) == rwant, (cmd, reg)
            assert _sharded(product_bin, cmd + ["-r", reg, bam], 2) == rwant, (cmd, reg)


def test_customized_index_names_the_index_files(tmp_path, product_bin, oracle_bin, rich):
    """-X / --customized-index (bam_plcmd.c:1190,1243-1262; bam2depth.c:873-911): the second half of the file arguments names the index of
    each input.  The indexes live in another directory under other names; region runs of one and of two inputs give the oracle's text,
    and a trace of the reader's start shows the named index was used (STA_NO_INDEX=1 gives the same text from a whole-file read)."""
    import shutil
    from bamio import sam_to_bam, write_bai
    sam, fa = rich
    bam = sam_to_bam(sam, str(tmp_path / "x.bam"), block=9000)
    write_bai(bam)
    os.mkdir(str(tmp_path / "idx"))
    ix1 = str(tmp_path / "idx" / "first.index"); ix2 = str(tmp_path / "idx" / "second.index")
    shutil.move(bam + ".bai", ix1); shutil.copy(ix1, ix2)
    bam2 = str(tmp_path / "x2.bam"); shutil.copy(bam, bam2)
    for cmd in (["mpileup", "-f", fa], ["depth", "-aa"]):
        for reg in ("c2:3000-9000", "c3:44000"):
            want1 = subprocess.run([oracle_bin] + cmd + ["-r", reg, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            want2 = subprocess.run([oracle_bin] + cmd + ["-r", reg, sam, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            assert _run(product_bin, cmd + ["-r", reg, "-X", bam, ix1]) == want1, (cmd, reg)
            long_name = "--customized-index" if cmd[0] == "mpileup" else "-X"       # (depth has only the short form: bam2depth.c:873)
            assert _run(product_bin, cmd + ["-r", reg, long_name, bam, bam2, ix1, ix2]) == want2, (cmd, reg)
            assert _run(product_bin, cmd + ["-r", reg, "-X", bam, ix1], {"STA_NO_INDEX": "1"}) == want1
    # an odd number of names is refused, as is -X with a file list
    p = subprocess.run([product_bin, "mpileup", "-X", bam, bam2, ix1], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"Odd number of filenames" in p.stderr
    lst = str(tmp_path / "list.txt"); open(lst, "w").write(bam + "\n")
    p = subprocess.run([product_bin, "mpileup", "-X", "-b", lst], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"cannot be combined with -X" in p.stderr
    # a file that is not a BAI: a warning, and the whole file is read (same text)
    want = subprocess.run([oracle_bin, "depth", "-r", "c2:3000-9000", sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    p = subprocess.run([product_bin, "depth", "-r", "c2:3000-9000", "-X", bam, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == want and b"could not load the index" in p.stderr


@pytest.mark.parametrize("cmd", [["mpileup", "-f", "{fa}"], ["mpileup", "-B", "-aa", "-f", "{fa}"], ["depth", "-H", "-aa"], ["depth", "-s", "-J"]],
                         ids=lambda c: "_".join(x for x in c if not x.startswith("{")))
def test_device_capture_is_the_host_capture(pairs, cmd):
    """sta_main_capture_device (the window text stays in device memory, what a sharded run on GPUs hands to the gather -- VERDICT r03
    item 8) == sta_main_capture byte for byte, through a device buffer the caller owns; small windows force many appends and growth."""
    import torch
    from samtools_amd import _capi
    sam, fa = pairs
    args = [a.format(fa=fa) for a in cmd[1:]] + [sam]
    os.environ["STA_WINDOW_COLS"] = "4096"
    try:
        rc_h, want = _capi.main_capture(cmd[0], args)
        rc_d, head, n_dev = _capi.main_capture_device(cmd[0], args)
    finally:
        os.environ.pop("STA_WINDOW_COLS", None)
    assert rc_h == 0 and rc_d == 0 and len(head) + n_dev == len(want) and n_dev > 0
    t = torch.empty(n_dev, dtype=torch.uint8, device="cuda")
    _capi.capture_device_take(t.data_ptr(), n_dev)
    assert head + t.cpu().numpy().tobytes() == want
    with pytest.raises(RuntimeError):
        _capi.capture_device_take(t.data_ptr(), n_dev)          # (taken once)
