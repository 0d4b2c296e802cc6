"""Host input plumbing (no GPU): the drivers' SAM/BAM reader -- threaded BGZF inflate + parse-ahead -- decodes the same records
whatever the thread count, block size or container (SAM text vs BAM), and reports damaged input instead of truncating.
sta_io_scan folds every decoded field into an order-dependent checksum; stage=True also runs the window pump + SoA stager."""
import gzip
import os

import pytest

from bamio import sam_to_bam
from synth import write_synth_sam
from synth_rich import write_rich_sam

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def capi():
    from samtools_amd import _capi
    return _capi


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("hostio"))
    sam, _ = write_synth_sam(d, n_ref=60000, depth=30, read_len=150, seed=11, paired=True, indel_rate=0.02)
    rich, _ = write_rich_sam(d, seed=3, n_templates=1500)
    return d, sam, rich


def test_bam_writer_matches_a_samtools_written_bam(tmp_path):
    """tests/bamio.py is pinned on a SAM/BAM pair the reference's test suite ships: same decompressed bytes."""
    want = gzip.open(os.path.join(GOLD, "mpileup", "ce#5b.bam")).read()
    out = sam_to_bam(os.path.join(GOLD, "mpileup", "ce#5b.sam"), str(tmp_path / "x.bam"))
    assert gzip.open(out).read() == want


@pytest.mark.parametrize("which", ["synth", "rich"])
def test_same_records_from_sam_and_bam_any_threads_any_block_size(capi, files, which):
    d, sam, rich = files
    src = sam if which == "synth" else rich
    ref = capi.io_scan(src, 1, False)
    ref_staged = capi.io_scan(src, 1, True)
    assert ref[0] > 1000
    for block in (0xff00, 4096, 700):
        bam = sam_to_bam(src, os.path.join(d, "%s_%d.bam" % (which, block)), level=1, block=block)
        for threads in (1, 3, 8):
            assert capi.io_scan(bam, threads, False) == ref, (block, threads)
        assert capi.io_scan(bam, 4, True) == ref_staged, block
    # ordinary gzip of the SAM text (not BGZF) goes through the single-thread path
    gz = os.path.join(d, which + ".sam.gz")
    with open(src, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
        fo.write(fi.read())
    assert capi.io_scan(gz, 4, False) == ref


def test_reference_fixture_bams_decode_identically_with_one_and_many_workers(capi):
    n = 0
    for sub in ("mpileup", "bedcov"):
        for fn in sorted(os.listdir(os.path.join(GOLD, sub))):
            if fn.endswith(".bam"):
                p = os.path.join(GOLD, sub, fn)
                assert capi.io_scan(p, 1, False) == capi.io_scan(p, 6, False), fn
                n += 1
    assert n >= 15


def test_damaged_and_truncated_bgzf_is_an_error_not_a_short_read(capi, files):
    d, sam, _ = files
    bam = sam_to_bam(sam, os.path.join(d, "dmg.bam"), level=1, block=8192)
    raw = bytearray(open(bam, "rb").read())
    good = capi.io_scan(bam, 4, False)
    # flip a byte inside the deflate data of a block in the middle: CRC / inflate must notice
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x5a
    p = os.path.join(d, "dmg1.bam"); open(p, "wb").write(bad)
    with pytest.raises(RuntimeError):
        capi.io_scan(p, 4, False)
    # cut in the middle of a block
    p = os.path.join(d, "dmg2.bam"); open(p, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError):
        capi.io_scan(p, 4, False)
    # a missing EOF marker block alone is tolerated (samtools only warns)
    p = os.path.join(d, "noeof.bam"); open(p, "wb").write(raw[:-28])
    assert capi.io_scan(p, 4, False) == good


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


WINDOWS = [{}, {"STA_WINDOW_COLS": "900"}, {"STA_WINDOW_COLS": "211", "STA_SCAN_DROP": "1"}, {"STA_WINDOW_COLS": "5000", "STA_WINDOW_READS": "50"},
           {"STA_WINDOW_COLS": "64", "STA_WINDOW_READS": "7", "STA_SCAN_DROP": "1"}]


@pytest.mark.parametrize("which", ["synth", "rich"])
def test_chunk_lane_stages_the_same_windows_as_the_record_lane(capi, files, which):
    """host_chunk.h (records decoded on several threads into SoA chunks, windows = slice concatenations) against
    host_pump.h (one decoded record at a time): every staged array of every window, the window bounds, which reads have a
    reference span, with the mpileup driver's overlap lookahead / mate keeping switched on, tiny windows, read-cap cuts and
    simulated -d drops."""
    d, sam, rich = files
    src = sam if which == "synth" else rich
    bam = sam_to_bam(src, os.path.join(d, which + "_lane.bam"), level=1, block=3000)
    for env in WINDOWS:
        ref = _with_env(env, lambda: capi.io_scan(src, 2, 1))
        for path, threads in ((src, 1), (src, 5), (bam, 3)):
            got = _with_env(env, lambda: capi.io_scan(path, threads, 2))
            assert got == ref, (env, os.path.basename(path), threads)
    assert _with_env(WINDOWS[1], lambda: capi.io_scan(src, 2, 1)) != _with_env(WINDOWS[0], lambda: capi.io_scan(src, 2, 1))


def test_chunk_lane_with_bq_tags_and_staging_threads(capi, files):
    """BQ:Z appears only from the middle of the file on (the staged BQ pool is materialised late, '@' before it), and the
    slices of a window are copied by several threads (STA_STAGE_THREADS set: for every window, however small)."""
    d, sam, _ = files
    lines = open(sam).read().split("\n")
    body = [i for i, l in enumerate(lines) if l and not l.startswith("@")]
    for k, i in enumerate(body):
        if k > len(body) // 2 and k % 3 == 0:
            f = lines[i].split("\t")
            lines[i] += "\tBQ:Z:" + "".join(chr(64 + (j * 7 + k) % 5) for j in range(len(f[9])))
    src = os.path.join(d, "bq_mid.sam"); open(src, "w").write("\n".join(lines))
    bam = sam_to_bam(src, os.path.join(d, "bq_mid.bam"), level=1, block=5000)
    for env in ({}, {"STA_WINDOW_COLS": "900"}, {"STA_WINDOW_COLS": "5000", "STA_WINDOW_READS": "50"}):
        ref = _with_env(env, lambda: capi.io_scan(src, 2, 1))
        for st in ("1", "3"):
            e2 = dict(env, STA_STAGE_THREADS=st)
            assert _with_env(e2, lambda: capi.io_scan(src, 3, 2)) == ref, (env, st)
            assert _with_env(e2, lambda: capi.io_scan(bam, 4, 2)) == ref, (env, st)
    assert _with_env({}, lambda: capi.io_scan(src, 2, 1)) != _with_env({}, lambda: capi.io_scan(sam, 2, 1))


def test_chunk_lane_on_the_reference_fixture_bams(capi):
    for sub in ("mpileup", "bedcov"):
        for fn in sorted(os.listdir(os.path.join(GOLD, sub))):
            if fn.endswith(".bam"):
                p = os.path.join(GOLD, sub, fn)
                for env in ({}, {"STA_WINDOW_COLS": "100"}):
                    a = _with_env(env, lambda: capi.io_scan(p, 2, 1))
                    b = _with_env(env, lambda: capi.io_scan(p, 4, 2))
                    assert a == b, (fn, env)


def test_chunk_lane_reports_unsorted_and_damaged_input(capi, files):
    d, sam, _ = files
    lines = open(sam).read().split("\n")
    hdr = [l for l in lines if l.startswith("@")]
    recs = [l for l in lines if l and not l.startswith("@")]
    recs[10], recs[400] = recs[400], recs[10]
    bad = os.path.join(d, "unsorted.sam")
    open(bad, "w").write("\n".join(hdr + recs) + "\n")
    for stage in (1, 2):
        with pytest.raises(RuntimeError):
            capi.io_scan(bad, 3, stage)
    bam = sam_to_bam(sam, os.path.join(d, "dmg_lane.bam"), level=1, block=8192)
    raw = bytearray(open(bam, "rb").read()); raw[len(raw) // 2] ^= 0x5a
    p = os.path.join(d, "dmg_lane2.bam"); open(p, "wb").write(raw)
    with pytest.raises(RuntimeError):
        capi.io_scan(p, 3, 2)
    # the chunk lane reads a BAM file through a mapping and cuts the blocks itself: its own view of the file's end
    raw = open(bam, "rb").read()
    good = capi.io_scan(bam, 3, 2)
    p = os.path.join(d, "cut_mid_block.bam"); open(p, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError):
        capi.io_scan(p, 3, 2)
    # whole blocks only, but the last record unfinished (8 KiB blocks: records straddle them)
    import struct
    offs, o = [], 0
    while o + 18 <= len(raw):
        offs.append(o)
        o += struct.unpack_from("<H", raw, o + 16)[0] + 1
    cut = offs[len(offs) // 2]
    p = os.path.join(d, "cut_mid_record.bam"); open(p, "wb").write(raw[:cut])
    with pytest.raises(RuntimeError):
        capi.io_scan(p, 3, 2)
    p = os.path.join(d, "noeof_lane.bam"); open(p, "wb").write(raw[:-28])
    assert capi.io_scan(p, 3, 2) == good
    assert _with_env({"STA_CHUNK_MAP": "0"}, lambda: capi.io_scan(bam, 3, 2)) == good


def test_chunk_lane_with_several_input_files(capi, files, tmp_path):
    """Multi-file windows (bam_mplp_* style: every window receives the reads of every file; the read cap of file 0 cuts the
    window for all of them): both lanes, mixed SAM / BAM inputs."""
    d, sam, rich = files
    other, _ = write_synth_sam(str(tmp_path), n_ref=60000, depth=12, read_len=100, seed=12, paired=True)
    bam = sam_to_bam(other, os.path.join(d, "other.bam"), level=1, block=4000)
    for env in WINDOWS:
        ref = _with_env(env, lambda: capi.io_scan([sam, other, sam], 2, 1))
        assert _with_env(env, lambda: capi.io_scan([sam, bam, sam], 4, 2)) == ref, env
        assert _with_env(env, lambda: capi.io_scan([sam, other, sam], 1, 2)) == ref, env
    assert capi.io_scan([sam, other], 2, 1) != capi.io_scan([other, sam], 2, 1)


def test_both_lanes_read_standard_input(files):
    """`-` = stdin (what `samtools view ... | samtools mpileup -` relies on, mpileup.reg:102-105): goes through gzread on the
    caller's thread; records and staged windows must equal those of the file."""
    import subprocess
    import sys
    d, sam, _ = files
    bam = sam_to_bam(sam, os.path.join(d, "stdin.bam"), level=1, block=6000)
    code = ("import sys; sys.path.insert(0, %r); from samtools_amd import _capi; "
            "print(_capi.io_scan('-', 3, int(sys.argv[1])))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from samtools_amd import _capi
    for stage in (0, 1, 2):
        want = str(_capi.io_scan(sam, 3, stage))
        for path in (sam, bam):
            with open(path, "rb") as fh:
                got = subprocess.run([sys.executable, "-c", code, str(stage)], stdin=fh, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            assert got.stdout.decode().strip() == want, (stage, os.path.basename(path))


def test_producer_bench_harness_builds_and_runs(files, tmp_path):
    """tests/cpu/stage_bench.cpp (decode threads -> chunk lane -> staged windows without a device; scripts/stage_bench.sh) stays
    buildable against the host sources and walks a BAM to the end."""
    import subprocess
    d, sam, _ = files
    bam = sam_to_bam(sam, os.path.join(d, "bench_in.bam"), level=1)
    exe = str(tmp_path / "stage_bench")
    subprocess.run(["bash", os.path.join(os.path.dirname(GOLD), "..", "scripts", "stage_bench.sh"), exe], check=True)
    p = subprocess.run([exe, bam, "3", "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_NO_PINNED="1", STA_STAGE_THREADS="2"))
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert b"windows" in p.stdout and b"reads" in p.stdout


def test_region_coordinates_the_way_hts_parse_decimal_reads_them(capi, tmp_path):
    """HTSlib reads region coordinates with hts_parse_decimal: thousands commas, a fraction, an exponent and the suffixes k / M / G
    (the reference's manual itself writes `-r chr1:1M-12M`, doc/samtools-coverage.1:148)"""
    sam = tmp_path / "r.sam"
    read = "%s\t0\tchr1\t%d\t60\t50M\t*\t0\t0\t" + "A" * 50 + "\t" + "I" * 50
    sam.write_text("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:249250621\n"
                   + "\n".join(read % ("r%d" % i, p) for i, p in enumerate([999900, 1000000, 5000000, 11999999, 12000000, 12000001, 25000000])) + "\n")
    bam = sam_to_bam(str(sam), str(tmp_path / "r.bam"), level=1)
    want = capi.io_scan_region(bam, "chr1:1000000-12000000", 1, False)[:2]
    assert want[0] == 4
    for spelling in ("chr1:1M-12M", "chr1:1,000,000-12,000,000", "chr1:1e6-1.2e7", "chr1:1000k-12000K", "chr1:0.001G-0.012g", "chr1:1.0M-12.0M"):
        assert capi.io_scan_region(bam, spelling, 1, False)[:2] == want, spelling
    assert capi.io_scan_region(bam, "chr1:12M", 1, False)[0] == 4            # from 12 000 000 to the end (the read that starts one base earlier reaches in)
    assert capi.io_scan_region(bam, "chr1:-1M", 1, False)[0] == 2            # up to 1 000 000
    for bad in ("chr1:1M-12Q", "chr1:x-5", "chr1:5M-1M"):
        with pytest.raises(RuntimeError):
            capi.io_scan_region(bam, bad, 1, False)
