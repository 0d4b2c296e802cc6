"""k_bgzf_inflate (kernels_inflate.hip) through the C-ABI (sta_bgzf_scan + sta_bgzf_inflate + sta_fetch_inflated): the inflated
bytes of whole BAM files -- the reference's fixtures, a BAM written at several compression levels and block sizes -- equal
Python's gzip; a damaged block is reported (count and index) and does not disturb the others.  SURVEY.md 8(f)-2.  -m gpu."""
import glob
import gzip
import os
import time

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="module")
def eng():
    from samtools_amd import _capi
    e = _capi.Engine(0)
    yield e
    e.close()


def test_reference_bams_inflate_to_what_gzip_gives(eng):
    bams = sorted(glob.glob(os.path.join(GOLD, "*", "*.bam")))
    assert len(bams) >= 5
    for p in bams:
        got, nbad, first = eng.bgzf_inflate(open(p, "rb").read())
        assert nbad == 0 and first is None, p
        assert got == gzip.open(p).read(), p


def test_levels_block_sizes_and_a_larger_file(eng, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from bamio import sam_to_bam
    from synth import write_synth_sam
    sam, _ = write_synth_sam(str(tmp_path), n_ref=80000, depth=30, read_len=150, seed=9, paired=True, indel_rate=0.02)
    for level, block in ((0, 0xff00), (1, 0xff00), (6, 0xff00), (9, 20000), (1, 700)):
        bam = sam_to_bam(sam, str(tmp_path / ("l%d_%d.bam" % (level, block))), level=level, block=block)
        data = open(bam, "rb").read()
        t0 = time.perf_counter()
        got, nbad, _ = eng.bgzf_inflate(data)
        dt = time.perf_counter() - t0
        assert nbad == 0
        assert got == gzip.open(bam).read(), (level, block)
        print("level %d block %d: %d -> %d bytes in %.1f ms (host call incl. copies)" % (level, block, len(data), len(got), dt * 1e3))


def test_a_damaged_block_is_counted_and_the_rest_is_intact(eng):
    from samtools_amd import _capi
    p = os.path.join(GOLD, "mpileup", "mpileup.1.bam")
    data = bytearray(open(p, "rb").read())
    blocks, n, total = _capi.bgzf_scan(bytes(data))
    assert n >= 3
    b = blocks[1]
    data[b.in_off + b.in_len // 2] ^= 0x10                     # somewhere inside the second block's DEFLATE data
    got, nbad, first = eng.bgzf_inflate(bytes(data))
    want = gzip.open(p).read()
    assert nbad == 1 and first == 1
    assert got[:blocks[1].out_off] == want[:blocks[1].out_off]
    assert got[blocks[2].out_off:] == want[blocks[2].out_off:]
