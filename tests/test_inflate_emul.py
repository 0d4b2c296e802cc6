"""The DEFLATE decoder of the device inflate kernel (samtools_amd/csrc/inflate_core.h, shared by k_bgzf_inflate and this CPU
harness) against zlib: every BGZF block of the reference's BAM fixtures, generated streams of every zlib level / strategy /
size, damaged streams (must fail cleanly; whatever is accepted equals zlib's result), built with AddressSanitizer + UBSan.
The BGZF header walk (sta_bgzf_scan) is host code of the library and is checked here too.  SURVEY.md 8(f)-2."""
import glob
import gzip
import os
import subprocess
import zlib

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("inflate") / "inflate_emul")
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    os.path.join(REPO, "tests", "cpu", "inflate_emul.cpp"), os.path.join(REPO, "samtools_amd", "csrc", "bgzf_scan.cpp"),
                    "-o", exe, "-lz"], check=True)
    return exe


def test_decoder_matches_zlib_on_every_block_of_the_reference_bams(emul):
    bams = sorted(glob.glob(os.path.join(GOLD, "*", "*.bam")))
    assert len(bams) >= 5
    p = subprocess.run([emul] + bams, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"files OK" in p.stdout, (p.stdout + p.stderr).decode()[-800:]


def test_decoder_on_generated_and_damaged_streams(emul):
    p = subprocess.run([emul, "--synth", "1200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"synth OK" in p.stdout, (p.stdout + p.stderr).decode()[-800:]


def test_damaged_bgzf_files_never_crash_the_header_walk_or_the_decoder(emul):
    p = subprocess.run([emul, "--fuzz", os.path.join(GOLD, "mpileup", "mpileup.1.bam"), "400"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"fuzz OK" in p.stdout, (p.stdout + p.stderr).decode()[-800:]


def test_decoder_on_a_bam_written_at_every_compression_level(emul, tmp_path):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from bamio import sam_to_bam
    paths = []
    for level in (0, 1, 6, 9):
        paths.append(sam_to_bam(os.path.join(GOLD, "mpileup", "ce#5b.sam"), str(tmp_path / ("l%d.bam" % level)), level=level, block=3000 if level == 6 else 0xff00))
    p = subprocess.run([emul] + paths, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"files OK" in p.stdout, (p.stdout + p.stderr).decode()[-800:]


def test_bgzf_scan_lists_the_blocks_zlib_sees():
    import sys
    sys.path.insert(0, REPO)
    from samtools_amd import _capi
    path = os.path.join(GOLD, "mpileup", "mpileup.1.bam")
    data = open(path, "rb").read()
    blocks, n, total = _capi.bgzf_scan(data)
    assert total == len(gzip.open(path).read())
    off = 0
    for i in range(n):
        b = blocks[i]
        raw = zlib.decompress(data[b.in_off:b.in_off + b.in_len], -15)
        assert len(raw) == b.out_len and b.out_off == off and (zlib.crc32(raw) & 0xffffffff) == b.crc32
        off += b.out_len
    assert n >= 2 and blocks[n - 1].out_len == 0            # the EOF marker block
    for bad in (data[:-5], b"\x1f\x8b\x08\x00" + data[4:], data[:100]):
        with pytest.raises(RuntimeError):
            _capi.bgzf_scan(bad)
