"""The drivers' own BGZF block decoder (samtools_amd/csrc/host_inflate.h: raw DEFLATE + CRC-32) against zlib.
It replaces zlib on the decode threads (two thirds of the time per block); zlib stays the authority -- a block the fast decoder does
not deliver with the right size and CRC is decoded again by zlib -- so these tests check (1) identity wherever zlib accepts a stream:
every block of the test inputs, generated streams of every block type / level / strategy, damaged streams; (2) memory safety on
damaged input (AddressSanitizer + UBSan build); (3) that the reader gives the same records either way."""
import glob
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("inflate") / "inflate_check")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
           os.path.join(HERE, "cpu", "inflate_check.cpp"), os.path.join(REPO, "samtools_amd", "csrc", "host_inflate.cpp"), "-lz"]
    subprocess.run(cmd, check=True)
    return exe


def test_every_block_of_the_test_inputs(checker):
    files = sorted(glob.glob(os.path.join(HERE, "golden", "*", "*.bam")))
    assert len(files) >= 5
    out = subprocess.run([checker] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-400:]
    assert b"bytes identical" in out.stdout


def test_generated_streams_of_every_kind(checker):
    out = subprocess.run([checker, "--gen", "600", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-400:]


def test_damaged_streams_never_crash_and_agree_where_zlib_accepts(checker):
    bam = os.path.join(HERE, "golden", "mpileup", "mpileup.1.bam")
    out = subprocess.run([checker, "--fuzz", "4000", "5", bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-400:]
    assert b"zlib rejected" in out.stdout


def test_reader_gives_the_same_records_with_and_without_the_fast_decoder(tmp_path):
    """checksum over every decoded field (sta_io_scan), fast decoder vs STA_INFLATE=zlib, several thread counts and block sizes"""
    from bamio import sam_to_bam
    sam = os.path.join(HERE, "golden", "dat", "mpileup.1.sam")
    code = ("import sys; sys.path.insert(0, %r); from samtools_amd import _capi; "
            "print(_capi.io_scan(sys.argv[1], int(sys.argv[2]), False), _capi.io_scan(sys.argv[1], int(sys.argv[2]), 2))" % REPO)
    for block, level in ((0xff00, 1), (3000, 6), (0xff00, 0)):
        bam = sam_to_bam(sam, str(tmp_path / ("b%d_%d.bam" % (block, level))), level=level, block=block)
        seen = set()
        for env_extra in ({}, {"STA_INFLATE": "zlib"}):
            for threads in ("1", "5"):
                p = subprocess.run([os.environ.get("PYTHON", "python"), "-c", code, bam, threads], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, **env_extra))
                assert p.returncode == 0, p.stderr.decode()[-300:]
                seen.add(p.stdout)
        assert len(seen) == 1, (block, level)
