"""BGZF blocks inflated on the device (csrc/kernels_inflate.hip, one wave per block) == zlib's inflate, byte for byte.

What it replaces: HTSlib bgzf.c inflate_block() under sam_read1 (bam_plcmd.c:409, bam2depth.c:541-543).  The same wave functions
run on the CPU in tests/test_inflate_emul.py; here the kernel itself: every deflate block type (stored, fixed, dynamic) from zlib
levels 0-9 and the Z_FIXED / Z_HUFFMAN_ONLY / Z_RLE strategies, sizes from 0 to the BGZF maximum, the blocks of the repository's
golden BAM files, damaged streams (non-zero status or -- caught by the CRC the caller checks -- other bytes, never a crash), and
a throughput figure on BAM-like data."""
import glob
import os
import struct
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def _kinds(rng, n):
    yield rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    yield rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()
    yield bytes([int(rng.integers(0, 256))]) * n
    per = int(rng.integers(2, 70))
    pat = rng.integers(0, 256, per, dtype=np.uint8).tobytes()
    yield (pat * (n // per + 1))[:n]
    q = np.where(rng.random(n) < 0.9, 37, rng.integers(2, 40, n)).astype(np.uint8)
    yield q.tobytes()
    # stretches repeated from 17 000 .. 30 000 bytes back: matches that reach behind the decoder's 16 KiB LDS ring
    far = rng.integers(0, 256, n, dtype=np.uint8)
    i = 30000
    while i < n:
        back, run = int(rng.integers(17000, 30000)), int(rng.integers(20, 400))
        far[i:i + run] = far[i - back:i - back + run][:max(0, min(run, n - i))]
        i += run + int(rng.integers(0, 50))
    yield far.tobytes()


@pytest.mark.gpu
def test_every_block_type_and_size():
    from samtools_amd import _capi
    rng = np.random.default_rng(5)
    streams, sizes, want = [], [], []
    for n in (0, 1, 2, 17, 255, 256, 4096, 40000, 65279, 65280):
        for data in _kinds(rng, n):
            for level, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                 (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
                streams.append(_deflate(data, level, strat)); sizes.append(n); want.append(data)
    got, status, _ = _capi.bgzf_inflate(streams, sizes)
    assert all(s == 0 for s in status), [i for i, s in enumerate(status) if s][:10]
    assert got == want


def _bgzf_blocks(path):
    raw = open(path, "rb").read()
    o, out = 0, []
    while o + 18 <= len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        bsize = None
        x = o + 12
        while x < o + 12 + xlen:
            si, sl = raw[x:x + 2], struct.unpack_from("<H", raw, x + 2)[0]
            if si == b"BC":
                bsize = struct.unpack_from("<H", raw, x + 4)[0] + 1
            x += 4 + sl
        data = raw[o + 12 + xlen:o + bsize - 8]
        crc, isize = struct.unpack_from("<II", raw, o + bsize - 8)
        out.append((data, isize, crc))
        o += bsize
    return out


@pytest.mark.gpu
def test_blocks_of_the_golden_bam_files():
    from samtools_amd import _capi
    files = sorted(glob.glob(os.path.join(HERE, "golden", "**", "*.bam"), recursive=True))
    assert files
    blocks = [b for f in files for b in _bgzf_blocks(f)]
    got, status, _ = _capi.bgzf_inflate([b[0] for b in blocks], [b[1] for b in blocks])
    assert all(s == 0 for s in status)
    for g, (data, isize, crc) in zip(got, blocks):
        assert len(g) == isize and (zlib.crc32(g) & 0xffffffff) == crc
        assert g == zlib.decompress(data, -15)


@pytest.mark.gpu
def test_damaged_streams_end_with_a_status_or_other_bytes_never_a_crash():
    from samtools_amd import _capi
    rng = np.random.default_rng(9)
    base = (b"read_%07d" % 7 + bytes(range(64)) + b"I" * 150) * 200
    streams, sizes, good = [], [], []
    for k in range(300):
        data = base[:int(rng.integers(100, 60000))]
        s = bytearray(_deflate(data, int(rng.integers(1, 10))))
        for _ in range(int(rng.integers(1, 4))):
            s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(s)); sizes.append(len(data)); good.append(data)
    got, status, _ = _capi.bgzf_inflate(streams, sizes)
    caught = sum(1 for s in status if s)
    assert caught > 100
    for g, s, d in zip(got, status, good):
        assert (g is None) == (s != 0)
    # and an undamaged stream right behind them still decodes
    got, status, _ = _capi.bgzf_inflate([_deflate(base[:40000], 6)], [40000])
    assert status == [0] and got[0] == base[:40000]


@pytest.mark.gpu
def test_throughput_on_bam_like_blocks(capsys):
    from samtools_amd import _capi
    rng = np.random.default_rng(3)
    ref = rng.choice(np.frombuffer(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88", dtype=np.uint8), 400000)
    recs, pos = [], 0
    for r in range(90000):
        pos += int(rng.integers(0, 10))
        q = np.where(rng.random(150) < 0.85, 37, rng.integers(2, 41, 150)).astype(np.uint8)
        recs.append(struct.pack("<iiIIiiii", 0, pos, 0x12c80a0c, 0x00010000, 150, -1, -1, 0) + b"read_%07d\0" % r + struct.pack("<I", 150 << 4)
                    + ref[pos // 2:pos // 2 + 75].tobytes() + q.tobytes())
    raw = b"".join(struct.pack("<I", len(x)) + x for x in recs)
    chunks = [raw[o:o + 65280] for o in range(0, len(raw), 65280)]
    streams = [_deflate(c, 1) for c in chunks]
    got, status, ms = _capi.bgzf_inflate(streams * 8, [len(c) for c in chunks] * 8)
    assert all(s == 0 for s in status) and got[:len(chunks)] == chunks and got[-len(chunks):] == chunks
    nbytes = 8 * len(raw)
    with capsys.disabled():
        print("\n[bgzf inflate on the device] %d blocks, %.1f MB -> %.1f MB in %.2f ms = %.1f GB/s of inflated bytes"
              % (len(streams) * 8, 8 * sum(map(len, streams)) / 1e6, nbytes / 1e6, ms, nbytes / ms / 1e6))
