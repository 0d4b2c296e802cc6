"""The record writer behind `calmd`, host side: a record calmd does not change leaves as the SAM line
sam_write1 would print (bam_md.c:486-489).  Two independent implementations meet here -- the engine keeps a record's aux fields as
text (host_io.h Rec::auxv, sta_io_write_sam), the oracle keeps HTSlib's binary aux block and formats it (o_calmd.c) -- on SAM and
BAM input with every aux type; and on the reference's own inputs both must hand back the file they were given."""
import os
import subprocess

import pytest

from bamio import sam_to_bam

DAT = os.path.join(os.path.dirname(__file__), "golden", "dat")

HDR = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:17\tLN:4200\n@RG\tID:g1\tSM:s\n@PG\tID:bwa\tPN:bwa\n@PG\tID:sort\tPN:samtools\tPP:bwa\n@CO\tfree text\n"
SEQ = "ACGTACGTAC"
RECS = [
    # every scalar type, in an order that is not alphabetical; values already in the form sam_format1 prints
    ("r1", 0, "17", 100, 30, "10M", "*", 0, 0, SEQ, "IIIIIIIIII",
     ["XA:A:q", "Xc:i:-5", "XC:i:200", "Xs:i:-3000", "XS:i:60000", "Xi:i:-70000", "XI:i:4000000000", "XZ:Z:some text, with:colons", "XH:H:1AE301", "RG:Z:g1"]),
    # floats go through kputd: six significant digits, half up on the truncated expansion (tests/test_kputd.py pins the rule)
    ("r2", 16, "17", 120, 0, "4M2I4M", "=", 300, 190, SEQ, "*", ["Xf:f:0.5", "Xg:f:3.14159", "Xh:f:1e+10", "Xj:f:-2.5e-07", "NM:i:2", "MD:Z:8"]),
    # arrays of every subtype, an empty array
    ("r3", 99, "17", 130, 60, "3S7M", "=", 200, 80, SEQ, "!!!!!!!!!!",
     ["Bc:B:c,-1,2,-128", "BC:B:C,0,255", "Bs:B:s,-32768,7", "BS:B:S,65535", "Bi:B:i,-2147483648,5", "BI:B:I,4294967295", "Bf:B:f,0.25,-1.5,1e+06", "Be:B:c"]),
    ("r4", 4, "*", 0, 0, "*", "*", 0, 0, "*", "*", ["XZ:Z:unmapped, no sequence"]),
    ("r5", 77, "*", 0, 0, "*", "*", 0, 0, SEQ, "ABCDEFGHIJ", []),
]


def write_case(path):
    with open(path, "w") as fh:
        fh.write(HDR)
        for q, fl, rn, pos, mq, cg, rnext, pnext, tlen, seq, qual, tags in RECS:
            fh.write("\t".join([q, str(fl), rn, str(pos), str(mq), cg, rnext, str(pnext), str(tlen), seq, qual] + tags) + "\n")


def oracle_rewrite(oracle_bin, path, ref):
    # -N: nothing is recomputed, every record only passes through the oracle's reader and writer
    p = subprocess.run([oracle_bin, "calmd", "-N", path, ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    return p.stdout


def test_every_aux_type_survives_both_writers_from_sam_and_bam(oracle_bin, tmp_path):
    from samtools_amd import _capi
    sam = str(tmp_path / "t.sam")
    write_case(sam)
    bam = sam_to_bam(sam, str(tmp_path / "t.bam"), level=1)
    ref = os.path.join(DAT, "mpileup.ref.fa")
    src = open(sam, "rb").read()
    for inp in (sam, bam):
        out = str(tmp_path / "o.sam")
        _capi.io_write_sam(inp, out)
        got = open(out, "rb").read()
        assert got == src, inp
        assert oracle_rewrite(oracle_bin, inp, ref) == src, inp


def test_text_that_is_not_in_written_form_is_normalised_the_same_way(oracle_bin, tmp_path):
    """what sam_parse1 + sam_format1 do to a field: '+5' and '007' are integers, 0.50 is a float, B values are stored at the
    subtype's width"""
    from samtools_amd import _capi
    sam = str(tmp_path / "n.sam")
    with open(sam, "w") as fh:
        fh.write("@SQ\tSN:17\tLN:4200\n")
        fh.write("\t".join(["q", "0", "17", "5", "9", "10M", "*", "0", "0", SEQ, "*", "Xa:i:+5", "Xb:i:007", "Xc:f:0.50", "Xd:f:123456.5",
                            "Xe:B:C,1,,2", "Xf:B:f,0.10,2.50", "Xg:f:1e3"]) + "\n")
    out = str(tmp_path / "o.sam")
    _capi.io_write_sam(sam, out)
    got = open(out, "rb").read()
    want = oracle_rewrite(oracle_bin, sam, os.path.join(DAT, "mpileup.ref.fa"))
    assert got == want
    rec = got.decode().splitlines()[1].split("\t")[11:]
    assert rec == ["Xa:i:5", "Xb:i:7", "Xc:f:0.5", "Xd:f:123457", "Xe:B:C,1,2", "Xf:B:f,0.1,2.5", "Xg:f:1000"]


@pytest.mark.parametrize("n", ["1", "2", "3"])
def test_reference_inputs_come_back_unchanged(oracle_bin, tmp_path, n):
    from samtools_amd import _capi
    sam = os.path.join(DAT, "mpileup.%s.sam" % n)
    out = str(tmp_path / "o.sam")
    _capi.io_write_sam(sam, out)
    assert open(out, "rb").read() == open(sam, "rb").read()
    bam = sam_to_bam(sam, str(tmp_path / "t.bam"), level=1, block=20000)
    _capi.io_write_sam(bam, out)
    assert open(out, "rb").read() == open(sam, "rb").read()


# ---- BAM output (calmd -b / -u): host_bamout.h behind sta_io_write_bam ----
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def bgzf_blocks(path):
    """(compressed size, inflated size) of every BGZF block + whether the file ends with the 28-byte EOF marker"""
    import struct
    import zlib
    raw = open(path, "rb").read()
    out, p = [], 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04" and raw[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack("<H", raw[p + 16:p + 18])[0] + 1
        data = zlib.decompress(raw[p + 18:p + bsize - 8], -15)
        crc, isz = struct.unpack("<II", raw[p + bsize - 8:p + bsize])
        assert isz == len(data) and crc == (zlib.crc32(data) & 0xffffffff)
        out.append((bsize, isz))
        p += bsize
    return out, raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


@pytest.mark.parametrize("level", [0, 6])
def test_bam_writer_reproduces_the_payload_of_a_bam_the_reference_ships(tmp_path, level):
    """test/mpileup/ce#5b.{sam,bam} are the same records as text and as a samtools-written BAM: encoding the text must give the
    BAM's inflated byte stream (header, bins, the smallest integer types, 4-bit bases), in well-formed BGZF blocks + EOF marker"""
    import gzip
    from samtools_amd import _capi
    out = str(tmp_path / "x.bam")
    _capi.io_write_bam(os.path.join(GOLD, "mpileup", "ce#5b.sam"), out, level)
    assert gzip.open(out).read() == gzip.open(os.path.join(GOLD, "mpileup", "ce#5b.bam")).read()
    blocks, has_eof = bgzf_blocks(out)
    assert has_eof and all(i <= 0xff00 for _, i in blocks)
    if level == 0:
        assert all(c > i for c, i in blocks[:-1])          # stored, not compressed (calmd -u)


def test_bam_writer_round_trips_every_aux_type_and_long_inputs(oracle_bin, tmp_path):
    from samtools_amd import _capi
    sam = str(tmp_path / "t.sam")
    write_case(sam)
    src = open(sam, "rb").read()
    bam = str(tmp_path / "t.bam"); back = str(tmp_path / "b.sam")
    _capi.io_write_bam(sam, bam, 6)
    _capi.io_write_sam(bam, back)
    assert open(back, "rb").read() == src
    assert oracle_rewrite(oracle_bin, bam, os.path.join(DAT, "mpileup.ref.fa")) == src       # the oracle's BAM reader agrees
    # many blocks: records never straddle a block boundary unless they are bigger than one
    big = os.path.join(DAT, "mpileup.1.sam")
    _capi.io_write_bam(big, bam, 0)
    _capi.io_write_sam(bam, back)
    assert open(back, "rb").read() == open(big, "rb").read()
    blocks, has_eof = bgzf_blocks(bam)
    assert has_eof and len(blocks) > 4
    import gzip
    import struct
    raw = gzip.open(bam).read()
    # walk the records and check each one lies inside one block of the inflated stream
    edges, acc = set(), 0
    for _, isz in blocks:
        acc += isz; edges.add(acc)
    l_text = struct.unpack("<i", raw[4:8])[0]; p = 8 + l_text
    n_ref = struct.unpack("<i", raw[p:p + 4])[0]; p += 4
    for _ in range(n_ref):
        l = struct.unpack("<i", raw[p:p + 4])[0]; p += 4 + l + 4
    assert p in edges                                       # the header ends its block
    while p < len(raw):
        bs = struct.unpack("<i", raw[p:p + 4])[0]
        assert not any(p < e < p + 4 + bs for e in edges)
        p += 4 + bs


def test_bam_to_bam_keeps_untouched_aux_bytes(tmp_path):
    """a BAM -> BAM run must hand untouched aux fields on byte for byte (sam_write1 on the same bam1_t does,
    bam_md.c:386-395): floats / doubles at full precision (not through kputd's six digits), B:f arrays, integers in the width
    they came in (an `i`-typed 3 stays four bytes)."""
    import gzip
    import struct
    from samtools_amd import _capi
    import bamio
    hdr = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c1\tLN:1000"]
    head = bamio.bam_header_bytes(hdr, ["c1"], [1000])
    aux = (b"XFf" + struct.pack("<f", 1.2345678806304932) + b"XDd" + struct.pack("<d", 3.141592653589793)
           + b"XIi" + struct.pack("<i", 3) + b"XSs" + struct.pack("<h", 7) + b"XBBf" + struct.pack("<I", 3) + struct.pack("<3f", 0.1, 1e-7, 123456.789)
           + b"XZZhello\0" + b"XCC" + struct.pack("<B", 200))
    qname = b"r1\0"; seq = bytes([0x12, 0x48]); qual = bytes([30, 31, 32, 33])
    core = struct.pack("<iiBBHHHIiii", 0, 99, len(qname), 60, 4681, 1, 0, 4, -1, -1, 0)
    body = core + qname + struct.pack("<I", (4 << 4) | 0) + seq + qual + aux
    rec = struct.pack("<I", len(body)) + body
    src = str(tmp_path / "in.bam"); out = str(tmp_path / "out.bam")
    open(src, "wb").write(bamio.bgzf_compress(head + rec, 1))
    _capi.io_write_bam(src, out, 6)
    raw = gzip.open(out).read()
    assert raw[len(head):].endswith(aux), (raw[len(head):], aux)
    # and the text route still prints them as sam_format1 does
    sam = str(tmp_path / "out.sam")
    _capi.io_write_sam(src, sam)
    line = [l for l in open(sam).read().split("\n") if l.startswith("r1")][0]
    assert "XF:f:1.23457" in line and "XI:i:3" in line and "XB:B:f,0.1,1e-07,123457" in line
