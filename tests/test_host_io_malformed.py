"""The BAM decoder does not trust on-disk lengths and ids: malformed records are errors, not out-of-bounds
reads, and a CIGAR parked in the CG:B,I tag (more than 65535 operations, SAM spec 4.2.2) is expanded.  No device needed."""
import gzip
import struct

import pytest

from bamio import _bgzf_block, _EOF, sam_to_bam

SAM = ("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:1000\n@SQ\tSN:c2\tLN:500\n"
       "r1\t0\tc1\t11\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tNM:i:0\tRG:Z:g1\n"
       "r2\t16\tc1\t21\t60\t4M2D6M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tXS:i:-70000\tBQ:Z:@@@@@@@@@@\n"
       "r3\t0\tc2\t5\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tXB:B:s,1,2,3\n")


def _raw(tmp_path):
    sam = tmp_path / "a.sam"
    sam.write_text(SAM)
    bam = tmp_path / "a.bam"
    sam_to_bam(str(sam), str(bam))
    return bytearray(gzip.open(str(bam)).read())


def _write(tmp_path, raw, name):
    p = tmp_path / name
    with open(p, "wb") as fo:
        fo.write(_bgzf_block(bytes(raw), 1))
        fo.write(_EOF)
    return str(p)


def _first_record(raw):
    l_text = struct.unpack("<i", raw[4:8])[0]
    o = 8 + l_text
    n_ref = struct.unpack("<i", raw[o:o + 4])[0]
    o += 4
    for _ in range(n_ref):
        o += 4 + struct.unpack("<i", raw[o:o + 4])[0] + 4
    return o, 8 + l_text


def _scan(path, stage=0):
    import samtools_amd as sa
    return sa._capi.io_scan(path, threads=2, stage=stage)


def test_intact_file_decodes(tmp_path):
    raw = _raw(tmp_path)
    assert _scan(_write(tmp_path, raw, "ok.bam"))[0] == 3
    assert _scan(_write(tmp_path, raw, "ok.bam"), stage=2)[0] == 3


@pytest.mark.parametrize("what", ["refid", "mate_refid", "neg_refid", "trunc_aux_int", "trunc_aux_z", "trunc_aux_b", "huge_b", "bad_type"])
def test_malformed_records_are_errors(tmp_path, what):
    raw = _raw(tmp_path)
    o, _ = _first_record(raw)
    bs = struct.unpack("<i", raw[o:o + 4])[0]
    if what == "refid":
        raw[o + 4:o + 8] = struct.pack("<i", 2)                 # n_ref is 2
    elif what == "mate_refid":
        raw[o + 4 + 20:o + 4 + 24] = struct.pack("<i", 7)
    elif what == "neg_refid":
        raw[o + 4:o + 8] = struct.pack("<i", -5)
    elif what == "trunc_aux_int":
        # drop the last 2 bytes of the first record (inside RG:Z) and then cut NM:i's value short by rewriting it as a 4-byte type
        k = raw.index(b"NMC", o)
        raw[k + 2] = ord("i")                                    # value now claims 4 bytes; shift the record end onto it
        raw[o:o + 4] = struct.pack("<i", k + 4 - (o + 4))
        del raw[k + 4:o + 4 + bs]
    elif what == "trunc_aux_z":
        k = raw.index(b"RGZg1\0", o)
        del raw[k + 4:o + 4 + bs]                                # no NUL before the record ends
        raw[o:o + 4] = struct.pack("<i", k + 4 - (o + 4))
    elif what in ("trunc_aux_b", "huge_b"):
        k = raw.index(b"XBBs", o)
        raw[k + 4:k + 8] = struct.pack("<I", 0x7fffffff if what == "huge_b" else 40)
    elif what == "bad_type":
        k = raw.index(b"NMC", o)
        raw[k + 2] = ord("?")
    path = _write(tmp_path, raw, what + ".bam")
    for stage in (0, 1, 2):
        with pytest.raises(RuntimeError):
            _scan(path, stage)


@pytest.mark.parametrize("what", ["neg_nref", "neg_lname", "huge_lname"])
def test_malformed_header_fails_to_open(tmp_path, what):
    raw = _raw(tmp_path)
    _, nref_at = _first_record(raw)
    if what == "neg_nref":
        raw[nref_at:nref_at + 4] = struct.pack("<i", -1)
    else:
        raw[nref_at + 4:nref_at + 8] = struct.pack("<i", -3 if what == "neg_lname" else 0x7ffffff0)
    with pytest.raises(RuntimeError):
        _scan(_write(tmp_path, raw, what + ".bam"))


def test_cg_tag_cigar_is_expanded(tmp_path):
    """placeholder <l_seq>S<span>N + CG:B,I == the real CIGAR (sam.c bam_tag2cigar)."""
    real = tmp_path / "real.sam"
    real.write_text("@SQ\tSN:c1\tLN:1000\nr2\t16\tc1\t21\t60\t4M2D6M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    parked = tmp_path / "parked.sam"
    parked.write_text("@SQ\tSN:c1\tLN:1000\nr2\t16\tc1\t21\t60\t10S12N\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tCG:B:I,%d,%d,%d\n"
                      % (4 << 4 | 0, 2 << 4 | 2, 6 << 4 | 0))
    a = sam_to_bam(str(real), str(tmp_path / "real.bam"))
    b = sam_to_bam(str(parked), str(tmp_path / "parked.bam"))
    for stage in (0, 2):
        assert _scan(a, stage) == _scan(b, stage)
    # bam_tag2cigar looks only at the first placeholder op and accepts subtype 'i' as well: `10S` alone + CG:B:i
    short = tmp_path / "short.sam"
    short.write_text("@SQ\tSN:c1\tLN:1000\nr2\t16\tc1\t21\t60\t10S\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tCG:B:i,%d,%d,%d\n"
                     % (4 << 4 | 0, 2 << 4 | 2, 6 << 4 | 0))
    d = sam_to_bam(str(short), str(tmp_path / "short.bam"))
    for stage in (0, 2):
        assert _scan(d, stage) == _scan(a, stage)
    # a CG shorter than the placeholder is left alone (CG_len < n_cigar)
    tiny = tmp_path / "tiny.sam"
    tiny.write_text("@SQ\tSN:c1\tLN:1000\nr2\t16\tc1\t21\t60\t10S12N\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\tCG:B:I,%d\n" % (10 << 4 | 0))
    t = sam_to_bam(str(tiny), str(tmp_path / "tiny.bam"))
    assert _scan(t) != _scan(a)
    # without the tag the placeholder is an ordinary CIGAR and decodes differently
    plain = tmp_path / "plain.sam"
    plain.write_text("@SQ\tSN:c1\tLN:1000\nr2\t16\tc1\t21\t60\t10S12N\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    c = sam_to_bam(str(plain), str(tmp_path / "plain.bam"))
    assert _scan(c) != _scan(a)


def test_second_input_with_extra_contigs_is_an_error_not_a_crash(tmp_path):
    """records of a later input may not name contigs the first input's header lacks (the drivers print names / lengths from it)"""
    one = tmp_path / "one.sam"
    one.write_text("@SQ\tSN:c1\tLN:1000\nr1\t0\tc1\t11\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n")
    two = tmp_path / "two.sam"
    two.write_text(SAM)
    with pytest.raises(RuntimeError):
        _scan([str(one), str(two)], stage=2)
    with pytest.raises(RuntimeError):
        _scan([str(one), str(two)], stage=1)
