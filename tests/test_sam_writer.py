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
