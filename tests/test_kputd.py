"""Float aux values in `mpileup --output-extra TAG` go through HTSlib's kputd (bam_plcmd.c:838-840), not printf("%g").
Hand-derived vectors from the published algorithm (value x 10^10 truncated, half a unit of the sixth significant digit added,
six digits kept, trailing zeros culled; "%g" outside [0.0001, 999999]) against the two independent restatements: the engine's
formatter (through the C-ABI, no device needed) and the oracle's (through its CLI)."""
import ctypes
import os
import struct
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))

# (float32 value as SAM text, expected text).  The third group is where kputd and "%g" differ: exact binary ties.
VECTORS = [
    ("0", "0"), ("-0.0", "-0"), ("1", "1"), ("-1.5", "-1.5"), ("0.1", "0.1"), ("0.5", "0.5"), ("2.5", "2.5"), ("100", "100"),
    ("1234.5", "1234.5"), ("3.14159274", "3.14159"), ("0.0001", "0.0001"), ("0.000123456789", "0.000123457"), ("999999", "999999"),
    ("1.9999995", "2"), ("-123.456", "-123.456"),
    ("999999.5", "1e+06"), ("0.00009", "9e-05"), ("1e10", "1e+10"), ("-2.5e-7", "-2.5e-07"),
    ("123456.5", "123457"), ("100000.5", "100001"), ("12345.25", "12345.3"), ("-100000.5", "-100001"),
]


def test_engine_formatter_vectors():
    import samtools_amd as sa
    from samtools_amd import _capi
    buf = ctypes.create_string_buffer(64)
    for text, want in VECTORS:
        v = struct.unpack("<f", struct.pack("<f", float(text)))[0]
        n = _capi.lib.sta_format_aux_float(v, buf, 64)
        assert n == len(want) and buf.value.decode() == want, (text, buf.value, want)
    # doubles ('d' values) take the same path
    for v, want in ((123456.5, "123457"), (0.1, "0.1"), (1e-5, "1e-05"), (0.30000000000000004, "0.3"), (999999.0000001, "999999")):
        _capi.lib.sta_format_aux_float(v, buf, 64)
        assert buf.value.decode() == want, (v, buf.value, want)


def test_oracle_prints_the_same_vectors(tmp_path, oracle_bin):
    sam = tmp_path / "f.sam"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:100000\n")
        for k, (text, _) in enumerate(VECTORS):
            fh.write("r%d\t0\tc\t%d\t60\t4M\t*\t0\t0\tACGT\tIIII\tXF:f:%s\n" % (k, 1 + 10 * k, text))
    out = subprocess.run([oracle_bin, "mpileup", "--output-extra", "XF", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    got = {}
    for line in out.splitlines():
        f = line.split("\t")
        got[(int(f[1]) - 1) // 10] = f[6]
    for k, (text, want) in enumerate(VECTORS):
        assert got[k] == want, (text, got[k], want)


import pytest


@pytest.mark.gpu
def test_engine_cli_prints_float_tags_like_the_oracle(tmp_path, oracle_bin, product_bin):
    """the same vectors end to end: `mpileup --output-extra XF,XD` on SAM and BAM input (f values; BAM also carries them as 'f')"""
    from bamio import sam_to_bam
    sam = tmp_path / "f.sam"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:100000\n")
        for k, (text, _) in enumerate(VECTORS):
            fh.write("r%d\t0\tc\t%d\t60\t4M\t*\t0\t0\tACGT\tIIII\tXF:f:%s\n" % (k, 1 + 10 * k, text))
    bam = sam_to_bam(str(sam), str(tmp_path / "f.bam"))
    want = subprocess.run([oracle_bin, "mpileup", "--output-extra", "XF", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    for inp in (str(sam), bam):
        got = subprocess.run([product_bin, "mpileup", "--output-extra", "XF", inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, got.stderr.decode()[-300:]
        assert got.stdout == want
