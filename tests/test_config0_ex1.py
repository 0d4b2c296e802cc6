"""BASELINE.json configs[0] on the CPU side: examples/ex1.sam.gz (headerless SAM, @SQ from the FASTA as `samtools view -bt
ex1.fa.fai` would add) + ex1.fa.  Here: the oracle gives the same text for the SAM and the BAM form and the product's reader
decodes both to the same records.  The GPU half (engine == oracle) is tests/test_gpu_benchsize_parity.py::test_config0_ex1."""
import subprocess

from test_gpu_benchsize_parity import _ex1


def test_ex1_oracle_sam_equals_bam_and_reader_agrees(tmp_path, oracle_bin):
    import samtools_amd as sa
    sam, bam, fa = _ex1(tmp_path)
    a = subprocess.run([oracle_bin, "mpileup", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    b = subprocess.run([oracle_bin, "mpileup", "-f", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert a == b and a.count(b"\n") > 3000
    assert a.startswith(b"seq1\t36\tG\t1\t^~.\t=\n")
    n1, h1 = sa._capi.io_scan(sam)
    n2, h2 = sa._capi.io_scan(bam)
    assert (n1, h1) == (n2, h2) and n1 == 3307
    assert sa._capi.io_scan(sam, stage=1) == sa._capi.io_scan(bam, stage=2)
