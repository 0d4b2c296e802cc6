"""The drivers' reference loader (host_io.h Fasta; stands where fai_load / faidx_fetch_seq64 stand at bam_plcmd.c:289-352).
With a .fai beside a plain FASTA only the index is read up front and a contig's bases are read when they are first asked for; the
result must be what parsing the whole file gives: same names, same bases, whatever the line width, the last line, the case, the order
the contigs are asked for in -- and a stale index must fall back to parsing, not hand out shifted bases."""
import gzip
import os
import random

import pytest

from samtools_amd import _capi


def write_fasta(path, contigs, widths, final_newline=True, crlf=False):
    """contigs: [(name with optional description, sequence)]; widths: line width per contig.  Returns the .fai text."""
    fai = []
    with open(path, "wb") as fh:
        off = 0
        for (name, seq), w in zip(contigs, widths):
            head = (">" + name + "\n").encode()
            fh.write(head); off += len(head)
            eol = b"\r\n" if crlf else b"\n"
            lines = [seq[i:i + w] for i in range(0, len(seq), w)] or []
            fai.append("%s\t%d\t%d\t%d\t%d" % (name.split()[0], len(seq), off, w, w + len(eol)))
            body = eol.join(l.encode() for l in lines)
            if lines:
                body += eol
            fh.write(body); off += len(body)
        if not final_newline:
            fh.seek(-1, os.SEEK_END); fh.truncate()
    return "\n".join(fai) + "\n"


def rand_seq(rnd, n):
    return "".join(rnd.choice("ACGTacgtNnRY") for _ in range(n))


def both(path):
    lazy = _capi.io_fasta_scan(path)
    back = _capi.io_fasta_scan(path, order=1)
    os.environ["STA_FASTA_WHOLE"] = "1"
    try:
        whole = _capi.io_fasta_scan(path)
    finally:
        del os.environ["STA_FASTA_WHOLE"]
    return lazy, back, whole


def test_index_driven_loading_equals_parsing(tmp_path):
    rnd = random.Random(5)
    for case in range(12):
        n = rnd.randint(1, 9)
        contigs = [("c%d some description" % i if rnd.random() < 0.3 else "c%d" % i, rand_seq(rnd, rnd.choice([0, 1, 59, 60, 61, 600, 4096, 70001]))) for i in range(n)]
        widths = [rnd.choice([1, 7, 60, 70, 80, 100000]) for _ in contigs]
        path = str(tmp_path / ("r%d.fa" % case))
        fai = write_fasta(path, contigs, widths, final_newline=case % 3 != 0 or not contigs[-1][1], crlf=case % 4 == 1)
        whole_only = _capi.io_fasta_scan(path)
        assert not whole_only[3] and whole_only[0] == n and whole_only[1] == sum(len(s) for _, s in contigs)
        open(path + ".fai", "w").write(fai)
        lazy, back, whole = both(path)
        assert lazy[3] and back[3] and not whole[3]
        assert lazy[:3] == back[:3] == whole[:3] == whole_only[:3], case


def test_reference_fasta_of_the_test_suite(tmp_path):
    """the FASTA the reference's mpileup tests use, indexed here the way `samtools faidx` would"""
    import shutil
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dat", "mpileup.ref.fa")
    path = str(tmp_path / "ref.fa")
    shutil.copy(src, path)
    raw = open(path, "rb").read()
    fai, off, name, seqlen, first, w, lw = [], 0, None, 0, 0, 0, 0
    for line in raw.split(b"\n")[:-1]:
        if line.startswith(b">"):
            if name is not None:
                fai.append("%s\t%d\t%d\t%d\t%d" % (name, seqlen, first, w, lw))
            name, seqlen, first, w, lw = line[1:].split()[0].decode(), 0, off + len(line) + 1, 0, 0
        else:
            if not w:
                w, lw = len(line), len(line) + 1
            seqlen += len(line)
        off += len(line) + 1
    fai.append("%s\t%d\t%d\t%d\t%d" % (name, seqlen, first, w, lw))
    plain = _capi.io_fasta_scan(path)
    open(path + ".fai", "w").write("\n".join(fai) + "\n")
    lazy = _capi.io_fasta_scan(path)
    assert lazy[3] and not plain[3] and lazy[:3] == plain[:3]


def test_stale_or_foreign_index_falls_back_to_parsing(tmp_path):
    rnd = random.Random(9)
    contigs = [("a", rand_seq(rnd, 500)), ("b", rand_seq(rnd, 700)), ("c", rand_seq(rnd, 90))]
    path = str(tmp_path / "s.fa")
    fai = write_fasta(path, contigs, [60, 60, 60])
    want = _capi.io_fasta_scan(path)
    # the file was edited after indexing: a base inserted into contig a shifts everything behind it
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:10] + b"G" + raw[10:])
    edited = _capi.io_fasta_scan(path)
    open(path + ".fai", "w").write(fai)
    got = _capi.io_fasta_scan(path)
    assert got[:3] == edited[:3] and got[:3] != want[:3]
    # an index of another file altogether
    open(path + ".fai", "w").write("zzz\t1000000\t5\t60\t61\n")
    foreign = _capi.io_fasta_scan(path)
    assert not foreign[3] and foreign[:3] == edited[:3]
    # a compressed FASTA is always parsed
    gz = str(tmp_path / "s.fa.gz")
    with open(path, "rb") as fi, gzip.open(gz, "wb") as fo:
        fo.write(fi.read())
    open(gz + ".fai", "w").write(fai)
    g = _capi.io_fasta_scan(gz)
    assert not g[3] and g[:3] == edited[:3]


def test_concurrent_fetches_under_thread_sanitizer(tmp_path):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "samtools_amd", "csrc")
    exe = str(tmp_path / "fasta_threads")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                    os.path.join(here, "cpu", "fasta_threads.cpp"), os.path.join(here, "cpu", "pinned_stub.cpp"), os.path.join(csrc, "host_io.cpp"), os.path.join(csrc, "host_bgzf.cpp"),
                    os.path.join(csrc, "host_inflate.cpp"), "-lz", "-pthread"], check=True)
    rnd = random.Random(3)
    contigs = [("s%d" % i, rand_seq(rnd, rnd.randint(100, 120000))) for i in range(12)]
    path = str(tmp_path / "t.fa")
    open(path + ".fai", "w").write(write_fasta(path, contigs, [70] * 12))
    p = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout.strip() == b"ok", (p.stdout, p.stderr[-600:])
