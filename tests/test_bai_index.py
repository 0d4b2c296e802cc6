"""Index-driven region reads (the host side of `-r` / sharded runs; sam_itr_querys at bam_plcmd.c:550): with a .bai beside the BAM
the reader starts at the linear index's virtual offset and stops behind the region; it must deliver exactly the records a scan
of the whole file with the region filter delivers.  No device needed.  The BAI writer is test infrastructure (tests/bamio.py)."""
import os

import numpy as np
import pytest

from bamio import sam_to_bam, write_bai
from synth import synth_ref, synth_reads, write_sam


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("bai")
    # three contigs, the second one empty; small BGZF blocks so that records straddle blocks and offsets inside blocks matter
    sam = str(d / "x.sam")
    parts = []
    for name, n, seed in (("c1", 300000, 1), ("c2", 50000, 2), ("c3", 200000, 3)):
        ref = synth_ref(n, seed=seed)
        rd = synth_reads(ref, depth=0 if name == "c2" else 4, read_len=150, seed=seed + 10, indel_rate=0.02) if name != "c2" else None
        parts.append((name, n, rd))
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for name, n, _ in parts:
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (name, n))
        for name, n, rd in parts:
            if rd is None:
                continue
            tmp = str(d / "t.sam")
            write_sam(tmp, rd, name, n)
            for line in open(tmp):
                if not line.startswith("@"):
                    fh.write(line)
    b = sam_to_bam(sam, str(d / "x.bam"), block=7000)
    write_bai(b)
    return b


REGIONS = ["c1", "c1:1-1000", "c1:150000-150100", "c1:299000-300000", "c2", "c3", "c3:1-50", "c3:100000-199999", "c1:16384-16385", "c3:199990"]


@pytest.mark.parametrize("reg", REGIONS)
def test_indexed_region_read_equals_filtered_full_scan(bam, reg):
    from samtools_amd import _capi
    n_ix, h_ix, used = _capi.io_scan_region(bam, reg, threads=2, use_index=True)
    n_fs, h_fs, used_fs = _capi.io_scan_region(bam, reg, threads=2, use_index=False)
    assert not used_fs
    assert (n_ix, h_ix) == (n_fs, h_fs)
    if reg != "c1" and not reg.startswith("c1:1-") and reg != "c2":
        assert used            # (a region at the very start of the file has offset = the first record: seeking is still "used")
    if reg == "c2":
        assert n_ix == 0


def test_region_counts_match_the_generator(bam):
    from samtools_amd import _capi
    n_all, _, _ = _capi.io_scan_region(bam, "c3", threads=1)
    n_part, _, _ = _capi.io_scan_region(bam, "c3:100001-100150", threads=1)
    assert n_all > 4000 and 0 < n_part < 40


def test_unsorted_input_is_filtered_to_its_last_record(tmp_path):
    """The early stop behind a region is for coordinate-sorted input only (@HD SO:coordinate, or an index seek): a file that does not
    say so is filtered to its last record, so records of the region that come after a later one are kept."""
    from samtools_amd import _capi
    ref = synth_ref(60000, seed=5)
    rd = synth_reads(ref, depth=3, read_len=100, seed=6)
    tmp = str(tmp_path / "t.sam"); write_sam(tmp, rd, "c1", len(ref))
    head = [l for l in open(tmp) if l.startswith("@") and not l.startswith("@HD")]
    recs = [l for l in open(tmp) if not l.startswith("@")]
    recs_rev = recs[::-1]                                         # descending positions: every record of the region follows a "later" one
    srt, uns = str(tmp_path / "s.sam"), str(tmp_path / "u.sam")
    open(srt, "w").write("@HD\tVN:1.6\tSO:coordinate\n" + "".join(head) + "".join(recs))
    open(uns, "w").write("@HD\tVN:1.6\tSO:unsorted\n" + "".join(head) + "".join(recs_rev))
    n_s, _, _ = _capi.io_scan_region(srt, "c1:1000-2000", threads=1, use_index=False)
    n_u, _, _ = _capi.io_scan_region(uns, "c1:1000-2000", threads=1, use_index=False)
    assert n_s > 10 and n_u == n_s


def test_an_index_older_than_its_data_file_is_used_with_a_warning(bam, tmp_path, capfd):
    """HTSlib warns about an index older than its data file and uses it all the same (copied or checked-out data often carries time
    stamps the wrong way round): so do the drivers (round 5; round 4 fell back to a whole-file read)."""
    import shutil
    from samtools_amd import _capi
    b2 = str(tmp_path / "y.bam")
    shutil.copy(bam, b2); shutil.copy(bam + ".bai", b2 + ".bai")
    os.utime(b2 + ".bai", (1, 1))                                 # the index is "from 1970", the data file is new
    n_ix, h_ix, used = _capi.io_scan_region(b2, "c3:100000-199999", threads=2, use_index=True)
    n_fs, h_fs, _ = _capi.io_scan_region(bam, "c3:100000-199999", threads=2, use_index=False)
    assert (n_ix, h_ix) == (n_fs, h_fs) and used
    assert "older than the data file" in capfd.readouterr().err
