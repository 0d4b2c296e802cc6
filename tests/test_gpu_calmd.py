"""8(f) row 3 on the GPU: `samtools-amd calmd` (k_md_len / k_md_emit / k_calmd_tag + the BAQ kernels in plain and extended
mode, the aux bookkeeping of bam_md.c:156-199 and the SAM record writer) against the oracle's restatement, byte for byte: the
whole output file and the warnings on stderr."""
import os
import subprocess

import pytest

from synth import write_synth_sam
from synth_rich import write_rich_sam
from bamio import sam_to_bam

pytestmark = pytest.mark.gpu
DAT = os.path.join(os.path.dirname(__file__), "golden", "dat")
OPTS = [[], ["-e"], ["-r"], ["-r", "-E"], ["-r", "-A"], ["-r", "-A", "-E", "-e"], ["-q"], ["-n", "2"], ["-e", "-n", "3", "-q", "-r"],
        ["-d"], ["-N", "-e"], ["-Q", "-r", "-d"]]


def run_both(oracle_bin, product_bin, args, env=None, product_args=None):
    want = subprocess.run([oracle_bin, "calmd"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = subprocess.run([product_bin, "calmd", "--no-PG"] + (product_args or args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == want.returncode, got.stderr.decode()[-300:]
    assert got.stderr == want.stderr, (args, got.stderr.decode()[-300:], want.stderr.decode()[-300:])
    return got.stdout, want.stdout


@pytest.mark.parametrize("n", ["1", "2", "3"])
def test_calmd_on_the_reference_inputs(oracle_bin, product_bin, n):
    sam, fa = os.path.join(DAT, "mpileup.%s.sam" % n), os.path.join(DAT, "mpileup.ref.fa")
    for opts in OPTS:
        got, want = run_both(oracle_bin, product_bin, opts + [sam, fa])
        assert got == want, opts
    # the stored aligner tags are reproduced by the device path: the input comes back byte for byte
    got, _ = run_both(oracle_bin, product_bin, [sam, fa])
    assert got == open(sam, "rb").read()


@pytest.mark.parametrize("batch", [None, "97"])
def test_calmd_equals_oracle_on_synthetic_and_messy_input(tmp_path, oracle_bin, product_bin, batch):
    env = dict(os.environ)
    if batch:
        env["STA_CALMD_BATCH"] = batch
    sam, fa = write_synth_sam(str(tmp_path), n_ref=30000, depth=20, read_len=150, seed=41, paired=True, indel_rate=0.08, max_indel=12)
    rich, rfa = write_rich_sam(str(tmp_path), seed=6, n_templates=1200)
    for src, ref in ((sam, fa), (rich, rfa)):
        bam = sam_to_bam(src, src[:-4] + ".bam", level=1, block=30000)
        for opts in OPTS:
            got, want = run_both(oracle_bin, product_bin, opts + [src, ref], env, opts + [bam, ref])
            assert got == want, (os.path.basename(src), opts)


def test_calmd_cap_mapq(tmp_path, oracle_bin, product_bin):
    """-C (bam_md.c:480-483: sam_cap_mapq lowers the MAPQ field; a read beyond the coefficient gets -1 = 255 in the unsigned field): alone,
    behind -r with and without -A (the cap sees the qualities the record carries at that point), with -q / -n (which rewrite qualities
    only afterwards).  Mismatch-rich reads and the hand-derived vectors of tests/test_cap_mapq_vectors.py (clip term, square root, drop)."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=12000, depth=20, read_len=100, seed=43, paired=True, sub_rate=0.04, indel_rate=0.05)
    changed = 0
    for opts in (["-C", "50"], ["-C", "40", "-r"], ["-C", "50", "-r", "-A", "-E"], ["-C", "50", "-e", "-q", "-n", "4"], ["-C", "10"]):
        got, want = run_both(oracle_bin, product_bin, opts + [sam, fa])
        assert got == want, opts
        base, _ = run_both(oracle_bin, product_bin, [o for o in opts if o not in ("-C", "50", "40", "10")] + [sam, fa])
        changed += got != base
    assert changed >= 4                      # (-C 10 is "off": capQ > 10)
    import test_cap_mapq_vectors as V
    vfa, vec = V._write(str(tmp_path))
    for name, vsam, want_mq in vec:
        p = subprocess.run([product_bin, "calmd", "-C", "50", "--no-PG", vsam, vfa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-300:]
        rec = [l for l in p.stdout.decode().splitlines() if not l.startswith("@")][0].split("\t")
        assert int(rec[4]) == (255 if want_mq is None else want_mq), name


def test_records_that_already_carry_baq_tags(tmp_path, oracle_bin, product_bin):
    """sam_prob_realn's tag branches (HTSlib realn.c): BQ:Z with -A is applied and renamed ZQ:Z, ZQ:Z without -A is taken back
    out of the qualities and renamed BQ:Z, the matching cases are left alone, a record with both loses its ZQ:Z; wrong stored
    MD / NM values are replaced on the way."""
    sam, fa = os.path.join(DAT, "mpileup.1.sam"), os.path.join(DAT, "mpileup.ref.fa")
    made = {}
    for name, opts in (("bq", ["-r"]), ("zq", ["-r", "-A"]), ("bq_ext", ["-r", "-E"])):
        out = subprocess.run([oracle_bin, "calmd"] + opts + [sam, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        made[name] = str(tmp_path / (name + ".sam"))
        open(made[name], "wb").write(out)
        assert (b"\tBQ:Z:" if name != "zq" else b"\tZQ:Z:") in out
    # both tags on every third tagged record, a wrong NM on every fifth, MD stripped on every seventh
    lines = open(made["bq"]).read().splitlines()
    k = 0
    both = []
    for l in lines:
        if not l.startswith("@") and "\tBQ:Z:" in l:
            k += 1
            f = l.split("\t")
            if k % 3 == 0:
                f.insert(11, "ZQ:Z:" + "@" * len(f[9]))
            if k % 5 == 0:
                f = [("NM:i:41" if t.startswith("NM:i:") else t) for t in f]
            if k % 7 == 0:
                f = [t for t in f if not t.startswith("MD:Z:")]
            l = "\t".join(f)
        both.append(l)
    made["both"] = str(tmp_path / "both.sam")
    open(made["both"], "w").write("\n".join(both) + "\n")
    for name, path in made.items():
        bam = sam_to_bam(path, path[:-4] + ".bam", level=1)
        for opts in (["-r"], ["-r", "-A"], ["-r", "-E"], ["-r", "-A", "-e", "-q"], []):
            got, want = run_both(oracle_bin, product_bin, opts + [path, fa], None, opts + [bam, fa])
            assert got == want, (name, opts)
    # and the round trip the two conversions promise: BQ -> (-A) ZQ -> (no -A) BQ gives the BQ file back
    z = subprocess.run([product_bin, "calmd", "--no-PG", "-r", "-A", made["bq"], fa], stdout=subprocess.PIPE, check=True).stdout
    zp = str(tmp_path / "z.sam"); open(zp, "wb").write(z)
    back = subprocess.run([product_bin, "calmd", "--no-PG", "-r", zp, fa], stdout=subprocess.PIPE, check=True).stdout
    assert back == open(made["bq"], "rb").read()


def test_pg_line(tmp_path, product_bin):
    """without --no-PG every end of a @PG chain in the header gets a samtools line chained to it (sam_hdr_add_pg, bam_md.c:425-431),
    with IDs made unique (samtools, samtools.1, ...)"""
    sam, fa = os.path.join(DAT, "mpileup.1.sam"), os.path.join(DAT, "mpileup.ref.fa")
    out = subprocess.run([product_bin, "calmd", "-e", sam, fa], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    src_hdr = [l for l in open(sam).read().splitlines() if l.startswith("@")]
    prev = [dict(t.split(":", 1) for t in l.split("\t")[1:]) for l in src_hdr if l.startswith("@PG")]
    ends = [p["ID"] for p in prev if p["ID"] not in {q.get("PP") for q in prev}]
    assert out[:len(src_hdr)] == src_hdr
    new = out[len(src_hdr):len(src_hdr) + max(1, len(ends))]
    ids = set()
    for l, pp in zip(new, ends):
        f = l.split("\t")
        assert f[0] == "@PG" and f[1].startswith("ID:samtools") and f[2] == "PN:samtools" and f[3] == "PP:" + pp
        assert f[-1].startswith("CL:samtools-amd calmd -e ") and f[-2].startswith("VN:")
        ids.add(f[1])
    assert len(ids) == len(new) == len(ends)
    plain = subprocess.run([product_bin, "calmd", "--no-PG", "-e", sam, fa], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert plain == out[:len(src_hdr)] + out[len(src_hdr) + len(new):]
    # a header without any @PG line: one line without PP
    bare = tmp_path / "bare.sam"
    bare.write_text("\n".join(l for l in open(sam).read().splitlines() if not l.startswith("@PG")) + "\n")
    out = subprocess.run([product_bin, "calmd", str(bare), fa], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    pg = [l for l in out if l.startswith("@PG")]
    assert len(pg) == 1 and pg[0].startswith("@PG\tID:samtools\tPN:samtools\tVN:")


def test_bam_output(tmp_path, oracle_bin, product_bin):
    """calmd -b / -u (bam_md.c:386-395): the reference's own test runs `calmd -uAr` and checks the BGZF magic (test/test.pl:3652-3661);
    here the BAM is also decoded again and must hold the records of the SAM output, header included"""
    from samtools_amd import _capi
    sam, fa = os.path.join(DAT, "mpileup.1.sam"), os.path.join(DAT, "mpileup.ref.fa")
    for flag in ("-u", "-b"):
        out = subprocess.run([product_bin, "calmd", "--no-PG", flag + "Ar", sam, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert out.returncode == 0, out.stderr.decode()[-300:]
        assert out.stdout[:2] == b"\x1f\x8b"
        bam = str(tmp_path / "o.bam"); open(bam, "wb").write(out.stdout)
        back = str(tmp_path / "o.sam")
        _capi.io_write_sam(bam, back)
        want = subprocess.run([oracle_bin, "calmd", "-Ar", sam, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        assert open(back, "rb").read() == want, flag
