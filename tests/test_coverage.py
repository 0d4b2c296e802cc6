"""`coverage` (tabular mode) against the reference's goldens test/coverage/{1..5}.expected
(test/test.pl:4143-4161; fixtures under tests/golden/coverage, input = test/dat/sample.sam).
CPU: oracle/o_coverage.c on the restated HTSlib iterator.  GPU: coverage.c's loop on the engine's bam_mplp_* surface."""
import os
import subprocess

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coverage")


def cases(tmp_path):
    s = os.path.join(G, "sample.sam")
    s1 = str(tmp_path / "sample1.sam")          # test.pl: sed '/A1/d'
    with open(s) as fi, open(s1, "w") as fo:
        fo.writelines(l for l in fi if "A1" not in l)
    return [("1.expected", [s]), ("1.expected", ["--min-depth", "1", s]), ("2.expected", ["--min-depth", "2", s]),
            ("3.expected", ["--min-depth", "2", "-Q", "8", "-q", "45", s]), ("4.expected", ["--min-depth", "1", s, s1]),
            ("5.expected", ["--min-depth", "4", s, s1])]


def run(exe, args, iterator=False):
    env = dict(os.environ, STA_COV_ITERATOR="1") if iterator else None      # reference loop on the bam_mplp_* surface instead of k_cov_cols
    p = subprocess.run([exe, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    return p.stdout


def test_oracle_coverage_matches_reference_goldens(oracle_bin, tmp_path):
    for exp, args in cases(tmp_path):
        assert run(oracle_bin, args) == open(os.path.join(G, exp), "rb").read(), exp


@pytest.mark.gpu
@pytest.mark.parametrize("iterator", [False, True], ids=["device_reduction", "iterator_loop"])
def test_engine_coverage_matches_reference_goldens(product_bin, oracle_bin, tmp_path, iterator):
    for exp, args in cases(tmp_path):
        assert run(product_bin, args, iterator) == open(os.path.join(G, exp), "rb").read(), exp
    # options without a golden: engine vs oracle (region, flags, read length, header off) on a bigger file
    big = os.path.join(os.path.dirname(G), "dat", "mpileup.1.sam")
    three = [os.path.join(os.path.dirname(G), "dat", "mpileup.%d.sam" % k) for k in (1, 2, 3)]
    for args in (["-r", "17:200-900", big], ["-H", "--ff", "UNMAP,DUP", "-l", "50", "-Q", "20", big], ["-q", "30", "-d", "10", big],
                 ["--min-depth", "40", "-Q", "30"] + three, ["-r", "17:4100-4200", big]):
        assert run(product_bin, args, iterator) == run(oracle_bin, args), args


# ---- histogram (-m / -A / -w) and depth plot (-D): coverage.c:223-304.  The reference holds no expected output for these modes
# (the oracle's plot is unpinned); what can be checked independently is that the numbers the plot prints agree with `depth`.
def hist_args(big, three):
    return (["-m", big], ["-A", "-w", "32", big], ["-D", "-w", "37", big], ["-m", "-r", "17:100-2000", big], ["-D", "-A", "-Q", "20", "--min-depth", "3"] + three,
            ["-w", "50", "-r", "17:4100-4200", big], ["-w", "2000", "-r", "17:150-200", big], ["-m", "-o", "-", big])


def test_oracle_plot_numbers_agree_with_depth(oracle_bin):
    big = os.path.join(os.path.dirname(G), "dat", "mpileup.1.sam")
    out = run(oracle_bin, ["-D", "-w", "37", big]).decode()
    # `depth -a -J`-free recount: coverage drops deletions and default-filtered reads; depth's default filter is the same flag set
    p = subprocess.run([oracle_bin, "depth", "-a", big], stdout=subprocess.PIPE, check=True)
    dep = [int(l.split(b"\t")[2]) for l in p.stdout.splitlines()]
    n = len(dep)
    width = n // 37
    bins = [0] * 37
    for i, d in enumerate(dep):
        if i // width < 37:
            bins[i // width] += d
    assert ("Histo max cov:   %.5g" % (max(bins) / width)) in out
    assert ("Histo bin width: %dbp" % width) in out
    rows = [l for l in out.splitlines() if l.startswith(">")]
    assert len(rows) == 10 and all(l.count("│") == 2 for l in rows)


@pytest.mark.gpu
def test_engine_histogram_and_depth_plot_match_oracle(product_bin, oracle_bin, tmp_path):
    big = os.path.join(os.path.dirname(G), "dat", "mpileup.1.sam")
    three = [os.path.join(os.path.dirname(G), "dat", "mpileup.%d.sam" % k) for k in (1, 2, 3)]
    env = dict(os.environ, COLUMNS="100")
    for args in hist_args(big, three):
        a = subprocess.run([product_bin, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        b = subprocess.run([oracle_bin, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert a.returncode == 0 and b.returncode == 0, (args, a.stderr[-300:], b.stderr[-300:])
        assert a.stdout == b.stdout, args
    # several contigs (an empty line between the plots), windows much narrower than a bin and much wider
    from synth_rich import write_rich_sam
    sam, _fa = write_rich_sam(str(tmp_path), seed=5, n_templates=3000)
    for wcols in ("700", "4194304"):
        e2 = dict(env, STA_WINDOW_COLS=wcols)
        for args in (["-m", sam], ["-D", "-w", "64", sam], ["-A", "-w", "17", "-q", "20", sam]):
            a = subprocess.run([product_bin, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e2)
            b = subprocess.run([oracle_bin, "coverage"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert a.returncode == 0 and b.returncode == 0, (args, a.stderr[-300:], b.stderr[-300:])
            assert a.stdout == b.stdout, (wcols, args)


def test_plot_title_axis_and_labels_as_in_the_reference_manual(oracle_bin, tmp_path):
    """The only histogram output the reference tree holds is the worked example of its manual (doc/samtools-coverage.1:147-178:
    `coverage -A -w 32 -r chr1:1M-12M` and `--plot-depth -w 32 -A -r chr1:24500000-25600000` on a hg19-sized chr1).  The bars depend
    on data nobody has, but the title line, the x axis (label positions, centring, the K / M rounding of readable_bps: the middle labels
    use the 0-based region start, the first one start + 1) and the bin width only depend on the region: they must come out as printed."""
    sam = tmp_path / "man.sam"
    read = "%s\t0\tchr1\t%d\t60\t50M\t*\t0\t0\t" + "A" * 50 + "\t" + "I" * 50
    sam.write_text("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:249250621\n" + read % ("r1", 5000000) + "\n" + read % ("r2", 25000000) + "\n")
    a = run(oracle_bin, ["-A", "-w", "32", "-r", "chr1:1000000-12000000", str(sam)]).decode().split("\n")
    assert a[0] == "chr1 (249.25Mbp)"
    assert a[11] == "        1.00M     4.44M     7.87M       12.00M "
    assert a[1].startswith(">") and a[1][8:11] == "% |" and a[1][43] == "|" and a[9].endswith("| Histo bin width: 343.8Kbp")
    assert a[10][45:].startswith("Histo max bin:   ")
    b = run(oracle_bin, ["-m", "-r", "chr1:24500000-25600000", "--plot-depth", "-w", "32", "-A", str(sam)]).decode().split("\n")
    assert b[0] == "chr1 (249.25Mbp)"
    assert b[11].rstrip() == "        24.50M    24.84M    25.19M      25.60M"
    # (the manual prints 34.5Kbp here; 1 100 001 / 32 = 34 375 bp prints as 34.4K with the %.1f of readable_bps, coverage.c:170)
    assert b[9].endswith("| Histo bin width: 34.4Kbp")
    assert b[10][45:].startswith("Histo max cov:   ") and b[1][:2] == "> " and b[1][10] == "|"
