"""Engine vs CPU oracle on seeded synthetic reads (SURVEY.md 8d generator), through both CLIs.
Covers the bench configurations at sizes the oracle finishes in seconds.  Needs a GPU: -m gpu."""
import os
import subprocess

import pytest

from synth import write_synth_sam

pytestmark = pytest.mark.gpu

CONFIGS = [
    # (id, generator kwargs, argv template)
    ("mpileup30_noref", dict(n_ref=60000, depth=30, read_len=150, seed=42, paired=False), ["mpileup", "{sam}"]),
    ("mpileup30_B", dict(n_ref=60000, depth=30, read_len=150, seed=43, paired=False), ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    ("mpileup30_baq", dict(n_ref=30000, depth=30, read_len=150, seed=44, paired=False), ["mpileup", "-f", "{fa}", "{sam}"]),
    ("mpileup30_baq_100k_reads", dict(n_ref=500000, depth=30, read_len=150, seed=55, paired=True), ["mpileup", "-f", "{fa}", "{sam}"]),
    ("mpileup_baq_indels", dict(n_ref=20000, depth=30, read_len=150, seed=54, paired=False, indel_rate=0.3, max_indel=14), ["mpileup", "-f", "{fa}", "{sam}"]),
    ("mpileup30_pairs_olap", dict(n_ref=60000, depth=30, read_len=150, seed=45, paired=True), ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    ("mpileup_EA_pairs", dict(n_ref=30000, depth=30, read_len=150, seed=46, paired=True), ["mpileup", "-E", "-A", "-f", "{fa}", "{sam}"]),
    ("mpileup300", dict(n_ref=8000, depth=300, read_len=150, seed=47, paired=False), ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    ("mpileup_sO", dict(n_ref=20000, depth=25, read_len=100, seed=48, paired=True), ["mpileup", "-s", "-O", "--output-BP-5", "--output-QNAME", "-Q0", "{sam}"]),
    ("mpileup_a", dict(n_ref=5000, depth=3, read_len=50, seed=49, paired=False), ["mpileup", "-a", "-f", "{fa}", "{sam}"]),
    ("depth30", dict(n_ref=60000, depth=30, read_len=150, seed=50, paired=False), ["depth", "{sam}"]),
    ("depth30_a", dict(n_ref=60000, depth=30, read_len=150, seed=51, paired=False), ["depth", "-a", "{sam}"]),
    ("depth_q20", dict(n_ref=60000, depth=30, read_len=150, seed=52, paired=False), ["depth", "-q", "20", "{sam}"]),
    ("depth_s_J", dict(n_ref=40000, depth=30, read_len=150, seed=53, paired=True, indel_rate=0.05), ["depth", "-s", "-J", "{sam}"]),
]


@pytest.mark.parametrize("cid,gen,argv", CONFIGS, ids=[c[0] for c in CONFIGS])
@pytest.mark.parametrize("window_cols", [None, 1000])
def test_engine_equals_oracle_on_synthetic(tmp_path, oracle_bin, product_bin, cid, gen, argv, window_cols):
    sam, fa = write_synth_sam(str(tmp_path), **gen)
    args = [a.format(sam=sam, fa=fa) for a in argv]
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    env = dict(os.environ)
    if window_cols:
        env["STA_WINDOW_COLS"] = str(window_cols)
    got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == 0, got.stderr.decode()[-500:]
    if got.stdout != want:
        g, w = got.stdout.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(g, w)):
            if a != b:
                pytest.fail("line %d differs\n got: %r\nwant: %r" % (i + 1, a[:300], b[:300]))
        pytest.fail("line count differs: got %d want %d" % (len(g), len(w)))


def test_template_with_three_records_in_the_overlap_hash(tmp_path, oracle_bin, product_bin):
    """overlap_remove is BY NAME (HTSlib sam.c overlap_remove): a record that leaves the pileup buffer deletes the hash entry
    of its template even when the entry belongs to another record of it.  tests/synth_rich.py seed 104 holds such a template (found by
    scripts/hunt4.py): a primary that overlap_push turns away (mate beyond its end), a supplementary alignment that puts the entry, the
    primary leaving the buffer before the mate arrives -- the mate must then find nothing, and the supplementary's ref-skip placeholder
    keeps its quality.  Tile path and generic walker, against the oracle."""
    from synth_rich import write_rich_sam
    sam, fa = write_rich_sam(str(tmp_path), seed=4, n_templates=5000)
    sam2, _ = write_rich_sam(str(tmp_path), seed=104, n_templates=1500)
    for args in (["mpileup", "-B", "-Q", "20", "-f", fa, sam, sam2], ["mpileup", "-f", fa, sam2],
                 ["mpileup", "-B", "-s", "--output-extra", "MD,XS", "-Q", "20", "-f", fa, sam, sam2]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        row = [l for l in want.split(b"\n") if l.startswith(b"c2\t6949\t")]
        assert len(row) == 1
        for env in ({}, {"STA_WINDOW_COLS": "900"}):
            got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert got.returncode == 0, got.stderr.decode()[-500:]
            assert [l for l in got.stdout.split(b"\n") if l.startswith(b"c2\t6949\t")] == row
            assert got.stdout == want, (args, env)


@pytest.mark.parametrize("opts", [["-B"], ["-B", "--output-BP-5", "--output-QNAME"]], ids=["tile", "generic"])
def test_depth_cap_on_bam_input_with_small_windows(tmp_path, oracle_bin, product_bin, opts):
    """-d 12 on a BAM (the lane whose staging arrays hold no CIGARs: the records' pools are cut out on the device) with windows of 3 000
    columns: the producer's bound on where the cap can trigger (driver_pipeline.h cap_may_trigger) has to take the reads' spans from the
    input lane.  It read the empty CIGAR pool, waited for no window, and the device thread's safety net ended the run
    ("the -d cap removed reads ... in a window the producer did not wait for"; found by scripts/hunt4.py seed 6)."""
    from synth_rich import write_rich_sam
    from bamio import sam_to_bam
    sam, fa = write_rich_sam(str(tmp_path), seed=6, n_templates=5000)
    bam = sam_to_bam(sam, str(tmp_path / "rich.bam"), level=1, block=20000)
    args = ["mpileup"] + opts + ["-d", "12", "-f", fa]
    want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    for env in ({"STA_WINDOW_COLS": "3000", "STA_PLP_BATCH": "700"}, {"STA_WINDOW_COLS": "900"}, {}):
        got = subprocess.run([product_bin] + args + [bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, env


_GENERIC_OPTS = {
    "sOx3": ["-s", "-O", "--output-QNAME"],
    # eight extra columns: as many as emit_column_1walk holds cursors for (kernels_plp.hip GEN_NX); flag columns come out in bit order, tags after
    "x8": ["-s", "-O", "--output-BP-5", "--output-extra", "QNAME,FLAG,POS,NM,RG"],
    # eleven: the per-column walks of emit_column stay in charge
    "x11": ["-s", "-O", "--output-BP-5", "--output-extra", "QNAME,FLAG,POS,MAPQ,RNAME,PNEXT,RLEN,NM"],
    "Q30_a": ["-a", "-Q", "30", "-s", "--output-extra", "RLEN,NM"],      # rows without entries ('*' in every string), rows without reads
}


def _long_names(sam):
    """rewrite QNAME r<k> -> a name whose length depends on k (mates keep one name)"""
    out = []
    for line in open(sam):
        if not line.startswith("@"):
            f = line.split("\t", 1)
            k = int(f[0][1:])
            if k % 5:
                line = ("INS%d:%s" % (k, "lane7:tile1101:x" * 2))[: 6 + k % 35] + "\t" + f[1]
        out.append(line)
    open(sam, "w").write("".join(out))


@pytest.mark.parametrize("env", [{"STA_XFAST": "0"}, {"STA_XFAST": "0", "STA_GENERIC_PASSES": "1"}, {"STA_XFAST": "0", "STA_GENERIC_LDS_CAP": "1024"}, {}],
                         ids=["1walk", "passes", "bytestores", "readmajor"])
@pytest.mark.parametrize("opts", list(_GENERIC_OPTS), ids=list(_GENERIC_OPTS))
def test_generic_walker_forms(tmp_path, oracle_bin, product_bin, opts, env):
    """(`readmajor`: the same rows through k_mplp_len_rm<true> + k_mplp_emit_deep<true>, the default since round 6 for up to eight extra
    columns beside -s; STA_XFAST=0 selects the walkers.)
    The generic walker's emit in its two forms (kernels_plp.hip emit_column_1walk: one measuring walk + one writing walk with a cursor
    per string of the row; emit_column: one walk per string, STA_GENERIC_PASSES=1 or more than GEN_NX extra columns), both through the
    LDS slice and with byte stores to the text, on reads with indels (deletion placeholders, inserted sequences in the base string) and
    two input files, against the oracle (bam_plcmd.c:480-855)."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=9000, depth=25, read_len=100, seed=91, paired=True, indel_rate=0.25, max_indel=6)
    d2 = tmp_path / "b"; d2.mkdir()
    sam2, _ = write_synth_sam(str(d2), n_ref=9000, depth=7, read_len=80, seed=92, paired=False, indel_rate=0.1)
    _long_names(sam)          # names of 6..40 characters: the 8-byte loads of put_text and their 4 / 2 / 1-byte tails
    args = ["mpileup", "-B"] + _GENERIC_OPTS[opts] + ["-f", fa, sam, sam2]
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 8000
    for wc in (None, "1000"):
        e = dict(os.environ, **env)
        if wc:
            e["STA_WINDOW_COLS"] = wc
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, (opts, env, wc)


def _spoil_for_extras(sam, seed):
    """every 7th read loses SEQ / QUAL (l_qseq 0: --output-BP-5 of a reverse read goes negative), every 3rd gets a long Z tag and a long name"""
    import random
    rng = random.Random(seed)
    out = []
    k = 0
    for line in open(sam):
        if not line.startswith("@"):
            f = line.rstrip("\n").split("\t")
            k += 1
            if k % 7 == 0:
                f[9] = "*"; f[10] = "*"
            if k % 3 == 0:
                f.append("XZ:Z:" + "tag%d_" % k * rng.choice((1, 2, 5, 9)))
            if k % 4 == 0:
                f.append("XI:i:%d" % rng.choice((0, -7, 255, 70000, -2000000000)))
            line = "\t".join(f) + "\n"
        out.append(line)
    open(sam, "w").write("".join(out))


@pytest.mark.parametrize("opts", [["-O", "--output-BP-5"], ["-s", "--output-BP-5", "--output-extra", "XZ,QNAME,XI"], ["-Q", "0", "-O", "--output-extra", "FLAG,RNEXT,PNEXT,XZ"],
                                  ["-a", "-a", "-Q", "25", "--output-extra", "RNAME,MAPQ,RLEN,POS", "--output-sep", ";", "--output-empty", "."],
                                  ["--no-output-ins", "--no-output-del", "--no-output-ends", "-O", "--output-QNAME"]],
                         ids=["O_BP5", "s_BP5_tags", "Q0_mates", "aa_sep", "no_ins_del_ends"])
def test_extra_columns_on_the_read_major_kernels(tmp_path, oracle_bin, product_bin, opts):
    """Extra columns through k_mplp_len_rm<true> / k_mplp_emit_deep<true> (bam_plcmd.c:727-855): piles deeper than one block of 64 reads
    (the row's first field has no separator only once), reads without SEQ (query positions from the 3' end go negative), fields longer
    than the sixteen bytes the emit holds in registers (names, Z tags), negative and wide integers, indel-rich reads (per-entry route),
    three files, windows of 700 columns -- against the oracle, with the walkers (STA_XFAST=0) as a second witness."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=6000, depth=150, read_len=100, seed=171, paired=True, indel_rate=0.15, max_indel=5)
    _long_names(sam); _spoil_for_extras(sam, 5)
    d2 = tmp_path / "b"; d2.mkdir()
    sam2, _ = write_synth_sam(str(d2), n_ref=6000, depth=9, read_len=60, seed=172, paired=False)
    d3 = tmp_path / "c"; d3.mkdir()
    sam3, _ = write_synth_sam(str(d3), n_ref=6000, depth=70, read_len=151, seed=173, paired=True, indel_rate=0.02)
    _spoil_for_extras(sam3, 6)
    # (-x: a pair with a SEQ-less mate ends the reference's run -- tweak_overlap_quality "fell off the end" -- and the extras do not depend on it)
    args = ["mpileup", "-B", "-x", "-d", "100000"] + opts + ["-f", fa, sam, sam2, sam3]
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 5000
    for env in ({}, {"STA_WINDOW_COLS": "700"}, {"STA_XFAST": "0"}):
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_DEBUG="1", **env))
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, (opts, env)
        assert (b"extra columns on the read-major kernels" in got.stderr) == ("STA_XFAST" not in env)


@pytest.mark.parametrize("form", ["split", "fused"])
def test_depth_in_its_two_forms(tmp_path, oracle_bin, product_bin, form):
    """k_depth_fused as one launch (ticket + decoupled look-back) and split into count | wave scan | emit (kernels_depth.hip; the default from
    256 tiles on, forced here on windows of every size): option sets of bam2depth.c:741-930 on three inputs, windows of 700 and 100 000
    columns and one window per contig, against the oracle."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=150000, depth=12, read_len=100, seed=181, paired=True, indel_rate=0.1, max_indel=8)
    d2 = tmp_path / "b"; d2.mkdir()
    sam2, _ = write_synth_sam(str(d2), n_ref=150000, depth=3, read_len=60, seed=182, paired=False)
    d3 = tmp_path / "c"; d3.mkdir()
    sam3, _ = write_synth_sam(str(d3), n_ref=150000, depth=40, read_len=151, seed=183, paired=True, indel_rate=0.02)
    bed = tmp_path / "r.bed"
    bed.write_text("chrS\t100\t5000\nchrS\t70000\t70001\nchrS\t90000\t149000\n")
    for opts in ([], ["-a"], ["-aa", "-Q", "20", "-q", "15"], ["-J", "-s"], ["-H", "-a", "-b", str(bed)], ["-r", "chrS:60000-120000", "-g", "DUP"], ["-l", "70", "-a"]):
        args = ["depth"] + opts + [sam, sam2, sam3]
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        for env in ({}, {"STA_WINDOW_COLS": "700"}, {"STA_WINDOW_COLS": "100000"}):
            got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_DEPTH_FORM=form, **env))
            assert got.returncode == 0, got.stderr.decode()[-500:]
            assert got.stdout == want, (opts, env, form)


@pytest.mark.parametrize("mode", ["band_even", "band_odd", "long_reads", "general", "plain_E_off"])
def test_baq_kernels_agree_with_oracle(tmp_path, oracle_bin, product_bin, mode):
    """The band-in-registers BAQ kernels (band width 7 with I rows stored every second row, band width 8 with every row; per-row
    MAP states in LDS) for even and odd read lengths, the same kernels with the states in the scratch slot (reads longer than
    256 bases) and the general-band kernel (STA_BAQ_FORCE_SLOW=1) must all reproduce the oracle."""
    read_len = {"band_even": 120, "band_odd": 121, "long_reads": 301, "general": 120, "plain_E_off": 150}[mode]
    sam, fa = write_synth_sam(str(tmp_path), n_ref=12000, depth=40, read_len=read_len, seed=61, paired=True, indel_rate=0.2, max_indel=10)
    args = ["mpileup", "-f", fa, sam] if mode == "plain_E_off" else ["mpileup", "-E", "-f", fa, sam]
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    env = dict(os.environ)
    if mode == "general":
        env["STA_BAQ_FORCE_SLOW"] = "1"
    got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert got.returncode == 0, got.stderr.decode()[-500:]
    assert got.stdout == want


@pytest.mark.parametrize("extra", [[], ["-q", "25"], ["-B"], ["-s", "--output-MQ"]])
def test_adjust_mq_equals_oracle(tmp_path, oracle_bin, product_bin, extra):
    """-C / --adjust-MQ (HTSlib sam_cap_mapq after BAQ; no reference golden uses it, so the oracle restatement is the
    only checker): mismatch-rich synthetic reads and the reference's real-data fixture with soft clips."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=15000, depth=30, read_len=100, seed=81, paired=True, sub_rate=0.03, indel_rate=0.05)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dat")
    for args in (["mpileup", "-C", "50"] + extra + ["-f", fa, sam],
                 ["mpileup", "-C", "40"] + extra + ["-f", os.path.join(g, "mpileup.ref.fa"), os.path.join(g, "mpileup.1.sam")]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, args
        assert want != subprocess.run([oracle_bin] + [a for a in args if a not in ("-C", "50", "40")], stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL, check=True).stdout      # -C changes something in this data


def test_output_extra_tags_and_rnext_equal_oracle(tmp_path, oracle_bin, product_bin):
    """--output-extra with aux tags (host-formatted text columns), RNEXT, --output-sep / --output-empty: BAM and SAM inputs
    of the reference's fixtures plus synthetic pairs (no tags at all: every entry prints the --output-empty character)."""
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sam, fa = write_synth_sam(str(tmp_path), n_ref=8000, depth=20, read_len=100, seed=91, paired=True)
    cases = [
        ["mpileup", "--output-extra", "NM,RG,FLAG,RNEXT,XT,MD", os.path.join(g, "mpileup", "mpileup.1.bam")],
        ["mpileup", "-s", "--output-extra", "RNEXT,PNEXT,AS,XS,NM", "--output-sep", ";", "--output-empty", "-", "-a", os.path.join(g, "dat", "mpileup.1.sam")],
        ["mpileup", "--output-extra", "QNAME,RNEXT,ZZ", "--output-empty", "?", "-f", fa, sam],
        ["mpileup", "--output-extra", "RNEXT", "-Q", "0", os.path.join(g, "mpileup", "mpileup.1.bam"), os.path.join(g, "mpileup", "mpileup.2.bam")],
    ]
    for args in cases:
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, args
        assert len(want) > 1000


def test_overlap_placeholder_sees_resolved_quality_only_if_mate_triggers_column(tmp_path, oracle_bin, product_bin):
    """HTSlib resolves a mate pair when the second mate is PUSHED; a column is handed out as soon as a read starting
    beyond it has been pushed.  A deletion placeholder shows the quality of the next base, which may lie in the overlap:
    columns before the mate's start see the resolved value only when the mate itself is the first read beyond the column.
    Hand-made pairs: deletion / ref skip ending exactly at the mate's start, with and without other reads starting in
    between, several names (the keeper of a pair is chosen by a hash of the name), default and 30-column windows."""
    import re

    def sam(name, cigar, extra_reads):
        qlen = sum(int(n) for n, op in re.findall(r"(\d+)([MIDNS=X])", cigar) if op in "MIS=X")
        a_q = ("5" * 50 + "&" + "5" * 40)[:qlen]      # base 50 (the first one after the deletion) has quality 5
        lines = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c\tLN:1000",
                 "%s\t99\tc\t101\t60\t%s\t=\t153\t122\t%s\t%s" % (name, cigar, "A" * qlen, a_q)]
        for pos in extra_reads:
            lines.append("x%d\t0\tc\t%d\t60\t30M\t*\t0\t0\t%s\t%s" % (pos, pos, "C" * 30, "I" * 30))
        lines.append("%s\t147\tc\t153\t60\t70M\t=\t101\t-122\t%s\t%s" % (name, "A" * 70, "?" * 70))
        return "\n".join(lines) + "\n"

    n = 0
    for name in ("p1", "p2", "q7"):
        for extra in ([], [152], [153], [151, 152], [140]):
            for cig in ("50M2D20M", "50M2N20M", "49M1D1N1D20M", "50M4D18M"):
                path = tmp_path / ("t%d.sam" % n); n += 1
                path.write_text(sam(name, cig, extra))
                for env_extra in ({}, {"STA_WINDOW_COLS": "30"}):
                    args = ["mpileup", "-Q", "0", str(path)]
                    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                    got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env_extra))
                    assert got.returncode == 0, got.stderr.decode()[-300:]
                    assert got.stdout == want, (name, extra, cig, env_extra)
    # the data really exercises both outcomes (resolved and unresolved placeholder qualities)
    a = subprocess.run([oracle_bin, "mpileup", "-Q", "0", str(tmp_path / "t0.sam")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    b = subprocess.run([oracle_bin, "mpileup", "-Q", "0", str(tmp_path / "t4.sam")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert [l for l in a.split(b"\n") if l.startswith(b"c\t151\t")] != [l for l in b.split(b"\n") if l.startswith(b"c\t151\t")]


@pytest.mark.parametrize("window", [None, "3000", "700"])
def test_depth_cap_is_exact_across_windows(tmp_path, oracle_bin, product_bin, window):
    """-d / bam_mplp_set_maxcnt is order dependent (live-node count when a read arrives): reads it drops must stay out of
    later windows and reads it kept must not be re-tested there (STA_AUX_ACCEPTED).  400x pairs, cap 50 and 120."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=12000, depth=400, read_len=150, seed=204, paired=True, indel_rate=0.03)
    env = dict(os.environ)
    if window:
        env["STA_WINDOW_COLS"] = window; env["STA_PLP_BATCH"] = window
    for args in (["mpileup", "-B", "-d", "50", "-f", fa, sam], ["mpileup", "-d", "120", "-f", fa, sam], ["plpdump", "-d", "100", sam]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert got.returncode == 0, got.stderr.decode()[-300:]
        assert got.stdout == want, args


def test_overlap_placeholder_with_mate_beyond_window_end(tmp_path, oracle_bin, product_bin):
    """Same corner as above, across a window boundary: the deletion / ref skip straddles the end of a window and the mate
    starts in the next one.  The window must still see the mate (host_pump lookahead, plp_api peek staging), because the
    mate is the push that releases the window's last columns.  Window ends swept over the whole run."""
    def sam(skip, extra):
        qlen = 70
        a_q = ("5" * 50 + "&" + "5" * 40)[:qlen]
        mate = 101 + 50 + skip
        lines = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c\tLN:2000",
                 "p1\t99\tc\t101\t60\t50M%dN20M\t=\t%d\t%d\t%s\t%s" % (skip, mate, 50 + skip + 70, "A" * qlen, a_q)]
        for pos in extra:
            lines.append("x%d\t4\tc\t%d\t0\t*\t*\t0\t0\t%s\t%s" % (pos, pos, "C" * 30, "I" * 30))       # filtered: never pushed
        lines.append("p1\t147\tc\t%d\t60\t70M\t=\t101\t%d\t%s\t%s" % (mate, -(50 + skip + 70), "A" * 70, "?" * 70))
        return "\n".join(lines) + "\n"

    n = 0
    for skip in (40, 200):
        for extra in ([], [120, 160]):
            path = tmp_path / ("w%d.sam" % n); n += 1
            path.write_text(sam(skip, extra))
            for args in (["mpileup", "-Q", "0", str(path)], ["mpileup", str(path)], ["plpdump", "-x", str(path)]):
                want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                for w in ("25", "60", "77", "100", "130", "190", "250"):
                    got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                         env=dict(os.environ, STA_WINDOW_COLS=w, STA_PLP_BATCH="1"))
                    assert got.returncode == 0, got.stderr.decode()[-300:]
                    assert got.stdout == want, (skip, extra, args[:2], w)


@pytest.mark.parametrize("seed", [1, 7])
def test_engine_equals_oracle_on_messy_multicontig_input(tmp_path, oracle_bin, product_bin, seed):
    """tests/synth_rich.py: clips, long ref skips, pads, =/X, every flag, mates on other contigs, unmapped mates, repeated
    names, missing SEQ/QUAL, RG/NM tags over three contigs; default and 900-column windows (scripts/hunt3.py in small)."""
    from synth_rich import write_rich_sam
    sam, fa = write_rich_sam(str(tmp_path), seed=seed, n_templates=2500)
    bed = tmp_path / "r.bed"; bed.write_text("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
    cases = [["mpileup", "-f", fa, sam], ["mpileup", "-B", "-Q", "0", "-s", "-O", "--output-extra", "FLAG,RNEXT,NM,RG", "-f", fa, sam],
             ["mpileup", "-B", "-l", str(bed), "-a", sam], ["mpileup", "-r", "c3:1000-30000", "-d", "15", "-f", fa, sam],
             ["mpileup", "-C", "50", "-f", fa, sam], ["depth", "-a", "-s", "-J", sam], ["plpdump", "-x", sam], ["plpdump", "-d", "12", sam],
             ["coverage", sam], ["bedcov", "-j", "-d", "8", "-c", str(bed), sam]]
    for args in cases:
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        for envx in ({}, {"STA_WINDOW_COLS": "900", "STA_PLP_BATCH": "700"}):
            got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx))
            assert got.returncode == 0, got.stderr.decode()[-300:]
            assert got.stdout == want, (args[:3], envx)


def test_overlap_rewrites_beyond_first_mates_end_survive_window_changes(tmp_path, oracle_bin, product_bin):
    """HTSlib's tweak_overlap_quality has a deletion branch that rewrites bases of the later mate BEYOND the earlier mate's
    end (earlier mate sits after a deletion, later mate jumps ahead over a ref skip).  Every window re-derives the
    resolution from the pushed records, so the earlier mate has to stay staged while the later one is live
    (Pump::retire keep_mates, plp_api ghosts)."""
    lines = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c\tLN:2000"]
    for k, (name, off) in enumerate((("p1", 0), ("p2", 300), ("q7", 600), ("zz9", 900))):
        a, b = 101 + off, 121 + off
        lines.append("%s\t99\tc\t%d\t60\t20M4D30M\t=\t%d\t120\t%s\t%s" % (name, a, b, "A" * 50, "F" * 50))
        lines.append("x%d\t0\tc\t%d\t60\t30M\t*\t0\t0\t%s\t%s" % (k, a + 5, "C" * 30, "I" * 30))
        lines.append("%s\t147\tc\t%d\t60\t10M60N30M\t=\t%d\t-120\t%s\t%s" % (name, b, a, "A" * 40, "".join(chr(40 + i) for i in range(40))))
    path = tmp_path / "q.sam"; path.write_text("\n".join(lines) + "\n")
    outs = set()
    for args in (["mpileup", "-Q", "0", str(path)], ["mpileup", str(path)], ["plpdump", "-x", str(path)]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        outs.add(want)
        for w in (None, "20", "33", "57", "64", "90", "150"):
            env = dict(os.environ)
            if w: env.update(STA_WINDOW_COLS=w, STA_PLP_BATCH="1")
            got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert got.returncode == 0, got.stderr.decode()[-300:]
            assert got.stdout == want, (args[:2], w)
    # the data does hit the branch: without overlap resolution the later mate's first bases after the skip keep their quality
    raw = subprocess.run([oracle_bin, "mpileup", "-Q", "0", "-x", str(path)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    res = subprocess.run([oracle_bin, "mpileup", "-Q", "0", str(path)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    pick = lambda t: [l for l in t.split(b"\n") if l.startswith(b"c\t191\t")]
    assert pick(raw) != pick(res)


def test_depth_cap_inside_a_deep_amplicon(tmp_path, oracle_bin, product_bin):
    """the default -d 8000 meeting a 12 000x amplicon inside an ordinary 20x contig (and -d 500 meeting it everywhere): the exact
    replay of bam_plp_push's cap now walks only the reads between the first and the last one the cap can possibly drop -- the text
    must stay the oracle's, in one window and with the amplicon cut by window boundaries"""
    import numpy as np
    from synth import synth_ref, synth_reads, synth_hotspot, write_sam, write_fasta
    n = 40000
    ref = synth_ref(n, seed=5)
    rd = synth_hotspot(ref, synth_reads(ref, depth=20, read_len=150, seed=6, indel_rate=0.02), hot_start=17000, hot_len=300, hot_depth=12000, seed=7)
    sam, fa = str(tmp_path / "h.sam"), str(tmp_path / "h.fa")
    write_sam(sam, rd, "chrS", n)
    write_fasta(fa, "chrS", ref)
    for args in (["mpileup", "-B", "-f", fa, sam], ["mpileup", "-B", "-d", "500", "-f", fa, sam], ["mpileup", "-d", "9000", "-f", fa, "-r", "chrS:16900-17500", sam]):
        want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        for env in ({}, {"STA_WINDOW_COLS": "17100"}, {"STA_WINDOW_COLS": "333"}):
            got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert got.returncode == 0, got.stderr.decode()[-400:]
            assert got.stdout == want, (args, env)


def test_depth_cap_meeting_the_lookahead_record_of_a_window(tmp_path, oracle_bin, product_bin):
    """The corner DESIGN.md section 2 used to list as a residual: the record that releases a window's last columns (the first one
    beyond the window that reaches bam_plp_push) being dropped by the -d cap, while a deletion placeholder of an overlapping pair
    looks at a base inside the overlap.  It cannot be observed: bam_plp_push's cap only ever drops a read that is NOT the first
    pushed read of its start position (iter->pos reaches a position only after a read starting there was pushed), and the record the
    host stops its lookahead at is the first pushed record of a position beyond the window; where the host cannot tell which records
    are pushed (-G, -l) it stages the whole run and the device picks the releasing read among them, skipping cap-dropped ones
    (read_advances_iterator).  Constructed here: a pile of reads on the mate's start position, the mate in front of, inside and behind
    the pile, caps below / at / above the pile height, every window end across the run, both lookahead modes."""
    def sam(pile, mate_at, skip_cigar):
        qlen = 70
        a_q = ("5" * 50 + "&" + "5" * 40)[:qlen]
        lines = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c\tLN:2000", "@RG\tID:a\tSM:s", "@RG\tID:b\tSM:s",
                 "p1\t99\tc\t101\t60\t%s\t=\t153\t122\t%s\t%s\tRG:Z:a" % (skip_cigar, "A" * qlen, a_q)]
        for k in range(8):       # depth in front of the pile, so that small caps bite at 153
            lines.append("d%d\t0\tc\t%d\t60\t90M\t*\t0\t0\t%s\t%s\tRG:Z:a" % (k, 110 + k, "G" * 90, "H" * 90))
        mate = "p1\t147\tc\t153\t60\t70M\t=\t101\t-122\t%s\t%s\tRG:Z:a" % ("A" * 70, "?" * 70)
        for k in range(pile + 1):
            if k == mate_at:
                lines.append(mate)
            if k < pile:
                lines.append("x%d\t0\tc\t153\t60\t30M\t*\t0\t0\t%s\t%s\tRG:Z:a" % (k, "C" * 30, "I" * 30))
        lines.append("z\t0\tc\t400\t60\t30M\t*\t0\t0\t%s\t%s\tRG:Z:a" % ("T" * 30, "I" * 30))
        return "\n".join(lines) + "\n"

    rg = tmp_path / "rg.txt"; rg.write_text("b\n")                 # -G b: excludes nothing here, but the host can no longer tell who is pushed
    bed = tmp_path / "all.bed"; bed.write_text("c\t0\t2000\n")
    n = 0
    outcomes = set()
    for cig in ("50M2D20M", "50M2N20M"):
        for pile, mate_at in ((6, 0), (6, 3), (12, 12)):
            path = tmp_path / ("c%d.sam" % n); n += 1
            path.write_text(sam(pile, mate_at, cig))
            for cap in ("4", "12", "40"):
                for mode in ([], ["-G", str(rg)], ["-l", str(bed)]):
                    args = ["mpileup", "-Q", "0", "-d", cap] + mode + [str(path)]
                    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                    outcomes.add(tuple(l for l in want.split(b"\n") if l.startswith(b"c\t151\t") or l.startswith(b"c\t153\t")))
                    for w in (None, "51", "52", "53"):       # window ends at 152, 153, 154 (origin 101)
                        env = dict(os.environ) if w is None else dict(os.environ, STA_WINDOW_COLS=w, STA_PLP_BATCH="1")
                        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
                        assert got.returncode == 0, got.stderr.decode()[-300:]
                        assert got.stdout == want, (cig, pile, mate_at, cap, mode, w)
    assert len(outcomes) >= 4      # the caps, the pile and the mate's place in it do change what columns 151 and 153 show


def _rewrite_some_records(sam, every, fn):
    """fn(fields) -> fields for every `every`-th alignment line"""
    out, k = [], 0
    for line in open(sam):
        if not line.startswith("@"):
            k += 1
            if k % every == 0:
                line = "\t".join(fn(line.rstrip("\n").split("\t"))) + "\n"
        out.append(line)
    open(sam, "w").write("".join(out))


@pytest.mark.parametrize("opts", [["-6"], ["-6", "-E"], ["-6", "-B"], ["-6", "-Q", "0", "-A"]], ids=["baq", "E", "B", "Q0_A"])
def test_illumina13_shift_comes_before_the_missing_quality_test_of_baq(tmp_path, oracle_bin, product_bin, opts):
    """mplp_func rewrites the qualities for -6 (bam_plcmd.c:431-435) BEFORE it calls sam_prob_realn (:451), whose `qual[0] == 0xff: do
    nothing` therefore sees 224 on a record without QUAL: such a read IS realigned under -6 (its qualities come out as the BAQ values,
    not as '~').  Found by scripts/hunt5.py on the CPU emulation of the kernels (round 5): the engine tested the unshifted byte."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=12000, depth=12, read_len=100, seed=611, paired=True, indel_rate=0.05)
    _rewrite_some_records(sam, 4, lambda f: f[:10] + ["*"] + f[11:])
    args = ["mpileup"] + opts + ["-f", fa, sam]
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 10000
    for wc in (None, "700"):
        e = dict(os.environ)
        if wc:
            e["STA_WINDOW_COLS"] = wc
        got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, (opts, wc)
    if opts == ["-6"]:
        # and the realignment is visible: without -6 the same reads print '~' for every base (0xff + 33 capped), with -6 and BAQ they do not
        plain = subprocess.run([product_bin, "mpileup", "-6", "-B", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert plain.count(b"~") > want.count(b"~")


@pytest.mark.parametrize("args", [["calmd", "--no-PG", "-r", "-A"], ["calmd", "--no-PG", "-r", "-A", "-E", "-n", "3"], ["mpileup"], ["mpileup", "-6"],
                                  ["mpileup", "-E", "-Q", "0"]], ids=["calmd_rA", "calmd_rAE_n3", "mpileup", "mpileup_6", "mpileup_E"])
def test_bq_tag_is_not_applied_to_records_the_realigner_turns_away(tmp_path, oracle_bin, product_bin, args):
    """sam_prob_realn returns on an unmapped record, a record without bases and a record whose first quality is 0xff BEFORE it looks for
    BQ:Z (realn.c): a read without QUAL keeps 0xff everywhere whatever its tag says (`calmd -r -A` prints '*', mpileup '~');
    under -6 the shifted byte is 224 and the tag IS applied.  The engine applies the tag pool to every byte up front (k_qual_prep) and
    puts such reads back in k_prep_reads.  Found by scripts/hunt6.py on the CPU emulation of the kernels (round 5)."""
    import random
    rnd = random.Random(77)
    sam, fa = write_synth_sam(str(tmp_path), n_ref=9000, depth=10, read_len=100, seed=613, paired=True, indel_rate=0.03)

    def tag(f):
        bq = "BQ:Z:" + "".join(chr(64 + (rnd.randint(1, 40) if rnd.random() < 0.3 else 0)) for _ in f[9])
        k = rnd.random()
        if k < 0.4:
            f = f[:10] + ["*"] + f[11:]                                  # no QUAL: the tag must be ignored (not under -6)
        elif k < 0.5:
            f = [f[0], str(int(f[1]) | 4)] + f[2:]                       # unmapped but placed: calmd hands it back untouched
        return f + [bq]
    _rewrite_some_records(sam, 3, tag)
    files = [sam, fa] if args[0] == "calmd" else ["-f", fa, sam]
    want = subprocess.run([oracle_bin] + args + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert len(want) > 100000
    for wc in (None, "900"):
        e = dict(os.environ)
        if wc:
            e["STA_WINDOW_COLS"] = wc
        got = subprocess.run([product_bin] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, (args, wc)
    if args == ["calmd", "--no-PG", "-r", "-A"]:
        # the rule is visible: records without QUAL come out with '*' in the quality column
        assert sum(1 for l in want.split(b"\n") if l and not l.startswith(b"@") and l.split(b"\t")[10] == b"*" and b"BQ:Z:" in l) > 20


@pytest.mark.parametrize("args", [["calmd", "--no-PG", "-r"], ["calmd", "--no-PG", "-r", "-A", "-E"], ["calmd", "--no-PG", "-e"], ["mpileup"], ["mpileup", "-B", "-a"]],
                         ids=["calmd_r", "calmd_rAE", "calmd_e", "mpileup", "mpileup_Ba"])
def test_fasta_contig_shorter_than_the_header_says(tmp_path, oracle_bin, product_bin, args):
    """A FASTA whose contig ends before @SQ LN.  mpileup skips reads that START behind the sequence's end (bam_plcmd.c:440-445); calmd has
    no such rule: bam_md.c:461-476 hands every placed record to sam_prob_realn, whose window is clipped at the end of the sequence
    (realn.c: `i >= ref_len: xe = i`), so a read starting up to bw/2 columns behind the end still gets its BQ:Z from the last few
    reference bases.  The engine ran calmd through mpileup's skip (found by scripts/hunt6.py on the CPU emulation, round 5).  A window
    wholly behind the end (l_ref <= 0) is undefined in the reference (probaln_glocal returns before writing state[] / q[]): engine and
    oracle leave such a record alone."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=9000, depth=12, read_len=100, seed=614, paired=True, indel_rate=0.03)
    lines = open(fa).read().split("\n")
    # cut so that a plain read starts one column behind the last base (its window still holds two reference bases)
    cut = next(int(f[3]) - 2 for f in (l.split("\t") for l in open(sam) if not l.startswith("@")) if int(f[3]) > 7000 and f[5] == "100M")
    seq = "".join(l for l in lines[1:] if l)[:cut]
    with open(fa, "w") as fh:
        fh.write(lines[0] + "\n")
        for i in range(0, len(seq), 60):
            fh.write(seq[i:i + 60] + "\n")
    if os.path.exists(fa + ".fai"):
        os.remove(fa + ".fai")
    files = [sam, fa] if args[0] == "calmd" else ["-f", fa, sam]
    want = subprocess.run([oracle_bin] + args + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert len(want) > 100000
    for wc in (None, "900"):
        e = dict(os.environ)
        if wc:
            e["STA_WINDOW_COLS"] = wc
        got = subprocess.run([product_bin] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, (args, wc)
    if args == ["calmd", "--no-PG", "-r"]:
        # the read the rule is about exists: it starts behind the last base and carries a computed BQ:Z
        assert any(int(l.split(b"\t")[3]) == cut + 2 and b"BQ:Z:" in l for l in want.split(b"\n") if l and not l.startswith(b"@"))


@pytest.mark.parametrize("cmd", ["mpileup", "calmd"])
def test_adjust_mq_on_records_without_seq(tmp_path, oracle_bin, product_bin, cmd):
    """-C on a record whose SEQ is '*' under a CIGAR with M operations: HTSlib's sam_cap_mapq walks seq / qual behind the record
    (undefined in the reference).  Engine and oracle take a base that is not there as neither a mismatch nor a clipped quality: the read
    keeps min(MAPQ, threshold).  Found by scripts/hunt5.py (engine and oracle each read their own neighbouring bytes)."""
    sam, fa = write_synth_sam(str(tmp_path), n_ref=9000, depth=10, read_len=100, seed=612, paired=False, sub_rate=0.02)
    _rewrite_some_records(sam, 5, lambda f: f[:9] + ["*", "*"] + f[11:])
    args = (["mpileup", "-C", "20", "-s", "-f", fa, sam] if cmd == "mpileup" else ["calmd", "--no-PG", "-C", "20", sam, fa])
    want = subprocess.run([oracle_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    got = subprocess.run([product_bin] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()[-500:]
    assert got.stdout == want and len(want) > 100000


def _add_upstream_supplementaries(sam, every, gap_lo, gap_hi, seed):
    """For every `every`-th proper pair add a supplementary alignment of the LATER mate (flag 2048 | its flags; RNEXT / PNEXT = the earlier
    mate, as an aligner writes them) that lies wholly upstream of the pair, gap_lo..gap_hi columns in front of it: three records of one
    template, the first of which has left the pileup buffer (and taken the template's overlap-hash entry with it) before the pair arrives."""
    import random
    rnd = random.Random(seed)
    head, recs = [], []
    for line in open(sam):
        (head if line.startswith("@") else recs).append(line)
    by_name = {}
    for line in recs:
        f = line.rstrip("\n").split("\t")
        by_name.setdefault(f[0], []).append(f)
    extra, k = [], 0
    for name, fs in by_name.items():
        if len(fs) != 2 or not (int(fs[0][1]) & 2):
            continue
        k += 1
        if k % every:
            continue
        a, b = fs                       # in position order
        L = rnd.randint(30, 60)
        spos = int(a[3]) - rnd.randint(gap_lo, gap_hi) - L
        if spos < 1:
            continue
        extra.append((spos, "\t".join([name, str(int(b[1]) | 2048), b[2], str(spos), b[4], "%dM" % L, "=", a[3], "0", b[9][:L], b[10][:L]]) + "\n"))
    allr = [(int(l.split("\t")[3]), i, l) for i, l in enumerate(recs)] + [(p, -1, l) for p, l in extra]
    allr.sort(key=lambda t: (t[0], t[1]))
    open(sam, "w").write("".join(head) + "".join(t[2] for t in allr))
    return len(extra)


@pytest.mark.parametrize("gap", [(10, 120), (0, 4)], ids=["gap10_120", "gap0_4"])
@pytest.mark.parametrize("opts", [[], ["-l", "{bed}"], ["-B", "-C", "50"]], ids=["plain", "bed", "B_C50"])
def test_supplementary_upstream_of_a_pair_across_window_cuts(tmp_path, oracle_bin, product_bin, opts, gap):
    """(Round 5's per-window replay; since round 6 the input lane keeps HTSlib's overlap hash itself in file order -- host_names.h -- and these
    are regression tests of that state machine.)  A record kept staged only for its mate's sake must not look like the holder of the template's overlap-hash
    entry in the next window when the reference freed it long ago: bam_plp_next frees a node once a read beyond its end was pushed, and
    overlap_remove deletes the entry BY NAME.  Three records of one template -- a supplementary alignment upstream of an overlapping
    primary pair -- with a window cut between them: the engine paired the supplementary with the first mate and left the real pair
    unresolved (found by scripts/hunt5.py on the CPU emulation, round 5).  Where the host knows who is pushed it drops the freed record at
    the cut; where it does not (-l, -C ...) it carries the records in between along and the device's replay decides.
    gap0_4: the supplementary ends right in front of the first primary and no read starts in between, so it is STILL in the buffer when
    that primary is pushed: the primary finds its entry, deletes it, and the pair is never resolved.  "Freed before its mate" has to mean
    before the NEXT record of its template; a record whose span ends at a cut stays while no pushed read has started beyond its end; and the
    records of a template with more than two records all stay while one of them does -- the engine dropped the supplementary at a later
    cut and let the primaries pair up in the replay (found by scripts/hunt6.py seed 29 on the CPU emulation, round 5)."""
    from bamio import sam_to_bam
    sam, fa = write_synth_sam(str(tmp_path), n_ref=6000, depth=12, read_len=200, seed=613, paired=True)      # insert ~ 300: every pair overlaps
    assert _add_upstream_supplementaries(sam, 2, gap[0], gap[1], 5) > 40
    bed = str(tmp_path / "r.bed")
    open(bed, "w").write("chrS\t0\t2500\nchrS\t2600\t5000\n")
    bam = sam_to_bam(sam, str(tmp_path / "s.bam"), level=1, block=3000)
    args = ["mpileup"] + [o.format(bed=bed) for o in opts] + ["-f", fa]
    want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 4000
    for wr in ("2", "3", "4", "5", "6", "8", "13"):
        for inp in (sam, bam):
            got = subprocess.run([product_bin] + args + [inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_WINDOW_READS=wr))
            assert got.returncode == 0, got.stderr.decode()[-500:]
            assert got.stdout == want, (opts, wr, os.path.basename(inp))
    got = subprocess.run([product_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_WINDOW_READS="5", STA_IO_LANE="rec"))
    assert got.returncode == 0 and got.stdout == want, (opts, "record-at-a-time lane")


def _add_four_record_templates(sam, every, seed):
    """For every `every`-th proper pair whose mates lie 120-500 columns apart add TWO supplementary alignments of the second mate without the
    proper-pair bit: one a few columns behind the first primary, one a few columns in front of the second primary, each naming the primary
    next to it as its mate -- the shape of template t495 of scripts/hunt8.py seed 104 (primary 115 @7070, supplementary 2225 @7076,
    supplementary 2225 @7416, primary 179 @7419).  In bam2depth.c's hash the first primary does not insert (its mate starts beyond its end),
    the first supplementary does, the second one finds that entry 340 columns later and takes it out, and the second primary inserts: it
    is NOT clipped.  A replay that sees only the last two records clips it."""
    import random
    rnd = random.Random(seed)
    head, recs = [], []
    for line in open(sam):
        (head if line.startswith("@") else recs).append(line)
    by_name = {}
    for line in recs:
        f = line.rstrip("\n").split("\t")
        by_name.setdefault(f[0], []).append(f)
    extra, k = [], 0
    for name, fs in by_name.items():
        if len(fs) != 2 or not (int(fs[0][1]) & 2):
            continue
        a, b = fs
        if not 120 <= int(b[3]) - int(a[3]) <= 500:
            continue
        k += 1
        if k % every:
            continue
        flag = (int(b[1]) & ~2) | 2048
        for base, mate, lo, hi in ((int(a[3]), a, 1, 9), (int(b[3]), b, -9, -1)):
            L = rnd.randint(40, 90)
            spos = base + rnd.randint(lo, hi)
            if spos < 1:
                continue
            extra.append((spos, "\t".join([name, str(flag), b[2], str(spos), b[4], "%dM" % L, "=", mate[3], "0", b[9][:L], b[10][:L]]) + "\n"))
    allr = [(int(l.split("\t")[3]), i, l) for i, l in enumerate(recs)] + [(p, len(recs) + j, l) for j, (p, l) in enumerate(extra)]
    allr.sort(key=lambda t: (t[0], t[1]))
    open(sam, "w").write("".join(head) + "".join(t[2] for t in allr))
    return len(extra)


def test_four_records_of_a_template_across_window_cuts(tmp_path, oracle_bin, product_bin):
    """Both name hashes of the path are sequential state over the whole file: depth -s's "never forgets" (bam2depth.c:598-623: an entry leaves
    only when a record of its name finds it), and HTSlib's overlap hash loses an entry BY NAME whenever any record of the template leaves
    the pileup buffer.  Round 5 replayed them per window from the staged records and got templates with four records wrong
    at window cuts (scripts/hunt8.py seed 104, `c3 7419` one count short under windows of 4 and 13 reads and of 37 columns,
    SAM and BAM lanes).  The input lanes now keep both hashes themselves in file order (host_names.h) and stage what every record found;
    this input has such a template every few hundred columns."""
    from bamio import sam_to_bam
    sam, fa = write_synth_sam(str(tmp_path), n_ref=9000, depth=10, read_len=100, seed=917, paired=True)      # mates ~200 columns apart
    assert _add_four_record_templates(sam, 3, 11) > 60
    bam = sam_to_bam(sam, str(tmp_path / "s.bam"), level=1, block=3000)
    for args in (["depth", "-s"], ["depth", "-s", "-J"], ["depth", "-s", "-g", "SECONDARY"], ["mpileup", "-f", fa], ["mpileup", "-B", "-Q", "0", "-f", fa]):
        want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert want.count(b"\n") > 6000
        for env in ({"STA_WINDOW_READS": "4"}, {"STA_WINDOW_READS": "13"}, {"STA_WINDOW_COLS": "37"}, {"STA_WINDOW_READS": "2"}, {}):
            for inp, lane in ((sam, "chunk"), (bam, "chunk"), (bam, "rec")):
                got = subprocess.run([product_bin] + args + [inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_IO_LANE=lane, **env))
                assert got.returncode == 0, got.stderr.decode()[-500:]
                assert got.stdout == want, (args, env, os.path.basename(inp), lane)


@pytest.mark.parametrize("gap", [(10, 120), (0, 4)], ids=["gap10_120", "gap0_4"])
def test_depth_s_with_three_records_of_a_template_across_window_cuts(tmp_path, oracle_bin, product_bin, gap):
    """depth -s (bam2depth.c:598-623): the first-seen record of a template puts its name and end into a hash, the next one finds it, is
    clipped below that end and takes the entry out -- so a THIRD record inserts again.  The engine replays the hash per window from the
    staged records: with a supplementary alignment upstream of an overlapping pair and a cut behind it, the first primary looked like the
    first-seen record and the second was clipped, where the reference had let the first primary consume the supplementary's entry and left
    the second alone (found by scripts/hunt6.py seed 47 with 5-read windows, round 5).  A record that has ended now stays staged while its
    mate does (round 5); since round 6 the input lane keeps the name hash itself in file order and stages every record's clip column
    (host_names.h, sta_reads.olap_clip), so no window depends on records it does not stage."""
    from bamio import sam_to_bam
    sam, fa = write_synth_sam(str(tmp_path), n_ref=6000, depth=12, read_len=200, seed=615, paired=True)      # insert ~ 300: every pair overlaps
    assert _add_upstream_supplementaries(sam, 2, gap[0], gap[1], 6) > 40
    bam = sam_to_bam(sam, str(tmp_path / "s.bam"), level=1, block=3000)
    for args in (["depth", "-s"], ["depth", "-s", "-J", "-aa", "-Q", "3"]):
        want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        plain = subprocess.run([oracle_bin] + [a for a in args if a != "-s"] + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert want.count(b"\n") > 4000 and want != plain
        for wr in ("2", "3", "5", "8", "13"):
            for inp, lane in ((sam, "chunk"), (bam, "chunk"), (sam, "rec")):
                got = subprocess.run([product_bin] + args + [inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, STA_WINDOW_READS=wr, STA_IO_LANE=lane))
                assert got.returncode == 0, got.stderr.decode()[-500:]
                assert got.stdout == want, (args, wr, os.path.basename(inp), lane)

