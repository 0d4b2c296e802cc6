"""`samtools stats`, the coverage distribution (COV section; SURVEY.md 8(f) row 3, second half).

Reference: the pileup round buffer of stats.c:311-391 fed by :1452-1508.  Its numbers depend on the buffer's bookkeeping (blocks
folding back modulo 5 x the longest read, the slot a whole-buffer flush leaves behind, the short copy when the buffer grows), so:
  * the oracle (oracle/o_stats.c) restates the buffer itself and is pinned by the COV lines of test/stat/*.expected (20 runs);
  * the engine keeps no buffer -- driver_stats.cpp turns aligned blocks into sorted marks with the bookkeeping applied, the device bins
    runs between marks (kernels_statcov.hip).  The host half is checked here on the CPU: `--marks-out` writes the marks and a few lines
    of Python (runs_to_section) do what the kernel does; the kernel itself is checked on the GPU against the same oracle output.
"""
import collections
import os
import random
import subprocess

import pytest

import regcases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stat")
HEAD = "# Coverage distribution. Use `grep ^COV | cut -f 2-` to extract this part.\n"


def bins(cmin, cmax, cstep):
    """stats.c:2396-2406"""
    if cstep > cmax - cmin + 1:
        cstep = cmax - cmin
        if cstep <= 0:
            cstep = 1
    ncov = 3 + (cmax - cmin) // cstep
    cmax = cmin + ((cmax - cmin) // cstep + 1) * cstep - 1
    return cmin, cmax, cstep, ncov


def runs_to_section(path, c=(1, 1000, 1)):
    """what k_statcov_bins does with sorted marks, and the lines of stats.c:1884-1892"""
    cmin, cmax, cstep, ncov = bins(*c)
    cov = [0] * ncov
    epochs = collections.defaultdict(list)
    is_sorted = True
    for l in open(path):
        if l.startswith("#sorted"):
            is_sorted = l.split()[1] == "1"
            continue
        e, p, d = l.split()
        epochs[int(e)].append((int(p), int(d)))
    for ms in epochs.values():
        assert all(a[0] <= b[0] for a, b in zip(ms, ms[1:])), "marks of an epoch must be sorted by position"
        depth = 0
        for i, (p, d) in enumerate(ms):
            depth += d
            run = ms[i + 1][0] - p if i + 1 < len(ms) else 0
            if depth and run > 0:
                cov[0 if depth < cmin else ncov - 1 if depth > cmax else 1 + (depth - cmin) // cstep] += run
        assert depth == 0
    if not is_sorted:
        return ""
    out = [HEAD]
    if cov[0]:
        out.append("COV\t[<%d]\t%d\t%d\n" % (cmin, cmin - 1, cov[0]))
    for i in range(1, ncov - 1):
        if cov[i]:
            out.append("COV\t[%d-%d]\t%d\t%d\n" % (cmin + (i - 1) * cstep, cmin + i * cstep - 1, cmin + i * cstep - 1, cov[i]))
    if cov[ncov - 1]:
        out.append("COV\t[%d<]\t%d\t%d\n" % (cmin + (ncov - 2) * cstep - 1, cmin + (ncov - 2) * cstep - 1, cov[ncov - 1]))
    return "".join(out)


def oracle_section(oracle_bin, args):
    p = subprocess.run([oracle_bin, "stats"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def engine_marks_section(product_bin, args, tmp_path, c=(1, 1000, 1)):
    m = str(tmp_path / "marks.txt")
    p = subprocess.run([product_bin, "stats", "--marks-out", m] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, (runs_to_section(m, c) if p.returncode == 0 else ""), p.stderr.decode()


def case_args(case):
    exp, opts, inp = case[:3]
    return exp, opts.replace("{G}", G).split() + [os.path.join(G, inp)] + (case[3].split() if len(case) > 3 else [])


@pytest.mark.parametrize("case", regcases.STATS_COV, ids=["%s-%d" % (c[0], i) for i, c in enumerate(regcases.STATS_COV)])
def test_reference_goldens(oracle_bin, product_bin, tmp_path, case):
    exp, args = case_args(case)
    want = open(os.path.join(G, exp + ".cov")).read()
    rc, out, err = oracle_section(oracle_bin, args)
    assert rc == 0 and out == want, err
    rc, out, err = engine_marks_section(product_bin, args, tmp_path)
    assert rc == 0 and out == want, err


def fuzz_sam(seed, path):
    """reads that make the ring's bookkeeping matter: ref skips and deletions longer than the buffer (fold), starts a whole buffer
    apart and contig changes (stale slot), read lengths that keep growing past 300 with depth pending (grow; deep piles make the
    low-bytes rule of the boundary element visible)"""
    rnd = random.Random(seed)
    contigs = [("c%d" % i, 3000000) for i in range(rnd.randint(1, 3))]
    lines = ["@HD\tVN:1.6\tSO:coordinate"] + ["@SQ\tSN:%s\tLN:%d" % c for c in contigs]
    mode = rnd.choice(["short", "grow", "splice", "mixed", "deep"])
    for name, ln in contigs:
        pos = rnd.randint(1, 50)
        maxlen = rnd.choice([40, 150, 299, 300, 301, 700])
        for k in range(rnd.randint(1, 120 if mode != "deep" else 700)):
            if mode in ("grow", "mixed") and rnd.random() < 0.15:
                maxlen = int(maxlen * rnd.choice([1.0, 1.3, 2.1, 3])) + rnd.randint(0, 3)
            L = rnd.randint(max(5, maxlen // 2), maxlen) if mode != "deep" else rnd.choice([50, 100])
            if mode == "deep" and rnd.random() < 0.01:
                L = rnd.choice([300, 301, 450, 900, 1300, 2000, 2600, 5300])
            ops, left = [], L
            if rnd.random() < 0.2:
                ops.append((rnd.randint(1, 30), "H"))
            if rnd.random() < 0.2 and left > 4:
                sc = rnd.randint(1, left // 3); ops.append((sc, "S")); left -= sc
            while left > 0:
                m = rnd.randint(1, left); ops.append((m, rnd.choice("MM=X"))); left -= m
                if left > 0:
                    t = rnd.random()
                    if t < 0.3:
                        i = rnd.randint(1, min(left, 5)); ops.append((i, "I")); left -= i
                    elif t < 0.6:
                        ops.append((rnd.randint(1, 30), "D"))
                    elif mode in ("splice", "mixed") and t < 0.95:
                        ops.append((rnd.choice([100, 1400, 1499, 1500, 1501, 2999, 3000, 3001, 7000, rnd.randint(1, 9000)]), "N"))
                    elif t < 0.8:
                        ops.append((rnd.randint(1000, 4000), "D"))
            flag = rnd.choice([0, 16, 99, 147, 0, 0, 1024, 256, 2048, 4, 512])
            lines.append("r%d\t%d\t%s\t%d\t30\t%s\t*\t0\t0\t%s\t*" % (k, flag, name, pos, "".join("%d%s" % o for o in ops), "A" * L))
            g = rnd.random()
            if mode == "deep":
                pos += rnd.choice([0, 0, 0, 1])
            elif g < 0.6:
                pos += rnd.randint(0, 40)
            elif g < 0.9:
                pos += rnd.randint(0, 600)
            else:
                pos += rnd.choice([1400, 1499, 1500, 1501, 3000, 5 * maxlen - 1, 5 * maxlen, 5 * maxlen + 1, rnd.randint(1500, 20000)])
            if pos > ln - 20000:
                break
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def plain_depth_section(sam):
    """the histogram of plain per-position depth over M/=/X blocks -- what the section would be without the ring's quirks"""
    import re
    dep = collections.Counter()
    for l in open(sam):
        if l.startswith("@"):
            continue
        f = l.split("\t")
        if int(f[1]) & (256 | 4):
            continue
        p = int(f[3]) - 1
        for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5]):
            n = int(n)
            if op in "M=X":
                for x in range(p, p + n):
                    dep[(f[2], x)] += 1
            if op in "MDN=X":
                p += n
    cov = collections.Counter(dep.values())
    return HEAD + "".join("COV\t[%d-%d]\t%d\t%d\n" % (d, d, d, c) for d, c in sorted(cov.items()))


def test_ring_bookkeeping_against_the_restated_ring(oracle_bin, product_bin, tmp_path):
    sam = str(tmp_path / "f.sam")
    quirky = 0
    for seed in range(60):
        fuzz_sam(seed, sam)
        rc, want, err = oracle_section(oracle_bin, [sam])
        rc2, got, err2 = engine_marks_section(product_bin, [sam], tmp_path)
        assert rc == rc2 == 0, (seed, err, err2)
        assert got == want, seed
        if seed < 12:
            quirky += want != plain_depth_section(sam)
    assert quirky >= 6       # the corpus does exercise the quirks: the output is NOT the plain depth histogram


def test_target_regions_against_the_restated_ring(oracle_bin, product_bin, tmp_path):
    """-t with generated target files (overlapping, nested, tiny and contig-wide intervals, names the header does not know) and the
    same regions as arguments, on the generated inputs above: is_in_regions + the chunk-clipped walk in front of the ring's quirks"""
    sam = str(tmp_path / "f.sam")
    differs = 0
    for seed in range(200, 230):
        fuzz_sam(seed, sam)
        rnd = random.Random(seed)
        n_contigs = sum(1 for l in open(sam) if l.startswith("@SQ"))
        ivals, lines = [], ["# targets", "nosuch\t5 90"]
        for c in range(n_contigs):
            if rnd.random() < 0.2:
                continue
            start = rnd.randint(1, 400)
            for _ in range(rnd.randint(1, 25)):
                ln = rnd.choice([1, 7, 60, 300, 1499, 1500, 4000, 100000])
                ivals.append(("c%d" % c, start, start + ln))
                lines.append("c%d\t%d %d" % (c, start, start + ln))
                start += rnd.choice([0, 1, 5, 200, 1500, 9000]) + (ln if rnd.random() < 0.7 else ln // 2)
        if not ivals:
            continue
        tf = str(tmp_path / "t.txt")
        open(tf, "w").write("\n".join(lines) + "\n")
        rc, want, err = oracle_section(oracle_bin, ["-t", tf, sam])
        rc2, got, err2 = engine_marks_section(product_bin, ["-t", tf, sam], tmp_path)
        assert rc == rc2 == 0, (seed, err, err2)
        assert got == want, seed
        regs = ["%s:%d-%d" % iv for iv in ivals]
        rc, want_r, err = oracle_section(oracle_bin, [sam] + regs)
        rc2, got_r, err2 = engine_marks_section(product_bin, [sam] + regs, tmp_path)
        assert rc == rc2 == 0 and got_r == want_r == want, (seed, err, err2)
        rc, whole, _ = oracle_section(oracle_bin, [sam])
        differs += whole != want
    assert differs >= 10


def paired_sam(seed, path, n_templates, far_mates=0):
    """templates whose mates overlap in every way (contained, staggered, touching, apart), spliced mates, a third line of a template
    (supplementary, same read number as the first), tlen both sides of the 2 x length rule; far_mates: that many templates whose
    second mate lies far downstream, so that the pair table fills up and its clean-up schedule runs"""
    rnd = random.Random(seed)
    recs = []
    pos = 100
    for t in range(n_templates):
        L = rnd.choice([50, 100, 150])
        far = t < far_mates
        off = rnd.choice([0, 1, L // 3, L - 1, L, L + 5, 3 * L]) if not far else 40000 + rnd.randint(0, 500)
        cig1 = rnd.choice(["%dM" % L, "%dM5D%dM" % (L // 2, L - L // 2), "%dM300N%dM" % (L // 3, L - L // 3), "5S%dM" % (L - 5)])
        cig2 = rnd.choice(["%dM" % L, "%dM2I%dM" % (L // 2, L - L // 2 - 2), "%dM40N%dM" % (L // 4, L - L // 4)])
        tlen = rnd.choice([off + L, 2 * L - 1, 2 * L, 2 * L + 1, 0])
        f1, f2 = rnd.choice([(99, 147), (97, 145), (65, 129), (73, 133), (0, 16)])
        recs.append((pos, "t%d\t%d\tc0\t%d\t30\t%s\t=\t%d\t%d\t%s\t*" % (t, f1, pos, cig1, pos + off, tlen, "A" * L)))
        recs.append((pos + off, "t%d\t%d\tc0\t%d\t30\t%s\t=\t%d\t%d\t%s\t*" % (t, f2, pos + off, cig2, pos, -tlen, "A" * L)))
        if rnd.random() < 0.1:
            recs.append((pos + off + 3, "t%d\t%d\tc0\t%d\t30\t%dM\t=\t%d\t%d\t%s\t*" % (t, (f1 | 2048) & ~0, pos + off + 3, L, pos + off, tlen, "A" * L)))
        pos += rnd.choice([0, 1, 3, 20]) if far_mates else rnd.choice([0, 3, 40, 400])
    recs.sort(key=lambda r: r[0])
    with open(path, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c0\tLN:9000000\n" + "\n".join(r[1] for r in recs) + "\n")


def test_mate_overlap_removal_against_the_restated_ring(oracle_bin, product_bin, tmp_path):
    sam = str(tmp_path / "p.sam")
    changed = 0
    for seed in range(12):
        paired_sam(seed, sam, 400)
        rc, want, err = oracle_section(oracle_bin, ["-p", sam])
        rc2, got, err2 = engine_marks_section(product_bin, ["-p", sam], tmp_path)
        assert rc == rc2 == 0 and got == want, (seed, err, err2)
        changed += want != oracle_section(oracle_bin, [sam])[1]
    assert changed >= 8
    # more than 10 000 templates waiting for their mates: the table is thinned out on the reference's schedule
    paired_sam(99, sam, 14000, far_mates=12000)
    rc, want, err = oracle_section(oracle_bin, ["-p", sam])
    rc2, got, err2 = engine_marks_section(product_bin, ["-p", sam], tmp_path)
    assert rc == rc2 == 0 and got == want, (err, err2)


def test_options_and_unsorted_input(oracle_bin, product_bin, tmp_path):
    sam = str(tmp_path / "f.sam")
    fuzz_sam(1007, sam)
    for opts, c in ((["-c", "2,9,3"], (2, 9, 3)), (["-c", "1,5,100"], (1, 5, 100)), (["-d"], None), (["-F", "0x10"], None), (["-f", "3", "-F", "1024"], None),
                    (["-l", "150"], None), (["-c", "3,3,1"], (3, 3, 1))):
        rc, want, err = oracle_section(oracle_bin, opts + [sam])
        rc2, got, err2 = engine_marks_section(product_bin, opts + [sam], tmp_path, c or (1, 1000, 1))
        assert rc == rc2 == 0 and got == want, (opts, err, err2)
    # -I: read groups by ID or by sample (stats.c:2151-2177)
    bam = os.path.join(G, "11_target.bam")
    for rg in ("grp2", "Sample", "nosuch"):
        rc, want, _ = oracle_section(oracle_bin, ["-I", rg, bam])
        rc2, got, _ = engine_marks_section(product_bin, ["-I", rg, bam], tmp_path)
        assert rc == rc2 == 0 and got == want, rg
    # a position going backwards inside a contig: is_sorted drops and the section is not printed at all (stats.c:1381, :1884)
    lines = open(sam).read().splitlines()
    recs = [i for i, l in enumerate(lines) if not l.startswith("@")]
    lines[recs[3]], lines[recs[2]] = lines[recs[2]], lines[recs[3]]
    bad = str(tmp_path / "u.sam")
    open(bad, "w").write("\n".join(lines) + "\n")
    f2, f3 = lines[recs[2]].split("\t"), lines[recs[3]].split("\t")
    if f2[2] == f3[2] and int(f2[3]) > int(f3[3]):
        rc, want, _ = oracle_section(oracle_bin, [bad])
        rc2, got, _ = engine_marks_section(product_bin, [bad], tmp_path)
        assert rc == rc2 == 0 and want == "" and got == ""
    # refused: what the section would need regions for
    for opts in (["-S", "RG"],):
        assert subprocess.run([product_bin, "stats"] + opts + [sam], stderr=subprocess.PIPE).returncode == 1


@pytest.mark.gpu
def test_device_bins_equal_the_oracle(oracle_bin, product_bin, tmp_path):
    def both(args, env=None):
        a = subprocess.run([oracle_bin, "stats"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([product_bin, "stats"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert a.returncode == b.returncode == 0, (args, a.stderr[-200:], b.stderr[-200:])
        assert a.stdout == b.stdout, args
        return a.stdout
    for case in regcases.STATS_COV:
        exp, args = case_args(case)
        assert both(args).decode() == open(os.path.join(G, exp + ".cov")).read(), exp
    sam = str(tmp_path / "f.sam")
    for seed in range(100, 130):
        fuzz_sam(seed, sam)
        both([sam])
        if seed % 10 == 0:
            both(["-c", "2,40,7", "-d", sam])
    # a 30x whole-contig input: several device batches, the LDS histogram and (with 5 000 bins) the global one
    from synth import write_synth_sam
    from bamio import sam_to_bam
    big, _fa = write_synth_sam(str(tmp_path), n_ref=400000, depth=30, read_len=150, seed=9, paired=True, indel_rate=0.02)
    bam = sam_to_bam(big, big[:-4] + ".bam", level=1)
    out = both([bam])
    assert out.count(b"COV\t") > 20
    for b in ("2", "3", "4097", "50000"):          # marks per device call: runs that straddle batches, carried depth
        both([bam], dict(os.environ, STA_STATS_BATCH=b) if b != "2" else dict(os.environ, STA_STATS_BATCH="2000"))
    both(["-c", "1,5000,1", bam])
    both(["-c", "10,20,4", "-F", "0x400", bam])
