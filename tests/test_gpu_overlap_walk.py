"""Mate-overlap resolution (HTSlib tweak_overlap_quality; switched on at bam_plcmd.c:586) over pairs whose CIGARs are
anything but plain: the device resolves a pair with the whole wave -- every lane steps the reference's two-cursor walk over the CIGARs alone and
hands out RUNS of per-base actions (kernels_overlap.hip resolve_pairs_wave / pair_event) -- so this input is built to reach every branch
of that walk: deletions and reference skips of either mate inside the overlap (the `*(cig - 1) == D` quirks included), insertions, pads,
soft and hard clips, =/X runs, mates starting on the same column, mates that do not overlap at all, equal qualities (the keeper of a tie comes from the name's hash) and qualities that sum past 200.
Engine text against the oracle's, through both pairing lanes: partners staged by the input lane (k_olap_pairs) and the window's own
name table (k_name_groups, STA_OLAP_DEVICE_TABLE=1: what bench.py and callers without a name hash use)."""
import os, random, subprocess
import pytest

pytestmark = pytest.mark.gpu

QUALS = [0, 2, 13, 20, 20, 30, 30, 40, 41, 60, 93, 93, 120]


def _cigar(rnd, plain_p):
    """[(op, len)] with an M-type operation first and last among the reference/query consuming ones."""
    if rnd.random() < plain_p:
        body = [("M", rnd.randint(20, 90))]
    else:
        body = [(rnd.choice("M=X"), rnd.randint(1, 40))]
        for _ in range(rnd.randint(1, 5)):
            body.append((rnd.choice("IDDNP"), rnd.randint(1, 12)))
            if rnd.random() < 0.15:
                body.append((rnd.choice("IDN"), rnd.randint(1, 4)))       # two gap operations in a row
            body.append((rnd.choice("MMM=X"), rnd.randint(1, 40)))
    head, tail = [], []
    if rnd.random() < 0.15: head.append(("H", rnd.randint(1, 9)))
    if rnd.random() < 0.3: head.append(("S", rnd.randint(1, 15)))
    if rnd.random() < 0.3: tail.append(("S", rnd.randint(1, 15)))
    if rnd.random() < 0.15: tail.append(("H", rnd.randint(1, 9)))
    return head + body + tail


def _record(rnd, ref, name, flag, pos, cig, mpos, tlen, bare):
    seq, x = [], pos
    for op, n in cig:
        if op in "M=X":
            for k in range(n):
                c = ref[x + k]
                seq.append(rnd.choice("ACGT") if rnd.random() < 0.1 else c)
            x += n
        elif op in "IS":
            seq.extend(rnd.choice("ACGTN") for _ in range(n))
        elif op in "DN":
            x += n
    qual = "".join(chr(33 + min(93, rnd.choice(QUALS))) for _ in seq)
    s = "".join(seq)
    if bare: s, qual = "*", "*"
    return "\t".join([name, str(flag), "c1", str(pos + 1), "50", "".join("%d%s" % (n, op) for op, n in cig), "=", str(mpos + 1), str(tlen), s, qual]) + "\n"


def _ref_span(cig):
    return sum(n for op, n in cig if op in "M=XDN")


def write_overlap_sam(d, seed, n_pairs=900, n_ref=5000, plain_p=0.25):
    rnd = random.Random(seed)
    ref = "".join(rnd.choice("ACGT") for _ in range(n_ref))
    fa = os.path.join(d, "o.fa")
    with open(fa, "w") as f:
        f.write(">c1\n")
        for i in range(0, n_ref, 60): f.write(ref[i:i + 60] + "\n")
    recs = []
    for k in range(n_pairs):
        ca, cb = _cigar(rnd, plain_p), _cigar(rnd, plain_p)
        sa = _ref_span(ca)
        pa = rnd.randint(0, n_ref - 700)
        u = rnd.random()
        pb = pa if u < 0.05 else pa + sa + rnd.randint(0, 30) if u < 0.12 else pa + rnd.randint(0, max(0, sa - 1))
        name = "q" * rnd.randint(1, 3) + "%d" % k + rnd.choice(["", "/x", ":ab", "_"])
        # (no mate without SEQ here: the reference reads qual[] of such a record out of bounds and then fails the whole run -- tweak_overlap_quality
        #  returns -1 at `iseq > l_qseq` and bam_plp_push gives up; DESIGN.md section 2 lists it with the other SEQ-less departures)
        recs.append((pa, 2 * k, _record(rnd, ref, name, 99, pa, ca, pb, 300, False)))
        recs.append((pb, 2 * k + 1, _record(rnd, ref, name, 147, pb, cb, pa, -300, False)))
    recs.sort(key=lambda t: (t[0], t[1]))
    sam = os.path.join(d, "o.sam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:%d\n" % n_ref)
        f.writelines(t[2] for t in recs)
    return sam, fa


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_overlap_walk_of_rich_cigars(tmp_path, oracle_bin, product_bin, seed):
    sam, fa = write_overlap_sam(str(tmp_path), seed)
    for args in (["mpileup", "-B", "-Q", "0", "-f", fa], ["mpileup", "-f", fa], ["mpileup", "-B", "-Q", "0", "-x", "-f", fa]):
        want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        assert want.count(b"\n") > 3000
        for env in ({}, {"STA_OLAP_DEVICE_TABLE": "1"}, {"STA_OLAP_DEVICE_TABLE": "1", "STA_IO_LANE": "rec"}):
            got = subprocess.run([product_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert got.returncode == 0, got.stderr.decode()[-500:]
            assert got.stdout == want, (seed, args, env)
    # the resolution changes the text (else the test proves nothing)
    a = subprocess.run([oracle_bin, "mpileup", "-B", "-Q", "0", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    b = subprocess.run([oracle_bin, "mpileup", "-B", "-Q", "0", "-x", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert a != b


def test_mate_without_seq_is_left_alone(tmp_path, oracle_bin, product_bin):
    """A properly paired mate without SEQ that overlaps its partner: the reference's walk reads qual[] of the empty record and then fails the whole
    run (tweak_overlap_quality returns -1 at `iseq > l_qseq`, bam_plp_push gives up: "error reading from input file").  The engine's walk stops
    at the bound check without touching anything: the text is that of the same input without overlap detection."""
    rnd = random.Random(5)
    ref = "".join(rnd.choice("ACGT") for _ in range(400))
    fa = str(tmp_path / "o.fa")
    open(fa, "w").write(">c1\n" + ref + "\n")
    sam = str(tmp_path / "o.sam")
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:400\n")
        f.write(_record(rnd, ref, "p1", 99, 10, [("M", 60)], 30, 80, False))
        f.write(_record(rnd, ref, "p2", 99, 20, [("M", 30), ("D", 3), ("M", 30)], 40, 80, True))
        f.write(_record(rnd, ref, "p1", 147, 30, [("M", 20), ("I", 2), ("M", 38)], 10, -80, True))
        f.write(_record(rnd, ref, "p2", 147, 40, [("M", 60)], 20, -80, False))
    bad = subprocess.run([oracle_bin, "mpileup", "-B", "-Q", "0", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode != 0 and b"error reading from input file" in bad.stderr
    want = subprocess.run([oracle_bin, "mpileup", "-B", "-Q", "0", "-x", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    for env in ({}, {"STA_OLAP_DEVICE_TABLE": "1"}):
        got = subprocess.run([product_bin, "mpileup", "-B", "-Q", "0", "-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert got.returncode == 0, got.stderr.decode()[-500:]
        assert got.stdout == want, env
