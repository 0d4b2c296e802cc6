"""A deliberately messy, seeded SAM generator for parity hunts (engine vs oracle): several contigs, clipped / spliced /
padded CIGARs, every flag the read filters look at, missing SEQ / QUAL, names with more than two records.
Not product code.  `write_rich_sam(outdir, seed, n_templates)` -> (sam_path, fasta_path)."""
import os
import random

BASES = "ACGT"


def _cigar_for(rng, L):
    """Random CIGAR consuming exactly L query bases (hard clips and pads consume none); returns (cigar string, ref span)."""
    r = rng.random()
    ops = []
    left = L
    if r < 0.55 or L < 30:
        ops = [(L, "M")]
    else:
        if rng.random() < 0.25:
            ops.append((rng.randint(1, 5), "H"))
        if rng.random() < 0.35:
            s = rng.randint(1, min(12, left - 20)); ops.append((s, "S")); left -= s
        tail = rng.randint(1, min(10, left - 15)) if rng.random() < 0.3 else 0
        left -= tail
        n_ev = rng.randint(1, 3)
        while n_ev and left > 12:
            m = rng.randint(4, left - 6)
            ops.append((m, rng.choice("M==X") if rng.random() < 0.15 else "M")); left -= m
            ev = rng.random()
            if ev < 0.35:
                k = rng.randint(1, min(9, left - 3)); ops.append((k, "I")); left -= k
                if rng.random() < 0.15: ops.append((rng.randint(1, 3), "P"))
                if rng.random() < 0.2: ops.append((rng.randint(1, 4), "D"))
            elif ev < 0.7:
                ops.append((rng.randint(1, 12), "D"))
                if rng.random() < 0.1: ops.append((rng.randint(1, 3), "I")) if left > 6 else None
                if ops[-1][1] == "I": left -= ops[-1][0]
            elif ev < 0.9:
                ops.append((rng.randint(5, 200), "N"))
            else:
                ops.append((rng.randint(1, 3), "P"))
            n_ev -= 1
        ops.append((left, "M"))
        if tail: ops.append((tail, "S"))
        if rng.random() < 0.2: ops.append((rng.randint(1, 5), "H"))
    # merge nothing, just compute
    span = sum(n for n, o in ops if o in "MDN=X")
    qlen = sum(n for n, o in ops if o in "MIS=X")
    assert qlen == L, (ops, L, qlen)
    return "".join("%d%s" % x for x in ops), span, ops


def write_rich_sam(outdir, seed=1, n_templates=4000, contigs=(("c1", 30000), ("c2", 9000), ("c3", 45000))):
    rng = random.Random(seed)
    refs = {n: "".join(rng.choice(BASES) for _ in range(l)) for n, l in contigs}
    tid_of = {n: i for i, (n, _) in enumerate(contigs)}
    recs = []      # (tid, pos0, order, line fields)
    order = 0

    def seq_for(cname, pos0, ops):
        out = []; x = pos0
        for n, o in ops:
            if o in "M=X":
                seg = refs[cname][x:x + n]
                seg = "".join(c if (rng.random() > 0.02 and o != "X") else rng.choice(BASES) for c in seg)
                out.append(seg + "N" * (n - len(seg))); x += n
            elif o in "IS": out.append("".join(rng.choice(BASES + ("N" if rng.random() < 0.05 else "A")) for _ in range(n)))
            elif o in "DN": x += n
        return "".join(out)

    def qual_for(L):
        mode = rng.random()
        if mode < 0.05: return "*"
        hi = rng.choice((41, 41, 60, 93))
        return "".join(chr(33 + (rng.randint(0, hi) if rng.random() < 0.3 else rng.choice((2, 11, 25, 37)))) for _ in range(L))

    def place(cname, L):
        clen = len(refs[cname])
        cig, span, ops = _cigar_for(rng, L)
        pos0 = rng.randint(0, max(0, clen - span - 1)) if span < clen else 0
        return cig, span, ops, pos0

    for t in range(n_templates):
        name = "t%d" % t
        cname = rng.choice([c for c, _ in contigs])
        mapq = rng.choice((0, 3, 20, 30, 60, 60, 60, 255))
        kind = rng.random()
        extra = 0
        for bit, p in ((256, 0.02), (512, 0.02), (1024, 0.03), (2048, 0.02)):
            if rng.random() < p: extra |= bit
        if kind < 0.6:
            # pair on the same contig (proper or not), mates within ~400 bp
            L1, L2 = rng.randint(40, 150), rng.randint(40, 150)
            cig1, span1, ops1, p1 = place(cname, L1)
            cig2, span2, ops2, _ = place(cname, L2)
            p2 = min(max(0, p1 + rng.randint(-30, 350)), max(0, len(refs[cname]) - span2 - 1))
            proper = rng.random() < 0.85
            f1 = 1 | (2 if proper else 0) | 64 | (32 if rng.random() < 0.5 else 0)
            f2 = 1 | (2 if proper else 0) | 128 | (16 if f1 & 32 else 0)
            if rng.random() < 0.5: f1 |= 16; f2 |= 32
            lo, hi = min(p1, p2), max(p1 + span1, p2 + span2)
            isz = hi - lo
            for (f, p, cig, ops, L, mp, sign) in ((f1, p1, cig1, ops1, L1, p2, 1 if p1 <= p2 else -1), (f2, p2, cig2, ops2, L2, p1, 1 if p2 < p1 else -1)):
                s = seq_for(cname, p, ops)       # (no SEQ '*' inside pairs: mate-overlap resolution walks the bases)
                q = qual_for(L)
                recs.append((tid_of[cname], p, order, [name, str(f | extra), cname, str(p + 1), str(mapq), cig, "=", str(mp + 1), str(sign * isz), s, q])); order += 1
            if rng.random() < 0.03:      # a third record with the same name (supplementary-like)
                L3 = rng.randint(40, 100); cig3, span3, ops3, p3 = place(cname, L3)
                recs.append((tid_of[cname], p3, order, [name, str(1 | 2 | 64 | 2048), cname, str(p3 + 1), str(mapq), cig3, "=", str(p2 + 1), "0", seq_for(cname, p3, ops3), qual_for(L3)])); order += 1
        elif kind < 0.7:
            # mate on another contig / unmapped mate
            L1 = rng.randint(40, 150); cig1, span1, ops1, p1 = place(cname, L1)
            if rng.random() < 0.5:
                other = rng.choice([c for c, _ in contigs if c != cname]); mp = rng.randint(0, len(refs[other]) - 200)
                f = 1 | 64 | (2 if rng.random() < 0.3 else 0) | (16 if rng.random() < 0.5 else 0)
                recs.append((tid_of[cname], p1, order, [name, str(f | extra), cname, str(p1 + 1), str(mapq), cig1, other, str(mp + 1), "0", seq_for(cname, p1, ops1), qual_for(L1)])); order += 1
            else:
                f = 1 | 8 | 64 | (16 if rng.random() < 0.5 else 0)
                recs.append((tid_of[cname], p1, order, [name, str(f | extra), cname, str(p1 + 1), str(mapq), cig1, "=", str(p1 + 1), "0", seq_for(cname, p1, ops1), qual_for(L1)])); order += 1
                # the unmapped mate, placed at the same position
                L2 = rng.randint(30, 100)
                recs.append((tid_of[cname], p1, order, [name, str(1 | 4 | 128), cname, str(p1 + 1), "0", "*", "=", str(p1 + 1), "0", "".join(rng.choice(BASES) for _ in range(L2)), qual_for(L2)])); order += 1
        else:
            L1 = rng.randint(30, 150); cig1, span1, ops1, p1 = place(cname, L1)
            f = 16 if rng.random() < 0.5 else 0
            s1 = seq_for(cname, p1, ops1) if rng.random() > 0.04 else "*"
            recs.append((tid_of[cname], p1, order, [name, str(f | extra), cname, str(p1 + 1), str(mapq), cig1, "*", "0", "0", s1, qual_for(L1) if s1 != "*" else "*"])); order += 1
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    sam = os.path.join(outdir, "rich_%d.sam" % seed)
    fa = os.path.join(outdir, "rich_%d.fa" % seed)
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for n, l in contigs: fh.write("@SQ\tSN:%s\tLN:%d\n" % (n, l))
        fh.write("@RG\tID:g1\tSM:s1\n@RG\tID:g2\tSM:s2\n")
        for _, _, _, f in recs:
            fh.write("\t".join(f) + "\tRG:Z:%s\tNM:i:%d\n" % (rng.choice(("g1", "g2")), rng.randint(0, 5)))
        for k in range(5):       # unplaced, unmapped reads at the end
            fh.write("u%d\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tFFFF\n" % k)
    with open(fa, "w") as fh:
        for n, _ in contigs:
            fh.write(">%s\n" % n)
            s = refs[n]
            for i in range(0, len(s), 70): fh.write(s[i:i + 70] + "\n")
    return sam, fa
