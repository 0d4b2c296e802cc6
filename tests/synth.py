"""Deterministic synthetic aligned reads: used by the tests, smoke() and bench.py.

One generator feeds both sides of every parity test: `synth_reads` returns the structure-of-arrays
staging layout of include/samtools_amd.h (numpy), `write_sam` writes the very same reads as SAM
text for the CLIs (engine and oracle).  Not product code."""
import os

import numpy as np

NT16 = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15}
CODE2CHR = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
QUAL_SET = np.array([2, 11, 25, 37], dtype=np.uint8)
QUAL_P = np.array([0.02, 0.05, 0.13, 0.80])


def synth_ref(n, seed=1):
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]


def synth_reads(ref, depth=30, read_len=150, seed=42, paired=False, sub_rate=0.001, indel_rate=0.005,
                mapq=60, origin=0, max_indel=3, start_span=None, n_reads=None, trim_rate=0.0, trim_max=50):
    """Returns a dict with the sta_reads arrays for ONE file covering the whole of `ref`.

    ref: uint8 array of ASCII bases (contig of length len(ref)); reads lie fully inside it.
    start_span / n_reads (unpaired only): draw that many start positions from [0, start_span) instead of depth * len(ref) / L
    of them from the whole of `ref` -- synth_chunked() builds long inputs out of such pieces.
    trim_rate / trim_max: that fraction of the reads without an indel lose 1..trim_max bases at their right-hand end (adapter / quality
    trimming: many read lengths in one input; nothing is drawn when trim_rate is 0, so the other inputs keep their seeds)."""
    rng = np.random.default_rng(seed)
    n_ref = len(ref)
    L = read_len
    if n_reads is None:
        n_reads = max(1, int(depth * n_ref / L))
    if paired:
        n_pairs = max(1, n_reads // 2)
        isz = np.maximum(np.rint(rng.normal(300, 30, n_pairs)).astype(np.int64), L)
        isz = np.minimum(isz, n_ref)
        s = rng.integers(0, np.maximum(n_ref - isz + 1, 1))
        left = s
        right = s + isz - L
        orient = rng.integers(0, 2, n_pairs)          # 0: read1 left/fwd (99/147), 1: read1 right/rev (83/163)
        pos = np.concatenate([left, right])
        flag = np.concatenate([np.where(orient == 0, 99, 163), np.where(orient == 0, 147, 83)]).astype(np.uint16)
        mpos = np.concatenate([right, left])
        tlen = np.concatenate([isz, -isz])
        pair_id = np.concatenate([np.arange(n_pairs), np.arange(n_pairs)])
        n_reads = 2 * n_pairs
    else:
        pos = rng.integers(0, n_ref - L + 1 if start_span is None else start_span, n_reads)
        flag = np.where(rng.integers(0, 2, n_reads) == 1, 16, 0).astype(np.uint16)
        mpos = np.full(n_reads, -1, dtype=np.int64)
        tlen = np.zeros(n_reads, dtype=np.int64)
        pair_id = np.arange(n_reads)
    order = np.argsort(pos, kind="stable")
    pos, flag, mpos, tlen, pair_id = pos[order], flag[order], mpos[order], tlen[order], pair_id[order]

    # bases: reference + substitutions
    idx = pos[:, None] + np.arange(L)[None, :]
    bases = ref[idx].copy()                                  # ASCII
    sub = rng.random(bases.shape) < sub_rate
    if sub.any():
        alt = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(sub.sum()))]
        bases[sub] = alt
    # == QUAL_SET[rng.choice(4, size=bases.shape, p=QUAL_P)] bit for bit (Generator.choice draws one uniform per element and
    # searches the cumulative distribution), five times faster at bench sizes
    cdf = QUAL_P.cumsum(); cdf /= cdf[-1]
    u = rng.random(bases.shape)
    qi = (u >= cdf[0]).astype(np.uint8); qi += u >= cdf[1]; qi += u >= cdf[2]
    quals = QUAL_SET[qi]
    del u, qi

    # CIGARs: default <L>M; a few reads carry one 1-3 bp insertion or deletion
    cig_n = np.ones(n_reads, dtype=np.int64)
    has_indel = rng.random(n_reads) < indel_rate
    cigars = {}
    for r in np.nonzero(has_indel)[0]:
        k = int(rng.integers(1, max_indel + 1))
        at = int(rng.integers(10, L - 10 - k))
        if rng.integers(0, 2) == 0 or pos[r] + L + k > n_ref:
            # insertion: query keeps L bases, k of them inserted -> reference span L-k
            cigars[int(r)] = [(at, 0), (k, 1), (L - at - k, 0)]
            bases[r, at + k:] = ref[pos[r] + at: pos[r] + L - k]
            bases[r, at:at + k] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, k)]
        else:
            cigars[int(r)] = [(at, 0), (k, 2), (L - at, 0)]
            bases[r, at:] = ref[pos[r] + at + k: pos[r] + L + k]
        cig_n[r] = 3
    cig_off = np.zeros(n_reads + 1, dtype=np.uint32)
    np.cumsum(cig_n, out=cig_off[1:])
    cigar = np.full(int(cig_off[-1]), (L << 4) | 0, dtype=np.uint32)
    for r, ops in cigars.items():
        o = int(cig_off[r])
        for j, (ln, op) in enumerate(ops):
            cigar[o + j] = (ln << 4) | op

    l_qseq = np.full(n_reads, L, dtype=np.int32)
    if trim_rate > 0:
        cut = (rng.random(n_reads) < trim_rate) & ~has_indel
        l_qseq[cut] = L - rng.integers(1, trim_max + 1, int(cut.sum()))
        for r in np.nonzero(cut)[0]:
            cigar[int(cig_off[r])] = (int(l_qseq[r]) << 4) | 0
        beyond = np.arange(L)[None, :] >= l_qseq[:, None]
        quals = quals.copy(); quals[beyond] = 0
        bases = bases.copy(); bases[beyond] = ord("N")

    Lp = (L + 7) & ~7
    qual_pool = np.zeros((n_reads, Lp), dtype=np.uint8)
    qual_pool[:, :L] = quals
    code = np.zeros(256, dtype=np.uint8) + 15
    for ch, v in NT16.items():
        code[ord(ch)] = v
    codes = np.zeros((n_reads, Lp), dtype=np.uint8)
    codes[:, :L] = code[bases]
    if trim_rate > 0:
        codes[:, :L][np.arange(L)[None, :] >= l_qseq[:, None]] = 0        # the pools are zero beyond a read's last base
    seq_pool = ((codes[:, 0::2] << 4) | codes[:, 1::2]).astype(np.uint8)
    base_off8 = (np.arange(n_reads, dtype=np.uint64) * (Lp >> 3)).astype(np.uint32)

    names = [("r%d" % p).encode() + b"\0" for p in pair_id]
    name_len = np.fromiter((len(x) for x in names), dtype=np.int64, count=n_reads)
    name_off = np.zeros(n_reads + 1, dtype=np.uint32)
    np.cumsum(name_len, out=name_off[1:])
    names_pool = np.frombuffer(b"".join(names), dtype=np.uint8)

    return {
        "n": n_reads, "L": L,
        "pos": (pos - origin).astype(np.int32), "flag": flag, "mapq": np.full(n_reads, mapq, dtype=np.uint8),
        "aux": np.zeros(n_reads, dtype=np.uint8), "l_qseq": l_qseq,
        "cig_off": cig_off, "base_off8": base_off8,
        "mtid": np.where(mpos >= 0, 0, -1).astype(np.int32) if paired else np.full(n_reads, -1, dtype=np.int32),
        "mpos": mpos.astype(np.int64), "isize": tlen.astype(np.int32), "name_off": name_off,
        "cigar": cigar, "seq": np.ascontiguousarray(seq_pool).reshape(-1), "qual": qual_pool.reshape(-1),
        "names": names_pool,
        "_bases": bases, "_quals": quals, "_abs_pos": pos,
    }


def cigar_str(rd, r):
    ops = rd["cigar"][int(rd["cig_off"][r]):int(rd["cig_off"][r + 1])]
    return "".join("%d%s" % (int(c) >> 4, "MIDNSHP=XB"[int(c) & 15]) for c in ops)


def write_sam(path, rd, ref_name, ref_len):
    with open(path, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (ref_name, ref_len))
        names = rd["names"].tobytes().split(b"\0")
        for r in range(rd["n"]):
            mp = int(rd["mpos"][r])
            fh.write("%s\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d\t%s\t%s\n" % (
                names[r].decode(), int(rd["flag"][r]), ref_name, int(rd["_abs_pos"][r]) + 1, int(rd["mapq"][r]),
                cigar_str(rd, r), "=" if mp >= 0 else "*", mp + 1, int(rd["isize"][r]),
                rd["_bases"][r][:int(rd["l_qseq"][r])].tobytes().decode(), (rd["_quals"][r][:int(rd["l_qseq"][r])] + 33).astype(np.uint8).tobytes().decode()))


def write_fasta(path, name, ref):
    with open(path, "w") as fh:
        fh.write(">%s\n" % name)
        s = ref.tobytes().decode()
        for i in range(0, len(s), 60):
            fh.write(s[i:i + 60] + "\n")


def write_synth_sam(outdir, n_ref=20000, depth=20, read_len=100, seed=7, paired=True, name="chrS", **kw):
    ref = synth_ref(n_ref, seed=1)
    rd = synth_reads(ref, depth=depth, read_len=read_len, seed=seed, paired=paired, **kw)
    sam = os.path.join(outdir, "synth.sam")
    fa = os.path.join(outdir, "synth.fa")
    write_sam(sam, rd, name, n_ref)
    write_fasta(fa, name, ref)
    return sam, fa


_CONCAT = ("flag", "mapq", "aux", "l_qseq", "mtid", "mpos", "isize", "cigar", "seq", "qual", "names", "_bases", "_quals")


def _piece_job(job):
    c0, sub, kw = job
    rd = synth_reads(sub, **kw)
    rd["_abs_pos"] = rd["_abs_pos"] + c0
    return rd


def synth_chunked(ref, chunk_cols, depth=30, read_len=150, seed=42, chunks=None, procs=1, **kw):
    """ONE long sorted input assembled from pieces: piece k holds the reads STARTING in columns [k * chunk_cols, (k + 1) * chunk_cols)
    (seed + k; they extend into the next piece's columns, so block cuts do split reads), generated independently so that a rank of
    a sharded run only has to build the pieces around its block.  chunks = iterable of piece indices (None: all).  With one piece
    covering the whole of `ref` the result is synth_reads(ref, seed=seed) itself.  procs > 1: the pieces are built by that many threads
    (same result).  Returns the merged dict (positions absolute)."""
    n = len(ref)
    L = read_len
    n_chunks = (n + chunk_cols - 1) // chunk_cols
    jobs = []
    for k in (range(n_chunks) if chunks is None else sorted(c for c in chunks if 0 <= c < n_chunks)):
        c0 = k * chunk_cols
        span = min(chunk_cols, n - L + 1 - c0)
        if span <= 0:
            continue
        sub = ref[c0:min(n, c0 + span + L + 16)]
        jobs.append((c0, sub, dict(depth=depth, read_len=L, seed=seed + k, start_span=span,
                                   n_reads=max(1, int(depth * (n if n_chunks == 1 else min(chunk_cols, n - c0)) / L)), **kw)))
    procs = min(int(procs or 1), len(jobs))
    if procs > 1:
        # the pieces are independent (own seed, own Generator): built by threads -- the work is large-array numpy calls, which release the
        # interpreter lock (no worker processes: the caller may hold a GPU context, and the pieces are hundreds of MB to hand back)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(procs) as pool:
            parts = list(pool.map(_piece_job, jobs))
    else:
        parts = [_piece_job(j) for j in jobs]
    if len(parts) == 1:
        out = parts[0]
        out["pos"] = out["_abs_pos"].astype(np.int32)
        return out
    out = {"n": sum(p["n"] for p in parts), "L": L}
    for f in _CONCAT:
        out[f] = np.concatenate([p[f] for p in parts])
    out["_abs_pos"] = np.concatenate([p["_abs_pos"] for p in parts])
    out["pos"] = out["_abs_pos"].astype(np.int32)
    for f, pool, unit in (("cig_off", "cigar", 1), ("name_off", "names", 1)):
        offs, base = [], 0
        for p in parts:
            offs.append(p[f][:-1].astype(np.int64) + base)
            base += len(p[pool])
        out[f] = np.concatenate(offs + [np.array([base], dtype=np.int64)]).astype(np.uint32)
    offs, base = [], 0
    for p in parts:
        offs.append(p["base_off8"].astype(np.int64) + base)
        base += len(p["qual"]) // 8
    out["base_off8"] = np.concatenate(offs).astype(np.uint32)
    return out


def _gather_pool(pool, off, perm):
    """variable-length per-read slices pool[off[r]:off[r+1]] re-ordered by perm -> (new pool, new offsets)"""
    ln = (off[1:].astype(np.int64) - off[:-1].astype(np.int64))[perm]
    new_off = np.zeros(len(perm) + 1, dtype=np.int64)
    np.cumsum(ln, out=new_off[1:])
    src = np.repeat(off[:-1].astype(np.int64)[perm], ln) + (np.arange(int(new_off[-1]), dtype=np.int64) - np.repeat(new_off[:-1], ln))
    return pool[src], new_off.astype(np.uint32)


def merge_reads(a, b):
    """two generated read sets over the same contig -> one position-sorted set (ties: a's reads first, each set's own order kept)"""
    n = a["n"] + b["n"]
    L = a["L"]
    Lp = (L + 7) & ~7
    pos = np.concatenate([a["_abs_pos"], b["_abs_pos"]])
    perm = np.argsort(pos, kind="stable")
    out = {"n": n, "L": L}
    for f in ("flag", "mapq", "aux", "l_qseq", "mtid", "mpos", "isize", "_bases", "_quals"):
        out[f] = np.concatenate([a[f], b[f]])[perm]
    out["_abs_pos"] = pos[perm]
    out["pos"] = out["_abs_pos"].astype(np.int32)
    out["qual"] = np.concatenate([a["qual"].reshape(-1, Lp), b["qual"].reshape(-1, Lp)])[perm].reshape(-1)
    out["seq"] = np.concatenate([a["seq"].reshape(-1, Lp // 2), b["seq"].reshape(-1, Lp // 2)])[perm].reshape(-1)
    out["base_off8"] = (np.arange(n, dtype=np.uint64) * (Lp >> 3)).astype(np.uint32)
    for off_f, pool_f in (("cig_off", "cigar"), ("name_off", "names")):
        pool = np.concatenate([a[pool_f], b[pool_f]])
        off = np.concatenate([a[off_f][:-1].astype(np.int64), b[off_f].astype(np.int64) + len(a[pool_f])])
        out[pool_f], out[off_f] = _gather_pool(pool, off, perm)
    return out


def synth_hotspot(ref, rd, hot_start, hot_len=300, hot_depth=10000, read_len=150, seed=1042):
    """`rd` plus a deep amplicon: hot_depth x coverage by reads lying inside [hot_start, hot_start + hot_len) (the
    "deep-amplicon shape" of BASELINE.json configs[3] inside an ordinary window).  Read names of the amplicon start with 'h'."""
    hot_start = max(0, min(int(hot_start), len(ref) - hot_len))
    sub = ref[hot_start:hot_start + hot_len]
    h = synth_reads(sub, depth=hot_depth, read_len=read_len, seed=seed)
    h["_abs_pos"] = h["_abs_pos"] + hot_start
    names = h["names"].copy()
    names[h["name_off"][:-1]] = ord("h")
    h["names"] = names
    return merge_reads(rd, h)
