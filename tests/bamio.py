"""Minimal SAM -> BAM (BGZF) writer for tests and host-I/O benchmarks (SAM spec v1 sections 4.1-4.2).
Not a general converter: what the synthetic generators and the reference's test SAMs contain."""
import struct
import zlib

_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_OPS = {c: i for i, c in enumerate("MIDNSHP=XB")}
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _aux(field):
    tag, typ, val = field[:2], field[3], field[5:]
    t = tag.encode()
    if typ == "A": return t + b"A" + val.encode()[:1]
    if typ == "i":
        v = int(val)
        for code, fmt, lo, hi in (("C", "<B", 0, 255), ("c", "<b", -128, 127), ("S", "<H", 0, 65535), ("s", "<h", -32768, 32767),
                                  ("I", "<I", 0, 2 ** 32 - 1), ("i", "<i", -2 ** 31, 2 ** 31 - 1)):
            if lo <= v <= hi: return t + code.encode() + struct.pack(fmt, v)
        raise ValueError(field)
    if typ == "f": return t + b"f" + struct.pack("<f", float(val))
    if typ in "ZH": return t + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub = val[0]; items = [x for x in val[2:].split(",") if x]
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        conv = float if sub == "f" else int
        return t + b"B" + sub.encode() + struct.pack("<I", len(items)) + b"".join(struct.pack("<" + fmt, conv(x)) for x in items)
    raise ValueError(field)


def _bgzf_block(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def bam_header_bytes(header_lines, names, lens):
    text = ("\n".join(header_lines) + "\n").encode() if header_lines else b""
    out = [b"BAM\1", struct.pack("<i", len(text)), text, struct.pack("<i", len(names))]
    for n, l in zip(names, lens):
        nb = n.encode() + b"\0"
        out.append(struct.pack("<i", len(nb)) + nb + struct.pack("<i", min(l, 2 ** 31 - 1)))
    return b"".join(out)


def bam_record_bytes(line, tid):
    """one SAM text line -> one BAM alignment record (block_size prefix included); tid: reference name -> id"""
    f = line.split("\t")
    qn = f[0].encode() + b"\0"
    flag = int(f[1], 0); ref = -1 if f[2] == "*" else tid[f[2]]; pos = int(f[3]) - 1; mapq = int(f[4])
    cig = []
    if f[5] != "*":
        num = ""
        for ch in f[5]:
            if ch.isdigit(): num += ch
            else: cig.append(int(num) << 4 | _OPS[ch]); num = ""
    mref = ref if f[6] == "=" else (-1 if f[6] == "*" else tid[f[6]])
    mpos = int(f[7]) - 1; tlen = int(f[8])
    seq = "" if f[9] == "*" else f[9]
    l = len(seq)
    sb = bytearray((l + 1) // 2)
    for i, ch in enumerate(seq): sb[i >> 1] |= _NT16.get(ch.upper(), 15) << (4 if i % 2 == 0 else 0)
    qb = b"\xff" * l if f[10] == "*" else bytes(ord(ch) - 33 for ch in f[10])
    rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
    end = pos + (rlen if rlen > 0 and not (flag & 4) else 1)
    body = (struct.pack("<iiBBHHHiiii", ref, pos, len(qn), mapq, _reg2bin(max(pos, 0), max(end, 1)), len(cig), flag, l, mref, mpos, tlen)
            + qn + b"".join(struct.pack("<I", c) for c in cig) + bytes(sb) + qb + b"".join(_aux(a) for a in f[11:]))
    return struct.pack("<i", len(body)) + body


def bgzf_compress(raw, level=1, block=0xff00):
    """raw bytes -> concatenated BGZF blocks (no EOF marker)"""
    return b"".join(_bgzf_block(raw[o:o + block], level) for o in range(0, len(raw), block))


def sam_to_bam(sam_path, bam_path, level=1, block=0xff00):
    """block: payload bytes per BGZF block (small values make many-block files for the threaded reader tests)."""
    header, names, lens, recs = [], [], [], []
    with open(sam_path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line: continue
            if line[0] == "@":
                header.append(line)
                if line.startswith("@SQ"):
                    d = dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f)
                    names.append(d["SN"]); lens.append(int(d["LN"]))
                continue
            recs.append(line)
    tid = {n: i for i, n in enumerate(names)}
    raw = bam_header_bytes(header, names, lens) + b"".join(bam_record_bytes(line, tid) for line in recs)
    with open(bam_path, "wb") as fo:
        for o in range(0, len(raw), block):
            fo.write(_bgzf_block(raw[o:o + block], level))
        fo.write(_EOF)
    return bam_path


def bam_filter(src, dst, require=0, exclude=0, level=1):
    """`samtools view -b -f require -F exclude` for tests: copies the header and the records whose FLAG passes."""
    import gzip
    raw = gzip.open(src).read()
    l_text = struct.unpack("<i", raw[4:8])[0]
    o = 8 + l_text
    n_ref = struct.unpack("<i", raw[o:o + 4])[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack("<i", raw[o:o + 4])[0]
        o += 4 + l_name + 4
    out = [raw[:o]]
    while o < len(raw):
        bs = struct.unpack("<i", raw[o:o + 4])[0]
        flag = struct.unpack("<H", raw[o + 4 + 14:o + 4 + 16])[0]
        if (flag & require) == require and not (flag & exclude):
            out.append(raw[o:o + 4 + bs])
        o += 4 + bs
    data = b"".join(out)
    with open(dst, "wb") as fo:
        for k in range(0, len(data), 0xff00):
            fo.write(_bgzf_block(data[k:k + 0xff00], level))
        fo.write(_EOF)
    return dst


def write_bai(bam_path, bai_path=None):
    """A BAI index (SAM spec 5.2: bins, chunks, 16 kbp linear index) for any coordinate-sorted BGZF BAM -- the test stand-in for
    `samtools index`.  Walks the BGZF blocks for the virtual offsets and the records for positions."""
    raw_file = open(bam_path, "rb").read()
    # blocks: compressed start, inflated start, inflated size
    blocks, o, u, parts = [], 0, 0, []
    while o < len(raw_file):
        xlen = struct.unpack("<H", raw_file[o + 10:o + 12])[0]
        bsize = None
        x = raw_file[o + 12:o + 12 + xlen]; k = 0
        while k + 4 <= xlen:
            sl = struct.unpack("<H", x[k + 2:k + 4])[0]
            if x[k:k + 2] == b"BC": bsize = struct.unpack("<H", x[k + 4:k + 6])[0] + 1
            k += 4 + sl
        data = zlib.decompress(raw_file[o + 12 + xlen:o + bsize - 8], -15)
        blocks.append((o, u, len(data))); parts.append(data)
        o += bsize; u += len(data)
    raw = b"".join(parts)
    starts = [b[1] for b in blocks]
    import bisect

    def voff(upos):
        i = bisect.bisect_right(starts, upos) - 1
        while i + 1 < len(blocks) and blocks[i][2] == 0: i += 1
        if upos >= blocks[i][1] + blocks[i][2] and i + 1 < len(blocks): i += 1
        return blocks[i][0] << 16 | (upos - blocks[i][1])

    l_text = struct.unpack("<i", raw[4:8])[0]
    p = 8 + l_text
    n_ref = struct.unpack("<i", raw[p:p + 4])[0]; p += 4
    for _ in range(n_ref):
        l_name = struct.unpack("<i", raw[p:p + 4])[0]; p += 4 + l_name + 4
    bins = [dict() for _ in range(n_ref)]
    lin = [dict() for _ in range(n_ref)]
    n_no_coor = 0
    while p < len(raw):
        bs = struct.unpack("<i", raw[p:p + 4])[0]
        ref, pos, l_rn, mapq, bn, n_cig, flag, l_seq = struct.unpack("<iiBBHHHi", raw[p + 4:p + 24])
        if ref < 0:
            n_no_coor += 1
        else:
            cig = struct.unpack("<%dI" % n_cig, raw[p + 36 + l_rn:p + 36 + l_rn + 4 * n_cig]) if n_cig else ()
            rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
            end = pos + (rlen if rlen > 0 and not (flag & 4) else 1)
            v0, v1 = voff(p), voff(p + 4 + bs)
            b = _reg2bin(max(pos, 0), max(end, 1))
            ch = bins[ref].setdefault(b, [])
            if ch and ch[-1][1] >> 16 == v0 >> 16: ch[-1][1] = v1          # same block: extend the chunk
            else: ch.append([v0, v1])
            for w in range(max(pos, 0) >> 14, ((max(end, 1) - 1) >> 14) + 1):
                if w not in lin[ref] or v0 < lin[ref][w]: lin[ref][w] = v0
        p += 4 + bs
    out = [b"BAI\1", struct.pack("<i", n_ref)]
    for t in range(n_ref):
        out.append(struct.pack("<i", len(bins[t])))
        for b in sorted(bins[t]):
            out.append(struct.pack("<Ii", b, len(bins[t][b])))
            for v0, v1 in bins[t][b]: out.append(struct.pack("<QQ", v0, v1))
        n_intv = max(lin[t]) + 1 if lin[t] else 0
        out.append(struct.pack("<i", n_intv))
        prev = 0
        for w in range(n_intv):                         # empty windows take the offset of the next filled one's predecessor (as samtools fills them)
            prev = lin[t].get(w, prev)
            out.append(struct.pack("<Q", prev))
    out.append(struct.pack("<Q", n_no_coor))
    bai_path = bai_path or bam_path + ".bai"
    with open(bai_path, "wb") as fo:
        fo.write(b"".join(out))
    return bai_path
