"""Seventh hunt (round 5, no GPU minutes left): MALFORMED inputs.  The messy generator's SAM with records damaged the way broken
pipelines damage them -- CIGARs whose query length disagrees with SEQ, zero-length operations, QUAL / BQ:Z / MM:Z of the wrong length
or with junk in them, positions behind the contig's end, unsorted records, over-long names, truncated lines -- and the BAM made from the
intact SAM damaged at the byte level: cut short at a random offset, or with bytes of the inflated stream overwritten before it is
compressed again (CRC-correct blocks around broken records).  The engine must END: exit status 0 or an error status with a message,
never a signal, a sanitizer report or a hang; where the oracle accepts the same input with status 0 the texts must be equal.
    STA_EXE=tests/cpu/hipemu/_build/asan/samtools_amd/bin/samtools-amd ASAN_OPTIONS=detect_leaks=0 python scripts/hunt7.py <seed> ..."""
import gzip, os, random, struct, subprocess, sys, zlib
sys.path.insert(0, "tests")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth_rich import write_rich_sam
from bamio import sam_to_bam, bgzf_compress
import hunt5

EXE = os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")
ORACLE = "oracle/_build/oracle_samtools"
N_CASES = int(os.environ.get("HUNT7_CASES", "30"))


def damage_sam(rnd, src, dst):
    out, kinds = [], rnd.sample(range(12), rnd.randint(1, 2))      # one or two kinds of damage per file: most kinds are not fatal alone
    for line in open(src):
        if line.startswith("@") or rnd.random() > 0.01:
            out.append(line); continue
        f = line.rstrip("\n").split("\t")
        k = rnd.choice(kinds)
        if k == 0: f[5] = "%dM" % (len(f[9]) + rnd.randint(1, 40))                 # CIGAR longer than SEQ
        elif k == 1: f[5] = "%dM" % max(1, len(f[9]) - rnd.randint(1, 30))         # shorter
        elif k == 2: f[5] = "0M" + f[5] + "0D0I"                                   # zero-length operations
        elif k == 3 and f[10] != "*": f[10] = f[10][:len(f[10]) // 2]              # QUAL too short
        elif k == 4: f.append("BQ:Z:" + "@" * rnd.randint(0, 5))                   # BQ:Z too short
        elif k == 5: f.append("MM:Z:" + rnd.choice(["C+m,900;", "C+m,1,x;", "Q+m,1;", "C+m", ";;;", "C+m,1,2,3;C+h,-1;"]) + "\tML:B:C,1")
        elif k == 6: f[3] = str(rnd.choice([0, 2 ** 31 - 1, 10 ** 7]))            # position 0 / behind every contig
        elif k == 7: f[0] = "n" * rnd.choice([254, 255, 300])                      # name at and over the BAM limit
        elif k == 8: f = f[:rnd.randint(3, 10)]                                    # truncated line
        elif k == 9: f[5] = rnd.choice(["*", "5Z", "M", "10", "4294967295M", "3M-2D3M"])
        elif k == 10: f[9] = f[9][:len(f[9]) // 2] + "?!" + f[9][len(f[9]) // 2 + 2:]     # junk in SEQ
        elif k == 11: f[1] = rnd.choice(["65535", "-1", "x"])
        out.append("\t".join(f) + "\n")
    if rnd.random() < 0.3 and len(out) > 50:                                       # unsorted: two records swapped
        i = rnd.randint(20, len(out) - 20); j = rnd.randint(20, len(out) - 20)
        out[i], out[j] = out[j], out[i]
    open(dst, "w").write("".join(out))


def inflate_all(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def damage_bam(rnd, src, dst):
    raw = open(src, "rb").read()
    k = rnd.random()
    if k < 0.35:                                                                   # cut short (mid-block, mid-record, no EOF block)
        open(dst, "wb").write(raw[:rnd.randint(30, len(raw) - 1)])
        return "cut"
    data = bytearray(inflate_all(src))
    l_text, = struct.unpack_from("<i", data, 4)
    n_ref, = struct.unpack_from("<i", data, 8 + l_text)
    p = 12 + l_text
    for _ in range(n_ref):
        l, = struct.unpack_from("<i", data, p); p += 8 + l
    first = p
    if k < 0.8:                                                                    # bytes inside the records overwritten
        for _ in range(rnd.randint(1, 6)):
            i = rnd.randint(first, len(data) - 1)
            data[i] = rnd.choice([0, 0xff, data[i] ^ (1 << rnd.randrange(8)), rnd.randrange(256)])
        what = "bytes"
    else:                                                                          # a record's block_size / l_read_name / n_cigar_op / l_seq rewritten
        recs, q = [], first
        while q + 36 <= len(data):
            bs, = struct.unpack_from("<i", data, q)
            if bs < 32 or q + 4 + bs > len(data): break
            recs.append(q); q += 4 + bs
        q = rnd.choice(recs)
        fld = rnd.choice([0, 12, 16, 20])      # block_size, l_read_name(+mapq,bin), flag_nc, l_seq
        if fld == 0: struct.pack_into("<i", data, q, rnd.choice([0, 31, 33, -5, 2 ** 30, struct.unpack_from("<i", data, q)[0] + rnd.choice([-3, 7])]))
        elif fld == 12: data[q + 12] = rnd.choice([0, 1, 255])
        elif fld == 16: struct.pack_into("<H", data, q + 16, rnd.choice([0, 65535, 3000]))
        else: struct.pack_into("<i", data, q + 20, rnd.choice([0, -1, 2 ** 31 - 1, 10 ** 6]))
        what = "field%d" % fld
    open(dst, "wb").write(bgzf_compress(bytes(data), level=1, block=rnd.choice([3000, 0xff00])))
    return what


def main():
    bad = total = 0
    for seed in hunt5.seeds:
        rnd = random.Random(seed * 15485863 + 7)
        out = "/tmp/hunt7_%d" % seed; os.makedirs(out, exist_ok=True)
        sam0, fa = write_rich_sam(out, seed=5000 + seed, n_templates=rnd.choice([300, 800]))
        bam0 = sam_to_bam(sam0, os.path.join(out, "ok.bam"), level=1, block=20000)
        bed = os.path.join(out, "r.bed")
        with open(bed, "w") as f:
            f.write("c1\t100\t9000\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
        for case in range(N_CASES):
            if rnd.random() < 0.5:
                inp = os.path.join(out, "bad_%d.sam" % case); damage_sam(rnd, sam0, inp); how = "sam"
            else:
                inp = os.path.join(out, "bad_%d.bam" % case); how = damage_bam(rnd, bam0, inp)
            k = rnd.random()
            if k < 0.45: args = hunt5.draw_mpileup(rnd, fa, bed)
            elif k < 0.6: args = hunt5.draw_depth(rnd, bed)
            elif k < 0.7: args = hunt5.draw_consensus(rnd)
            elif k < 0.8: args = hunt5.draw_calmd(rnd)
            else: args = hunt5.draw_other(rnd, fa, bed)[0]
            files = [inp, fa] if args[0] == "calmd" else [inp]
            env = {}
            if rnd.random() < 0.5: env["STA_WINDOW_COLS"] = str(rnd.choice([37, 300, 3000]))
            if rnd.random() < 0.3: env["STA_WINDOW_READS"] = str(rnd.choice([5, 50]))
            if inp.endswith(".bam") and rnd.random() < 0.3: env["STA_GPU_INFLATE"] = "1"
            try:
                p = subprocess.run([EXE] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=600)
                rc, got, err = p.returncode, p.stdout, p.stderr
            except subprocess.TimeoutExpired:
                rc, got, err = -999, b"", b"timeout"
            total += 1
            why = None
            if rc < 0 or rc > 2: why = "ended by a signal / odd status"
            elif b"Sanitizer" in err or b"runtime error" in err: why = "sanitizer report"
            elif rc != 0 and not err.strip(): why = "error status without a message"
            elif rc == 0 and inp.endswith(".sam"):
                o = subprocess.run([ORACLE] + args + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                if o.returncode == 0 and o.stdout != got: why = "text differs from the oracle's (both accept the input)"
            print("%s seed %d case %d [%s] %s %s rc=%d" % ("FAIL" if why else "ok  ", seed, case, how, env, " ".join(a if len(a) < 30 else "~" + os.path.basename(a) for a in args + files), rc), flush=True)
            if why:
                bad += 1
                print("   ", why, "\n    stderr:", err.decode(errors="replace")[-600:].replace("\n", " | "))
    print("hunt7: %d failures in %d runs" % (bad, total))


if __name__ == "__main__":
    main()
