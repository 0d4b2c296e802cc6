"""Adds MD:Z tags to a SAM file (test input preparation: `samtools consensus` reads MD in its Bayesian mode,
bam_consensus.c:1138-1203).  Plain restatement of the MD definition in the SAM specification."""
import re


def read_fasta(path):
    seqs, name = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        elif name is not None:
            seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def md_of(pos0, cigar, seq, ref):
    out, run, q, r = [], 0, 0, pos0
    for n, op in re.findall(r"(\d+)([MIDNSHP=X])", cigar):
        n = int(n)
        if op in "M=X":
            for k in range(n):
                rb = ref[r + k].upper() if r + k < len(ref) else "N"
                if seq[q + k].upper() == rb or seq[q + k] == "=":
                    run += 1
                else:
                    out.append(str(run)); out.append(rb); run = 0
            q += n; r += n
        elif op == "D":
            out.append(str(run)); out.append("^" + "".join(ref[r + k].upper() if r + k < len(ref) else "N" for k in range(n))); run = 0
            r += n
        elif op == "N":
            r += n
        elif op in "IS":
            q += n
    out.append(str(run))
    return "".join(out)


def add_md_tags(sam_in, fasta, sam_out, every=1):
    """every = k: only every k-th eligible record gets the tag (the tag is optional in the reference's code path too)"""
    ref = read_fasta(fasta)
    i = 0
    with open(sam_in) as fi, open(sam_out, "w") as fo:
        for line in fi:
            if line.startswith("@"):
                fo.write(line); continue
            f = line.rstrip("\n").split("\t")
            if f[2] in ref and f[5] != "*" and f[9] != "*" and not int(f[1]) & 4:
                i += 1
                if i % every == 0:
                    f.append("MD:Z:" + md_of(int(f[3]) - 1, f[5], f[9], ref[f[2]]))
            fo.write("\t".join(f) + "\n")
    return sam_out
