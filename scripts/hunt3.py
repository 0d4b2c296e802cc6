"""Third bug hunt: messy multi-contig inputs (tests/synth_rich.py), many option sets, default and tiny windows."""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth_rich import write_rich_sam
from bamio import sam_to_bam
out = "/tmp/hunt3"; os.makedirs(out, exist_ok=True)
seeds = [int(x) for x in sys.argv[1:]] or [1, 2]
for seed in seeds:
    sam, fa = write_rich_sam(out, seed=seed, n_templates=6000)
    sam2, _ = write_rich_sam(out, seed=seed + 100, n_templates=2500)
    # depth -q on a record without SEQ reads qual[] out of bounds in the reference (bam2depth.c:165-195 has no l_qseq
    # check on the M path): undefined there, so those records are left out of the -q cases
    sam_q = os.path.join(out, "rich_%d_seq.sam" % seed)
    with open(sam_q, "w") as fo:
        for l in open(sam):
            if l[0] == "@" or l.split("\t")[9] != "*": fo.write(l)
    bam = sam_to_bam(sam, os.path.join(out, "rich_%d.bam" % seed), level=1, block=20000)      # engine reads BAM (threaded BGZF), oracle the SAM text
    bed = os.path.join(out, "r%d.bed" % seed)
    with open(bed, "w") as f:
        f.write("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
    CASES = [
        ["mpileup", "-f", fa, sam], ["mpileup", "-B", "-a", "-a", "-f", fa, sam], ["mpileup", "-A", "-x", sam], ["mpileup", "-E", "-A", "-f", fa, sam, sam2],
        ["mpileup", "-B", "-Q", "0", "-q", "5", "--ff", "UNMAP", "--rf", "PAIRED", "-s", "-O", "--output-extra", "FLAG,RNEXT,NM,RG,QNAME", "-f", fa, sam],
        ["mpileup", "-B", "-l", bed, "-a", sam], ["mpileup", "-r", "c3:1000-30000", "-d", "15", "-f", fa, sam], ["mpileup", "-6", "-B", "--reverse-del", "--no-output-ins", sam],
        ["mpileup", "-G", os.path.join(out, "rg.txt"), "-C", "50", "-f", fa, sam],
        ["depth", sam], ["depth", "-a", "-s", "-J", sam, sam2], ["depth", "-aa", "-q", "10", "-Q", "5", "-l", "60", sam_q], ["depth", "-b", bed, "-g", "0x400", "-G", "16", sam], ["depth", "-r", "c2", "-a", sam],
        ["plpdump", sam], ["plpdump", "-x", sam, sam2], ["plpdump", "-p", sam], ["plpdump", "-d", "12", sam],
        ["glf", "-f", fa, sam], ["glf", "-Q", "3", sam],
        ["coverage", sam, sam2], ["coverage", "-r", "c3:5000-20000", "-Q", "10", "-q", "5", "-l", "60", sam], ["bedcov", "-j", "-d", "8", "-c", bed, sam, sam2], ["bedcov", "-Q", "20", "-g", "1024", bed, sam],
    ]
    open(os.path.join(out, "rg.txt"), "w").write("g2\n")
    for args in CASES:
        o = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        want = o.stdout.split(b"\n")
        for envx in ({}, {"STA_WINDOW_COLS": "900", "STA_PLP_BATCH": "700"}):
            eargs = [bam if (a == sam and envx) else a for a in args]
            p = subprocess.run([os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")] + eargs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx))
            got = p.stdout.split(b"\n")
            nd = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
            ok = p.returncode == o.returncode and len(got) == len(want) and not nd
            print("%s seed %d %-7s %s rc=%d/%d lines %d/%d differing %d %s" % ("ok  " if ok else "FAIL", seed, "small" if envx else "default", " ".join(a if len(a) < 24 else "~" + os.path.basename(a) for a in args)[:110], p.returncode, o.returncode, len(got), len(want), len(nd), nd[:3]))
            if not ok:
                for i in nd[:1]:
                    print("   got ", got[i][:300]); print("   want", want[i][:300])
                if p.returncode != o.returncode: print("   stderr engine:", p.stderr.decode()[-200:].replace("\n", " | "), " oracle:", o.stderr.decode()[-200:].replace("\n", " | "))
