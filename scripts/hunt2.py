"""Second bug hunt: option combinations at scale (engine vs oracle)."""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth import write_synth_sam
out = "/tmp/hunt2"; os.makedirs(out, exist_ok=True)
def gen(tag, **kw):
    d = os.path.join(out, tag); os.makedirs(d, exist_ok=True)
    return write_synth_sam(d, **kw)
s1, fa = gen("a", n_ref=120000, depth=25, read_len=150, seed=201, paired=True, indel_rate=0.05, max_indel=5)
s2, _ = gen("b", n_ref=120000, depth=15, read_len=100, seed=202, paired=True, indel_rate=0.08, max_indel=4)
s3, _ = gen("c", n_ref=120000, depth=8, read_len=150, seed=203, paired=False, indel_rate=0.02)
deep, dfa = gen("deep", n_ref=20000, depth=400, read_len=150, seed=204, paired=True, indel_rate=0.03)
bed = os.path.join(out, "r.bed")
with open(bed, "w") as f:
    for b in range(1000, 119000, 7000): f.write("chrS\t%d\t%d\n" % (b, b + 1500))
CASES = [
    ("three_files_baq", ["mpileup", "-f", fa, s1, s2, s3]),
    ("three_files_B_a", ["mpileup", "-B", "-a", "-a", "-f", fa, s1, s2, s3]),
    ("maxdepth_50", ["mpileup", "-B", "-d", "50", "-f", dfa, deep]),
    ("maxdepth_120_baq", ["mpileup", "-d", "120", "-f", dfa, deep]),
    ("bed_l", ["mpileup", "-B", "-l", bed, "-f", fa, s1, s2]),
    ("bed_l_a", ["mpileup", "-B", "-a", "-l", bed, "-f", fa, s1]),
    ("region", ["mpileup", "-r", "chrS:50000-70000", "-f", fa, s1, s3]),
    ("extras", ["mpileup", "-B", "-s", "-O", "--output-extra", "FLAG,QNAME,RNEXT,PNEXT,RLEN,MAPQ", "--output-BP-5", "-f", fa, s2]),
    ("rev_del_noends", ["mpileup", "-B", "--reverse-del", "--no-output-ends", "--no-output-ins", "--no-output-del", "-f", fa, s1]),
    ("q_Q_ff", ["mpileup", "-q", "10", "-Q", "26", "--ff", "UNMAP,DUP,REVERSE", "-f", fa, s1]),
    ("illumina_E", ["mpileup", "-6", "-E", "-f", fa, s3]),
    ("depth_a_two", ["depth", "-a", s1, s2]),
    ("depth_q_l_G", ["depth", "-q", "20", "-Q", "1", "-l", "120", "-G", "16", s1, s3]),
    ("depth_b_aa", ["depth", "-aa", "-b", bed, s1]),
    ("bedcov_multi", ["bedcov", "-j", "-d", "20", "-c", bed, s1, s2]),
    ("coverage_multi", ["coverage", "-Q", "20", "--min-depth", "30", s1, s2, s3]),
    ("plpdump_three", ["plpdump", s1, s2, s3]),
    ("plpdump_maxcnt", ["plpdump", "-d", "100", deep]),
]
for name, args in CASES:
    t0 = time.time()
    o = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    want = o.stdout.split(b"\n")
    t1 = time.time()
    for envx in ({}, {"STA_WINDOW_COLS": "7000", "STA_PLP_BATCH": "2500"}):
        p = subprocess.run([os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx))
        got = p.stdout.split(b"\n")
        nd = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
        print("%-18s %s rc=%d/%d lines %d/%d differing %d first %s (oracle %.1fs) %s" % (name, "small" if envx else "default", p.returncode, o.returncode, len(got), len(want), len(nd), nd[:3], t1 - t0, p.stderr.decode()[-120:].replace("\n", "|") if p.returncode else ""))
        for i in nd[:1]:
            print("   got ", got[i][:300]); print("   want", want[i][:300])
