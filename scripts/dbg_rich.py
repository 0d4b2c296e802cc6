import os, subprocess, sys
sys.path.insert(0, "tests")
from synth_rich import write_rich_sam
seed = int(sys.argv[1]); n = int(sys.argv[2])
out = "/tmp/dbgrich"; os.makedirs(out, exist_ok=True)
sam, fa = write_rich_sam(out, seed=seed, n_templates=n)
bed = os.path.join(out, "r.bed"); open(bed, "w").write("c1\t100\t9000\nc1\t9500\t9600\nc2\t0\t4000\tname\nc3\t20000\t44000\n")
cases = [["mpileup", "-f", fa, sam], ["mpileup", "-B", "-Q", "0", "-s", "-O", "--output-extra", "FLAG,RNEXT,NM,RG", "-f", fa, sam],
         ["mpileup", "-B", "-l", bed, "-a", sam], ["mpileup", "-r", "c3:1000-30000", "-d", "15", "-f", fa, sam],
         ["mpileup", "-C", "50", "-f", fa, sam], ["depth", "-a", "-s", "-J", sam], ["plpdump", "-x", sam], ["plpdump", "-d", "12", sam],
         ["coverage", sam], ["bedcov", "-j", "-d", "8", "-c", bed, sam]]
for args in cases:
    want = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
    for envx in ({}, {"STA_WINDOW_COLS": "900", "STA_PLP_BATCH": "700"}):
        got = subprocess.run(["samtools_amd/bin/samtools-amd"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx)).stdout.split(b"\n")
        nd = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
        print(args[:4], bool(envx), len(got), len(want), len(nd))
        for i in nd[:3]: print("  got ", got[i][:200]); print("  want", want[i][:200])
