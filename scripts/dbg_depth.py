import os, subprocess, sys
sys.path.insert(0, "tests")
from synth_rich import write_rich_sam
out = "/tmp/hunt3"; os.makedirs(out, exist_ok=True)
sam, fa = write_rich_sam(out, seed=1, n_templates=6000)
for opts in (["-q", "10"], ["-Q", "5"], ["-l", "60"], ["-q", "10", "-l", "60"], ["-aa", "-q", "10"]):
    args = ["depth"] + opts + [sam]
    want = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
    got = subprocess.run(["samtools_amd/bin/samtools-amd"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.split(b"\n")
    nd = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    print(opts, len(got), len(want), len(nd), [(got[i], want[i]) for i in nd[:2]])
