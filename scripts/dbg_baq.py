import os, subprocess, sys
sys.path.insert(0, "tests")
from synth import write_synth_sam
out = "gpurun_out/dbg"; os.makedirs(out, exist_ok=True)
sam, fa = write_synth_sam(out, n_ref=20000, depth=30, read_len=150, seed=54, paired=False, indel_rate=0.3, max_indel=14)
args = ["mpileup", "-f", fa, sam]
want = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
for mode in ("0", "1"):
    env = dict(os.environ, STA_DEBUG="1")
    if mode == "1": env["STA_BAQ_FORCE_SLOW"] = "1"
    p = subprocess.run(["samtools_amd/bin/samtools-amd"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    got = p.stdout.split(b"\n")
    nd = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    print("mode force_slow=%s: %d differing lines of %d; stderr: %s" % (mode, len(nd), len(want), p.stderr.decode()[-300:]))
    for i in nd[:8]:
        print(" line", i + 1); print("  got ", got[i][:200]); print("  want", want[i][:200])
