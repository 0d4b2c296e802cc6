import os, subprocess, sys
sys.path.insert(0, "tests")
from synth import write_synth_sam
out = "gpurun_out/dbg"; os.makedirs(out, exist_ok=True)
sam, fa = write_synth_sam(out, n_ref=500000, depth=30, read_len=150, seed=55, paired=True)
args = ["mpileup", "-f", fa, sam]
want = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
for name, envx in (("default", {}), ("no_side", {"STA_BAQ_NO_SIDE_STREAM": "1"}), ("force_slow", {"STA_BAQ_FORCE_SLOW": "1"}), ("no_olap(-x)", {})):
    env = dict(os.environ, STA_DEBUG="1"); env.update(envx)
    a = args if name != "no_olap(-x)" else ["mpileup", "-x", "-f", fa, sam]
    w = want if name != "no_olap(-x)" else subprocess.run(["oracle/_build/oracle_samtools"] + a, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
    p = subprocess.run(["samtools_amd/bin/samtools-amd"] + a, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    got = p.stdout.split(b"\n")
    nd = [i for i, (x, y) in enumerate(zip(got, w)) if x != y]
    print("%s: %d differing lines of %d; first %s; %s" % (name, len(nd), len(w), nd[:5], p.stderr.decode().replace("\n", " | ")[-200:]))
