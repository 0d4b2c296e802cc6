"""Parity hunt without a GPU: the consensus path's device step functions on the CPU (tests/cpu/cons_emul.cpp, built by
scripts/build_cons_emul.sh) against the oracle on freshly generated inputs -- indel-heavy pairs with and without MD tags, the messy
multi-contig generator -- times every option set of tests/cons_cases.py, windows cut every 1 Mi / 997 / 64 columns.
    python scripts/hunt_cons_emul.py [rounds=5] [seed=1]"""
import os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests"))
from cons_cases import OPTION_SETS
from mdtag import add_md_tags
from synth import write_synth_sam
from synth_rich import write_rich_sam

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(prefix="hunt_cons")
emul = os.path.join(tmp, "cons_emul")
subprocess.run([os.path.join(R, "scripts", "build_cons_emul.sh"), emul], check=True)
ORA = os.path.join(R, "oracle", "_build", "oracle_samtools")
env = dict(os.environ, STA_NO_PINNED="1")
bad = n = 0
for k in range(rounds):
    d = os.path.join(tmp, "r%d" % k); os.makedirs(os.path.join(d, "rich"), exist_ok=True)
    s = rnd.randint(1, 10 ** 6)
    sam1, fa1 = write_synth_sam(d, n_ref=rnd.choice([6000, 12000]), depth=rnd.choice([8, 30, 60]), read_len=rnd.choice([100, 150]), seed=s, paired=True,
                                indel_rate=rnd.choice([0.02, 0.1, 0.3]), max_indel=rnd.choice([3, 7, 12]))
    sam1md = add_md_tags(sam1, fa1, os.path.join(d, "pairs_md.sam"), every=rnd.choice([1, 2]))
    sam2, fa2 = write_rich_sam(os.path.join(d, "rich"), seed=s + 1, n_templates=rnd.choice([800, 2000]))
    sam2md = add_md_tags(sam2, fa2, os.path.join(d, "rich", "rich_md.sam"), every=3)
    for sam, fa in ((sam1, fa1), (sam1md, fa1), (sam2md, fa2)):
        for opts in OPTION_SETS:
            args = [a.format(fa=fa) for a in opts] + [sam]
            want = subprocess.run([ORA, "consensus"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            for wc in ("1048576", "997", "64"):
                got = subprocess.run([emul, "consensus"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, STA_WINDOW_COLS=wc))
                n += 1
                if got.returncode != want.returncode or got.stdout != want.stdout:
                    bad += 1
                    print("MISMATCH round %d seed %d %s window %s: %s" % (k, s, os.path.basename(sam), wc, " ".join(opts)), flush=True)
    print("round %d (seed %d): %d runs, %d mismatches" % (k, s, n, bad), flush=True)
print("done: %d runs, %d mismatches" % (n, bad))
