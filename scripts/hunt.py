"""Bug hunt: engine vs oracle on larger seeded inputs than the test-suite uses (rare order/geometry corners)."""
import os, subprocess, sys, time
sys.path.insert(0, "tests")
from synth import write_synth_sam
out = "/tmp/hunt"; os.makedirs(out, exist_ok=True)
CASES = [
    ("pairs_indels_baq", dict(n_ref=200000, depth=30, read_len=150, seed=101, paired=True, indel_rate=0.08, max_indel=6), ["mpileup", "-f", "{fa}", "{sam}"]),
    ("pairs_indels_B", dict(n_ref=400000, depth=30, read_len=150, seed=102, paired=True, indel_rate=0.15, max_indel=9), ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    ("pairs_EA", dict(n_ref=150000, depth=40, read_len=100, seed=103, paired=True, indel_rate=0.05, max_indel=4, sub_rate=0.01), ["mpileup", "-E", "-A", "-f", "{fa}", "{sam}"]),
    ("depth_s_J", dict(n_ref=400000, depth=30, read_len=150, seed=104, paired=True, indel_rate=0.1, max_indel=8), ["depth", "-s", "-J", "-q", "11", "{sam}"]),
    ("deep_200x", dict(n_ref=40000, depth=200, read_len=150, seed=105, paired=True, indel_rate=0.05), ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    ("plpdump_pairs", dict(n_ref=150000, depth=30, read_len=150, seed=106, paired=True, indel_rate=0.1, max_indel=7), ["plpdump", "{sam}"]),
    ("C50_pairs", dict(n_ref=150000, depth=30, read_len=150, seed=107, paired=True, indel_rate=0.05, sub_rate=0.02), ["mpileup", "-C", "50", "-f", "{fa}", "{sam}"]),
    ("bedcov_like_cov", dict(n_ref=300000, depth=30, read_len=150, seed=108, paired=True, indel_rate=0.1, max_indel=8), ["coverage", "-Q", "12", "{sam}"]),
]
for name, gen, argv in CASES:
    d = os.path.join(out, name); os.makedirs(d, exist_ok=True)
    sam, fa = write_synth_sam(d, **gen)
    args = [a.format(sam=sam, fa=fa) for a in argv]
    t0 = time.time()
    want = subprocess.run(["oracle/_build/oracle_samtools"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.split(b"\n")
    t1 = time.time()
    for envx in ({}, {"STA_WINDOW_COLS": "5000", "STA_PLP_BATCH": "3000"}):
        p = subprocess.run([os.environ.get("STA_EXE", "samtools_amd/bin/samtools-amd")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **envx))
        got = p.stdout.split(b"\n")
        nd = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
        print("%-18s %s rc=%d lines %d/%d differing %d first %s (oracle %.1fs)" % (name, "small-windows" if envx else "default", p.returncode, len(got), len(want), len(nd), nd[:3], t1 - t0))
        for i in nd[:2]:
            print("   got ", got[i][:250]); print("   want", want[i][:250])
