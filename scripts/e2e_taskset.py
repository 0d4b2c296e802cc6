"""scratch: does pinning the driver to one NUMA node / fewer cores change the end-to-end time?  (GPU box: 2 x 64-core EPYC 9575F,
NUMA node0 = CPUs 0-63,128-191, node1 = 64-127,192-255).  Needs the input of scripts/e2e_big.py in /dev/shm/sta_e2e."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "samtools_amd", "bin", "samtools-amd")
B, F = "/dev/shm/sta_e2e/big.bam", "/dev/shm/sta_e2e/big.fa"
env = dict(os.environ, STA_IO_THREADS="16", STA_STAGE_THREADS="4")


def t(cpus, args):
    a = time.perf_counter()
    p = subprocess.run(["taskset", "-c", cpus, E] + args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    return "%.2f%s" % (time.perf_counter() - a, "" if p.returncode == 0 else "(rc %d)" % p.returncode)


for cpus in sys.argv[1:] or ["0-255", "0-63", "0-31", "0-63,128-191", "64-127", "64-95"]:
    print("taskset %-14s depth -a %s %s   mpileup -f %s   mpileup -B -f %s" % (cpus, t(cpus, ["depth", "-a", B]), t(cpus, ["depth", "-a", B]),
          t(cpus, ["mpileup", "-f", F, B]), t(cpus, ["mpileup", "-B", "-f", F, B])), flush=True)
