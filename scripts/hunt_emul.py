"""Parity hunt without a GPU: the tile kernels' step functions (tests/cpu/plp_emul.cpp runs samtools_amd/csrc/plp_tile.h on the CPU)
against the oracle on randomly drawn windows -- depth, CIGAR mess, -Q, LDS slice size, --no-output-ends, -a.
    python scripts/hunt_emul.py [cases=100] [seed=1]"""
import os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests"))
import test_plp_emul as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(prefix="hunt_emul")
exes = []
for mode in (0, 1):
    exe = os.path.join(tmp, "plp_emul%d" % mode)
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DPLP_EMUL_ANY=%d" % mode, "-I/opt/rocm/include",
                    os.path.join(R, "tests", "cpu", "plp_emul.cpp"), "-o", exe], check=True)
    exes.append(exe)
ORA = os.path.join(R, "oracle", "_build", "oracle_samtools")
bad = 0
for k in range(n_cases):
    n_cols = rnd.choice([3000, 6000, 9000, 15000]); depth = rnd.choice([1, 3, 10, 30, 30, 60, 120, 250])
    seed = rnd.randint(1, 10 ** 6); minq = rnd.choice([0, 13, 13, 20, 30, 41, 93]); cap = rnd.choice([1024, 2048, 4096, 6144, 12288])
    no_ends = rnd.random() < 0.2; all_ = rnd.random() < 0.2; messy = rnd.choice([0.0, 0.08, 0.3, 0.7])
    ref, rd, span, simple = T._messy_reads(n_cols, depth, seed, frac_messy=messy)
    d = os.path.join(tmp, "c"); os.makedirs(d, exist_ok=True)
    d, sam, fa = T._dump(d, ref, rd, span, simple, n_cols)
    args = ["mpileup", "-B", "-Q", str(minq), "-d", "1000000", "-f", fa] + (["--no-output-ends"] if no_ends else []) + (["-a"] if all_ else [])
    want = subprocess.run([ORA] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    for exe in exes:
        got = subprocess.run([exe, d, str(minq), str(cap), str(int(no_ends)), str(int(all_))], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if got.returncode != 0 or got.stdout != want:
            bad += 1
            print("MISMATCH case %d: n_cols=%d depth=%d seed=%d minq=%d cap=%d no_ends=%d all=%d messy=%.2f (%s) rc=%d" % (k, n_cols, depth, seed, minq, cap, no_ends, all_, messy, os.path.basename(exe), got.returncode), flush=True)
            break
    if k % 20 == 19:
        print("%d cases, %d mismatches" % (k + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (n_cases, bad))
