#!/bin/bash
# what the GPU box's host side really offers: cgroup CPU quota, affinity, and how a CPU-bound job scales with processes
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
nproc; taskset -p $$; lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA|Model name" 
python3 - <<'PY'
import time, zlib, os, multiprocessing as mp
data = os.urandom(1 << 16) * 4 + bytes(1 << 18)
comp = zlib.compress(data, 1)
def work(_):
    t = time.perf_counter(); n = 0
    while time.perf_counter() - t < 1.0:
        zlib.decompress(comp); n += 1
    return n
for p in (1, 8, 16, 32, 64, 128):
    with mp.Pool(p) as pool:
        t0 = time.perf_counter(); r = pool.map(work, range(p)); dt = time.perf_counter() - t0
    print("procs %3d: %6d inflates/s total, %5d per proc" % (p, sum(r) / 1.0, sum(r) / p))
PY
