#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops from hipcc's -save-temps assembly (no GPU needed).

    hipcc ... -save-temps=obj -c kernels_baq.hip -o /tmp/isa/x.o
    python scripts/isa_loops.py /tmp/isa/kernels_baq-hip-amdgcn-amd-amdhsa-gfx950.s k_baq_bwdILi7ELi2ELb1

Every wave64 VALU instruction occupies its SIMD for four cycles on CDNA (16 lanes per clock, fp64 included), so the number of vector
instructions in a loop body x its trip count is the issue floor of the loop.  Prints, per basic block that lies inside a loop
(a backward branch spans it), the instruction counts by class."""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_"):
        if op.endswith("_f64") or "_f64_" in op:
            if op.startswith(("v_cmp", "v_cmpx")): return "v_cmp"
            return "v_f64"
        if op.startswith("v_cndmask"): return "v_cndmask"
        if op.startswith(("v_cmp", "v_cmpx")): return "v_cmp"
        if op.startswith(("v_mov", "v_accvgpr")): return "v_mov"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "v_lane"
        return "v_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "s_wait"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_load", "flat_load", "buffer_load")): return "vmem_ld"
    if op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic")): return "vmem_st"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    order = {}
    for i in range(start + 1, end + 1):
        l = lines[i].split(";")[0].strip()
        if not l or l.startswith("."):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                cur = m.group(1); blocks[cur] = []
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1); blocks[cur] = []; continue
        blocks[cur].append(l)
    names = list(blocks)
    idx = {n: i for i, n in enumerate(names)}
    # loops: a branch to an earlier (or same) block
    loops = []
    for n in names:
        for ins in blocks[n]:
            op = ins.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = ins.split()[-1]
                if tgt in idx and idx[tgt] <= idx[n]:
                    loops.append((idx[tgt], idx[n]))
    print("kernel", lines[start][:100])
    tot = Counter()
    for n in names:
        for ins in blocks[n]:
            tot[classify(ins.split()[0])] += 1
    print("whole kernel:", dict(tot))
    for lo, hi in sorted(set(loops)):
        c = Counter()
        ops = Counter()
        for n in names[lo:hi + 1]:
            for ins in blocks[n]:
                op = ins.split()[0]
                c[classify(op)] += 1
                ops[op] += 1
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("loop %s .. %s (%d blocks): VALU %d  %s" % (names[lo], names[hi], hi - lo + 1, valu, dict(c)))
        if "-v" in sys.argv:
            print("   ", ops.most_common(40))


main()
