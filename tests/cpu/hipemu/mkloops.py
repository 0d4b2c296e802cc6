#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (tests/cpu/hipemu): the loop table of the emulated library.

hipemu has to decide which lanes of a diverged wave go first without seeing the program's control-flow graph.  This script reads
the disassembly of the emulated library and writes, per function, the address ranges of its loops -- every backward jump
(target <= source) inside a function is a loop [target, source]; the build keeps blocks in source order, so a loop is one range --
as lines `F lo hi name` / `L lo hi`.  hipemu.cpp orders waiting lanes by (iteration counts of the loops they share, code address).

usage: mkloops.py <llvm-objdump> <library> <out>"""
import re
import subprocess
import sys


def main():
    objdump, lib, out = sys.argv[1:4]
    p = subprocess.run([objdump, "-d", "--no-show-raw-insn", lib], stdout=subprocess.PIPE, check=True, text=True)
    fn_re = re.compile(r"^([0-9a-f]{8,16}) <(.+)>:$")
    ins_re = re.compile(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$")
    funcs = []          # [lo, hi, name, {target: max source}]
    cur = None
    for line in p.stdout.split("\n"):
        m = fn_re.match(line)
        if m:
            cur = [int(m.group(1), 16), int(m.group(1), 16), m.group(2), {}]
            funcs.append(cur)
            continue
        if cur is None:
            continue
        m = ins_re.match(line)
        if not m:
            continue
        a = int(m.group(1), 16)
        cur[1] = a
        op = m.group(2)
        if op[0] == "j" or op.startswith("loop"):
            t = re.match(r"0x([0-9a-f]+)", m.group(3))
            if t:
                tgt = int(t.group(1), 16)
                if cur[0] <= tgt <= a:
                    cur[3][tgt] = max(cur[3].get(tgt, 0), a)
    n_loops = 0
    with open(out, "w") as fh:
        for lo, hi, name, loops in funcs:
            fh.write("F %x %x %s\n" % (lo, hi, name))
            for tgt in sorted(loops):
                fh.write("L %x %x\n" % (tgt, loops[tgt]))
                n_loops += 1
    print("mkloops: %d loops in %d functions -> %s" % (n_loops, len(funcs), out))


if __name__ == "__main__":
    main()
