#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (tests/cpu/hipemu): the control-flow table of the emulated library.

hipemu has to decide which lanes of a diverged wave go first.  The hardware runs an inner divergent region to its end while the
lanes that skipped it stay masked where control flow joins again (the immediate post-dominator).  The emulator gets the same order
from the control-flow graph of the code it runs: this script disassembles the built library, cuts every function into basic blocks,
finds the natural loops (back edge = edge to a dominator) and numbers the blocks in reverse post-order of the graph WITHOUT its back
edges.  A lane is then earlier than another one when it has gone round a loop they are both in fewer times, or -- same iteration --
when its block comes first in that order (if block B can be reached from block A without a back edge, A's number is lower: the lanes
at A are still on their way to B).  Independent of how the compiler laid the blocks out.

Output, one function after the other:
    F <lo> <hi> <name>
    B <start> <rpo> <1 if the block is a loop header> <loop id> ...     # loops that contain the block, outermost first (ids are global)
usage: mkloops.py <llvm-objdump> <library or executable> <out>"""
import re
import subprocess
import sys

NORETURN = re.compile(r"__asan_report|__asan_handle_no_return|__ubsan_handle_\w*_abort|\babort\b|__stack_chk_fail|__cxa_throw|_Unwind_Resume|"
                      r"__assert_fail|__cxa_call_unexpected|St9terminate|__clang_call_terminate|_ZN6hipemu\d+_GLOBAL__N_13die|__cxa_rethrow|"
                      r"__throw_|__cxa_bad")


def analyse(insns):
    """insns: [(addr, mnemonic, operand text)] of one function -> [(block start, rpo, [loop headers outermost first])]"""
    lo, hi = insns[0][0], insns[-1][0]
    addr_idx = {a: i for i, (a, _, _) in enumerate(insns)}
    leaders = {lo}
    term = {}                      # instruction index -> (targets inside the function, falls through)
    for i, (a, op, arg) in enumerate(insns):
        nxt = insns[i + 1][0] if i + 1 < len(insns) else None
        if op.startswith("j") or op.startswith("loop"):
            m = re.match(r"0x([0-9a-f]+)", arg)
            tgt = int(m.group(1), 16) if m else None
            inside = tgt is not None and tgt in addr_idx
            if inside:
                leaders.add(tgt)
            if nxt is not None:
                leaders.add(nxt)
            uncond = op in ("jmp", "jmpq")
            term[i] = ([tgt] if inside else [], not uncond)
        elif op.startswith("ret") or op in ("ud2", "hlt", "int3"):
            if nxt is not None:
                leaders.add(nxt)
            term[i] = ([], False)
        elif op.startswith("call") and NORETURN.search(arg):
            if nxt is not None:
                leaders.add(nxt)
            term[i] = ([], False)
    starts = sorted(leaders)
    bidx = {a: k for k, a in enumerate(starts)}
    n = len(starts)
    succ = [[] for _ in range(n)]
    for k, a in enumerate(starts):
        end = addr_idx[starts[k + 1]] if k + 1 < n else len(insns)
        last = end - 1
        if last in term:
            tg, fall = term[last]
            for t in tg:
                succ[k].append(bidx[t])
            if fall and k + 1 < n:
                succ[k].append(k + 1)
        elif k + 1 < n:
            succ[k].append(k + 1)
    # depth-first search from the entry: post-order, back edges by the stack (an edge to a block that is still open); for a reducible
    # graph those are exactly the edges to a dominator
    state = [0] * n                # 0 new, 1 open, 2 done
    post = []
    back = []
    stack = [(0, 0)]
    state[0] = 1
    while stack:
        v, i = stack[-1]
        if i < len(succ[v]):
            stack[-1] = (v, i + 1)
            w = succ[v][i]
            if state[w] == 0:
                state[w] = 1
                stack.append((w, 0))
            elif state[w] == 1:
                back.append((v, w))
        else:
            state[v] = 2
            post.append(v)
            stack.pop()
    reached = [st == 2 for st in state]
    # natural loops: the blocks that reach the back edge's source without passing the header
    pred = [[] for _ in range(n)]
    for v in range(n):
        for w in succ[v]:
            pred[w].append(v)
    loops = {}                     # header -> set of blocks
    for u, h in back:
        body = loops.setdefault(h, {h})
        work = [u]
        while work:
            x = work.pop()
            if x in body:
                continue
            body.add(x)
            work.extend(pred[x])
    in_loops = [[] for _ in range(n)]
    for h, body in loops.items():
        for v in body:
            in_loops[v].append((len(body), h))
    for v in range(n):
        in_loops[v].sort(key=lambda t: (-t[0], t[1]))          # outermost first
    back_set = set(back)
    # The order: reverse post-order of the graph without back edges in which EVERY LOOP IS CONTIGUOUS (all of a loop's blocks before
    # anything behind its exits): region by region -- the function, then every loop -- with the region's inner loops collapsed to
    # single nodes, and each collapsed loop expanded in place.
    FUNC = -1
    def chain(v):
        return [h for _, h in in_loops[v]]
    children = {FUNC: []}
    parent = {}
    for h in loops:
        c = chain(h)               # the loops around the header, itself last
        par = c[-2] if len(c) >= 2 else FUNC
        parent[h] = par
        children.setdefault(par, []).append(h)
        children.setdefault(h, [])
    def node_in(region, v):
        """the node of `region` (a loop header or FUNC) that block v belongs to: v itself, or the child loop around it; None = outside"""
        c = chain(v)
        if region == FUNC:
            return ("L", c[0]) if c else ("B", v)
        if region not in c:
            return None
        i = c.index(region)
        return ("L", c[i + 1]) if i + 1 < len(c) else ("B", v)
    def order_region(region, entry):
        members = list(loops[region]) if region != FUNC else list(range(n))
        nsucc = {}
        for v in members:
            nv = node_in(region, v)
            for w in succ[v]:
                if (v, w) in back_set:
                    continue
                nw = node_in(region, w)
                if nw is None or nw == nv:
                    continue
                nsucc.setdefault(nv, [])
                if nw not in nsucc[nv]:
                    nsucc[nv].append(nw)
        seen, post2 = set(), []
        def dfs(root):
            st = [(root, 0)]
            seen.add(root)
            while st:
                x, i = st[-1]
                ss = nsucc.get(x, [])
                if i < len(ss):
                    st[-1] = (x, i + 1)
                    y = ss[i]
                    if y not in seen:
                        seen.add(y)
                        st.append((y, 0))
                else:
                    post2.append(x)
                    st.pop()
        dfs(node_in(region, entry))
        seq = list(reversed(post2))
        rest = []
        for v in sorted(members):  # nodes the entry does not reach (landing pads, irreducible entries): behind everything, by address
            nv = node_in(region, v)
            if nv not in seen:
                seen.add(nv)
                rest.append(nv)
        out_blocks = []
        for kind, x in seq + rest:
            if kind == "B":
                out_blocks.append(x)
            else:
                out_blocks.extend(order_region(x, x))
        return out_blocks
    sys.setrecursionlimit(10000)
    seq = order_region(FUNC, 0)
    rpo = [0] * n
    for r, v in enumerate(seq):
        rpo[v] = r
    out = []
    for k, a in enumerate(starts):
        out.append((a, rpo[k], 1 if k in loops else 0, [starts[h] for h in chain(k)]))
    return out, len(loops)


def main():
    objdump, lib, out = sys.argv[1:4]
    p = subprocess.run([objdump, "-d", "--no-show-raw-insn", lib], stdout=subprocess.PIPE, check=True, text=True)
    fn_re = re.compile(r"^([0-9a-f]{8,16}) <(.+)>:$")
    ins_re = re.compile(r"^\s*([0-9a-f]+):\s+(\S+)\s*(.*)$")
    funcs = []
    cur = None
    for line in p.stdout.split("\n"):
        m = fn_re.match(line)
        if m:
            cur = (m.group(2), [])
            funcs.append(cur)
            continue
        if cur is None:
            continue
        m = ins_re.match(line)
        if m:
            cur[1].append((int(m.group(1), 16), m.group(2), m.group(3)))
    n_loops = n_blocks = 0
    loop_ids = {}
    with open(out, "w") as fh:
        for name, insns in funcs:
            if not insns:
                continue
            blocks, nl = analyse(insns)
            n_loops += nl
            n_blocks += len(blocks)
            fh.write("F %x %x %s\n" % (insns[0][0], insns[-1][0], name))
            for a, r, hdr, chain in blocks:
                ids = [str(loop_ids.setdefault(h, len(loop_ids))) for h in chain]
                fh.write("B %x %d %d%s\n" % (a, r, hdr, "".join(" " + i for i in ids)))
    print("mkloops: %d functions, %d blocks, %d loops -> %s" % (len(funcs), n_blocks, n_loops, out))


if __name__ == "__main__":
    main()
