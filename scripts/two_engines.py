#!/usr/bin/env python3
"""Experiment (round 5): N engines, each on its own stream and host thread, stepping the SAME bench window concurrently -- what the drivers'
STA_DEV_THREADS=N does with consecutive windows (driver_pipeline.h DevEngines).  Prints ms per step for N = 1, 2, 3.
    python scripts/two_engines.py [workload] [steps]"""
import os, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import samtools_amd as sa
import bench
from synth import synth_ref

wl = sys.argv[1] if len(sys.argv) > 1 else "mpileup30"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
spec = bench.WORKLOADS[wl]
n_cols = spec["cols"]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ref = synth_ref(n_cols, seed=1)
rd = bench.make_reads(wl, ref, n_cols, 42, chunks=None)
ref_t = torch.from_numpy(ref.copy()).to(dev)
w, keep, in_bytes = bench.build_window(torch, np, sa, rd, n_cols, dev, origin=0, col_beg=0, col_end=n_cols, tlen=n_cols, star_tags=spec["n_tags"])
par = sa.MplpParams.defaults(); par.has_fai = 1
par.flag = (par.flag | spec["flags_on"]) & ~spec["flags_off"]
if spec["max_depth"]: par.max_depth = spec["max_depth"]
par.n_tags = spec["n_tags"]

def make(n):
    out = []
    for k in range(n):
        st = torch.cuda.Stream()
        e = sa.Engine(0, st.cuda_stream)
        e.set_reference(0, ref_t.data_ptr(), n_cols, 1)
        e.stage_window(w)
        info = e.mpileup_plan(par)
        cap = int(info.out_bytes) + 4096
        out.append((e, st, torch.empty(cap, dtype=torch.uint8, device=dev), cap, int(info.piled_bases)))
    return out

def run(engs, steps):
    def loop(k):
        e, st, buf, cap, _ = engs[k]
        for i in range(k, steps, len(engs)):
            e.stage_window(w); e.mpileup_run(par, buf.data_ptr(), cap)
        e.sync()
    th = [threading.Thread(target=loop, args=(k,)) for k in range(len(engs))]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

import hashlib
for n in (1, 2, 3, 1, 2):
    engs = make(n)
    run(engs, 2 * n)
    ms = [run(engs, steps) for _ in range(2)]
    piled = engs[0][4]
    hs = {hashlib.sha256(bytes(e[2][: e[3] - 4096].cpu().numpy())).hexdigest()[:12] for e in engs}
    print("engines", n, "ms/step", [round(x, 3) for x in ms], "Mbases/s", round(piled / min(ms) / 1e3), "text", hs, flush=True)
    del engs
