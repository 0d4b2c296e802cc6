#!/usr/bin/env python3
"""bench.py -- Mbases piled / s of the MI355X mpileup/depth engine (BASELINE.json metric).

One "step" = one pass of the hot path over one window of synthetic, position-sorted reads that is
already resident in HBM: read filters -> quality prep -> BAQ -> overlap -> per-column measure ->
scan -> pileup text, all through the C-ABI (include/samtools_amd.h).  At N > 1 every rank owns a
different window (reference positions shard into independent windows: weak scaling) and the
per-window text is gathered on rank 0 with one RCCL gather inside the timed step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mpileup30|mpileup30_B|mpileup300|depth30|glf30|calmd30]

Launched by the driver for N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
Prints ONE JSON line on rank 0.  The CPU oracle appears only in the cpu_baseline leg (rank 0, N=1).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

WORKLOADS = {
    # name: (kind, depth, default window columns, algorithmic bytes per piled base, cli args)
    "mpileup30": ("mpileup", 30, 4 << 20, 4.3, ["mpileup", "-f", "{fa}", "{sam}"]),
    "mpileup30_B": ("mpileup", 30, 4 << 20, 4.3, ["mpileup", "-B", "-f", "{fa}", "{sam}"]),
    "mpileup300": ("mpileup", 300, 1 << 19, 3.75, ["mpileup", "-f", "{fa}", "{sam}"]),
    "depth30": ("depth", 30, 8 << 20, 0.21, ["depth", "-a", "{sam}"]),
    # rows widened into after the pileup path; single GPU, results stay on the device
    "glf30": ("glf", 30, 4 << 20, 1.5 + 128.0 / 30.0, ["glf", "-f", "{fa}", "{sam}"]),
    "calmd30": ("calmd", 30, 4 << 20, 4.0, ["calmd", "-r", "{sam}", "{fa}"]),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E


def alg_bytes_per_base(kernel, wl_bytes_per_base, band_cells=15):
    """Algorithmic HBM bytes per piled base for the kernels that can dominate a step.

    BAQ forward/backward: the (M, I) part of every forward row -- 2 * band_cells doubles per query base -- is written once
    by the forward kernel and read once by the backward/MAP kernel; inputs (quality, packed base, reference base) add ~2.5 B.
    Pileup kernels: SURVEY.md 8d figure for the whole measure+emit pair (4.3 B/base at 30x), attributed to the emit kernel."""
    if kernel in ("baq_fwd", "baq_bwd"):
        return 2 * band_cells * 8 + 2.5
    if kernel == "mplp_len":
        return 1.0 + 12.0 / 30.0
    return wl_bytes_per_base


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="mpileup30", choices=sorted(WORKLOADS))
    ap.add_argument("--cols", type=int, default=0, help="window columns per GPU per step (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cols", type=int, default=0)
    return ap.parse_args()


def build_window(torch, np, sa, rd, ref_t, n_cols, dev):
    """numpy SoA -> device tensors -> sta_window (STA_MEM_DEVICE)."""
    keep = {}
    def up(name):
        arr = rd[name]
        if arr.dtype == np.uint32: arr = arr.view(np.int32)
        elif arr.dtype == np.uint16: arr = arr.view(np.int16)
        elif arr.dtype == np.uint64: arr = arr.view(np.int64)
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        keep[name] = t
        return t.data_ptr()
    reads = sa.Reads()
    reads.n_reads = rd["n"]
    for f in ("pos", "flag", "mapq", "aux", "l_qseq", "cig_off", "base_off8", "mtid", "mpos", "isize", "name_off",
              "cigar", "seq", "qual", "names"):
        setattr(reads, f, up(f))
    reads.bq = None
    reads.n_cigar_total = len(rd["cigar"])
    reads.n_bases_total = len(rd["qual"])
    reads.n_name_bytes = len(rd["names"])
    files = (sa.Reads * 1)(reads)
    w = sa.Window()
    w.tid = 0; w.origin = 0; w.col_beg = 0; w.col_end = n_cols
    w.tname = b"chrS"; w.tlen = n_cols
    w.n_files = 1; w.files = files; w.mem = 1
    w.has_bed = 0; w.has_reg = 0
    keep["files"] = files
    in_bytes = sum(int(rd[f].nbytes) for f in ("pos", "flag", "mapq", "aux", "l_qseq", "cig_off", "base_off8", "mtid",
                                                 "mpos", "isize", "name_off", "cigar", "seq", "qual")) + n_cols
    return w, keep, in_bytes


def cpu_baseline(wl, sample_cols):
    """Oracle (plain-C restatement, kind='port') timed single-threaded on a bounded sample of the same workload."""
    from synth import synth_ref, synth_reads, write_sam, write_fasta
    kind, depth, _, _, argv = WORKLOADS[wl]
    oracle = os.path.join(REPO, "oracle", "_build", "oracle_samtools")
    if not os.path.exists(oracle):
        return None
    ref = synth_ref(sample_cols, seed=1)
    rd = synth_reads(ref, depth=depth, read_len=150, seed=42)
    with tempfile.TemporaryDirectory() as tmp:
        sam, fa = os.path.join(tmp, "s.sam"), os.path.join(tmp, "s.fa")
        write_sam(sam, rd, "chrS", sample_cols)
        write_fasta(fa, "chrS", ref)
        args = [a.format(sam=sam, fa=fa) for a in argv]
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as dn:
            subprocess.run([oracle] + args, stdout=dn, stderr=dn, check=True)
        dt = time.perf_counter() - t0
    bases = int(rd["n"]) * 150
    return {"value": bases / dt / 1e6, "unit": "Mbases/s", "cores": 1, "kind": "port",
            "sample": "%s on %d synthetic reads (%d Mbases, %d columns, SAM text input, output to /dev/null), %.1f s wall, "
                      "oracle restatement (not the upstream binary: HTSlib is absent)" % (
                          " ".join(x for x in argv if x != "{sam}").replace("{fa}", "ref.fa"), rd["n"], bases // 1000000, sample_cols, dt)}


def main():
    a = parse()
    import numpy as np
    import torch
    import samtools_amd as sa
    from synth import synth_ref, synth_reads

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    kind, depth, def_cols, alg_bytes_per_base_wl, _ = WORKLOADS[a.workload]
    n_cols = a.cols or def_cols
    # every rank generates its own window (different seed): reference windows are independent shards
    ref = synth_ref(n_cols, seed=1 + rank)
    rd = synth_reads(ref, depth=depth, read_len=150, seed=42 + rank)
    stream = torch.cuda.current_stream().cuda_stream
    eng = sa.Engine(local, stream)
    ref_t = torch.from_numpy(ref.copy()).to(dev)
    eng.set_reference(0, ref_t.data_ptr(), n_cols, 1)
    w, keep, in_bytes = build_window(torch, np, sa, rd, ref_t, n_cols, dev)
    if kind == "mpileup":
        par = sa.MplpParams.defaults()
        par.has_fai = 1
        if a.workload.endswith("_B"):
            par.flag &= ~sa.MPLP.REALN
    elif kind == "depth":
        par = sa.DepthParams.defaults()
        par.all_pos = 1
    elif world > 1:
        raise SystemExit("workload %s is a single-GPU measurement" % a.workload)

    def plan():
        eng.stage_window(w)
        if kind == "glf":
            return eng.glf_plan()
        if kind == "calmd":
            return eng.calmd_plan(flag=1)          # -r: BAQ (plain mode) + tag, then MD / NM
        return eng.mpileup_plan(par) if kind == "mpileup" else eng.depth_plan(par)

    info = plan()
    out_bytes = int(info.out_bytes)
    piled = int(info.piled_bases) or int(rd["n"]) * 150      # (the calmd plan reports no pileup counters: every base is aligned)
    from samtools_amd import shard
    sizes = recv = None
    cap = out_bytes + 4096
    if dist is not None:
        sizes = shard.exchange_sizes(out_bytes, dev)      # once: the synthetic window is the same every step
        cap = max(sizes) + 4096                            # every rank's buffers can be sent padded to the largest piece
    out_t = torch.empty(cap, dtype=torch.uint8, device=dev)

    # N > 1: the text of step k is gathered on rank 0 (ONE RCCL gather per step) while step k+1 computes: two output
    # buffers, asynchronous gather on RCCL's own stream, sizes exchanged once (the synthetic window is the same every step)
    n_buf = 2 if dist is not None else 1
    out_bufs = [out_t] + [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n_buf - 1)]
    pending = [None] * n_buf
    if dist is not None and rank == 0:
        recv = [[torch.empty(max(sizes), dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(n_buf)]
    step_no = [0]

    def step():
        i = step_no[0] % n_buf
        step_no[0] += 1
        if pending[i] is not None:
            pending[i].wait()              # the buffer's previous gather must have left it
            pending[i] = None
        plan()
        if kind == "mpileup":
            eng.mpileup_emit(out_bufs[i].data_ptr(), cap)
        elif kind == "depth":
            eng.depth_emit(out_bufs[i].data_ptr(), cap)
        if dist is not None:
            # the single collective of the path: per-window column text -> rank 0 over RCCL/xGMI (samtools_amd/shard.py)
            pending[i] = shard.gather_text(out_bufs[i], dst=0, sizes=sizes, recv=recv[i] if recv else None, async_op=True)

    def drain():
        for i in range(n_buf):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    eng.profile(True)
    eng.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                # every gather of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_get()
    eng.profile(False)

    tt = torch.tensor([dt, float(piled)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, piled_all = float(tmax[0].item()), float(tsum[1].item())
    else:
        dt_all, piled_all = dt, float(piled)

    if rank == 0:
        # correctness spot check of what was just timed: line count and final newline
        if kind in ("mpileup", "depth"):
            head = bytes(out_t[:min(out_bytes, 1 << 16)].cpu().numpy().tobytes())
            assert out_bytes == 0 or head.count(b"\n") > 0
        value = piled_all * a.steps / dt_all / 1e6
        # dominant kernel by accumulated HIP-event time (rank 0)
        def roof(name):
            launches, ms = prof[name]
            per_step = max(1, launches // max(1, a.steps))
            avg_ms = ms / max(1, launches)
            bpb = alg_bytes_per_base(name, alg_bytes_per_base_wl)
            per_launch = bpb * piled / per_step
            ach = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic_of(name, per_step), "alg_bytes_per_unit": bpb, "units_per_launch": piled / per_step,
                    "avg_launch_ms": avg_ms, "launches_per_step": per_step}

        pmc = {}
        pmc_path = os.path.join(REPO, "profiles", "r01_%s_pmc_traffic.json" % a.workload)
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
        KNAME = {"baq_fwd": "void k_baq_fwd<7>", "baq_bwd": "void k_baq_bwd<7>", "mplp_emit": "k_mplp_emit_fast", "mplp_len": "k_mplp_len_fast",
                 "depth_emit": "k_depth_emit", "depth_len": "k_depth_len", "depth_count": "k_depth_count"}

        def traffic_of(name, per_step):
            """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE + WRITE_SIZE, KB);
            measured at the default window size only (null otherwise).  Counter calibration: DESIGN.md section 5."""
            ent = pmc.get(KNAME.get(name, name))
            if not ent or a.cols or "FETCH_SIZE" not in ent or "WRITE_SIZE" not in ent:
                return None
            # gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a 16-byte-per-lane streaming read at
            # half its bytes; k_baq_bwd reads its forward rows that way (its ~30.2 GB of rows show up as ~16.6 GB)
            fetch_corr = 2.0 if name == "baq_bwd" else 1.0
            return (ent["FETCH_SIZE"]["per_launch"] * fetch_corr + ent["WRITE_SIZE"]["per_launch"]) * 1024.0

        # kernels on the side stream (band-8 BAQ groups) overlap the main ones: they cannot be "the" dominant kernel
        main = {k: v for k, v in prof.items() if not k.startswith("baq8")}
        dom_name = max(main.items(), key=lambda kv: kv[1][1])[0] if main else None
        res = {
            "metric": "Mbases piled/s (mpileup, 30x 150bp)" if a.workload.startswith("mpileup30") else "Mbases piled/s (%s)" % a.workload,
            "value": value, "unit": "Mbases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_all / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/f64" if ((kind == "mpileup" and not a.workload.endswith("_B")) or kind in ("glf", "calmd")) else "u8",
            "data": "synthetic",
            "config": {"workload": a.workload, "command": " ".join(x for x in WORKLOADS[a.workload][4] if x != "{sam}").replace("{fa}", "ref.fa"),
                       "read_len": 150, "depth": depth, "window_cols_per_gpu": n_cols, "reads_per_gpu": int(rd["n"]),
                       "piled_bases_per_gpu_step": piled, "out_bytes_per_gpu_step": out_bytes,
                       "staged_in_bytes_per_gpu": in_bytes, "parallelism": "window-sharded x%d, 1 RCCL gather" % world},
            "roofline": roof(dom_name) if dom_name else None,
            "kernels_ms_per_step": {k: v[1] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        }
        emit_name = {"mpileup": "mplp_emit", "depth": "depth_emit", "glf": "glf_cols", "calmd": "md_emit"}[kind]
        if emit_name in prof and emit_name != dom_name:
            res["roofline_pileup"] = roof(emit_name)
        # whole-step algorithmic rate (every kernel of the step, SURVEY.md 8d bytes): the number to compare with 8 TB/s end to end
        res["step_alg_GBps"] = alg_bytes_per_base_wl * piled_all / (dt_all / a.steps) / 1e9 / max(1, world)
        if world == 1 and not a.no_cpu_baseline:
            # bounded sample: BAQ runs at ~6 Mbases/s on one core, so 2 M columns (400 k reads, 60 Mbases) is ~10 s of CPU work;
            # the other workloads keep the same sample (the oracle needs 0.2-0.7 s there: generating and writing the SAM text
            # in Python costs far more than the run, so a bigger sample would only slow the bench down)
            sample = 2000000
            sample = a.cpu_sample_cols or sample
            if depth >= 300:
                sample //= 10
            cb = cpu_baseline(a.workload, sample)
            if cb:
                res["cpu_baseline"] = cb
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
