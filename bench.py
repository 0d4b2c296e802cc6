#!/usr/bin/env python3
"""bench.py -- Mbases piled / s of the MI355X mpileup/depth engine (BASELINE.json metric).

One "step" = one pass of the hot path over one window of synthetic, position-sorted reads that is
already resident in HBM: read filters -> quality prep -> BAQ -> overlap -> per-column measure ->
scan -> pileup text, all through the C-ABI (include/samtools_amd.h).

N > 1 (launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`):
ONE sorted input is generated identically on every rank, its reference columns are split into N contiguous
blocks, every rank stages the reads that can touch its block plus the mate halo and piles ONLY
its own columns, and the per-block text is collected on rank 0 with one size all-gather + one variable-size
gather over RCCL (samtools_amd/shard.py) inside the timed step.  Per-GPU work is fixed as N grows (the input
has N x the single-GPU window): weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mpileup30|mpileup30_B|mpileup300|mpileup100|mpileup30_EA_pairs|
                    mpileup30_hotspot|mpileup30_indel|mpileup30_trim|depth30|glf30|calmd30|consensus30 ...]
                    [--verify] [--no-pmc] [--no-cpu-baseline]

Prints ONE JSON line on rank 0.  The CPU oracle appears only as the checker / cpu_baseline leg (rank 0).
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

def _pile_bpb(depth):
    """algorithmic HBM bytes per piled base of the pileup text path: seq 0.5 + qual 1 + record header / CIGAR /
    offsets 0.19 + 2 output characters + (reference base 1 + fixed row text 17) / depth  ->  4.29 @30x, 3.87 @100x, 3.75 @300x"""
    return 3.69 + 18.0 / depth


def _wl(kind, depth, cols, argv, baq=False, bpb=None, gen=None, flags_on=0, flags_off=0, max_depth=None, files=1, n_tags=0):
    """one workload: kind, depth, default window columns per GPU, CLI form (what the oracle runs), generator options,
    sta_mplp_params deltas.  bpb = algorithmic bytes per piled base of the WHOLE step (pileup text path + ~1 B/base of reference
    window when BAQ runs); bpb_pileup = the text path alone (what the emit kernel is priced with)."""
    pile = _pile_bpb(depth)
    return {"kind": kind, "depth": depth, "cols": cols, "argv": argv, "baq": baq, "bpb": bpb if bpb is not None else pile + (1.0 if baq else 0.0),
            "bpb_pileup": pile if kind == "mpileup" else (bpb if bpb is not None else pile), "gen": gen or {}, "flags_on": flags_on, "flags_off": flags_off,
            "max_depth": max_depth, "files": files, "n_tags": n_tags}


_REALN, _REDO_BAQ, _NO_ORPHAN = 1 << 4, 1 << 6, 1 << 3      # STA_MPLP_* (include/samtools_amd.h)
_PRINT_MAPQ_CHAR, _PRINT_QPOS, _PRINT_QNAME = 1 << 11, 1 << 12, 1 << 13
WORKLOADS = {
    # BASELINE.json configs[2] (the metric's configuration): mpileup -f, BAQ on, 30x 150 bp
    # (the headline: bench.py steps a window of 16 M columns -- 503 Mbases, 3.36 M reads, 1.28 GB of text, ~2.5 GB resident with the inputs; the
    #  parity tests of tests/test_gpu_benchsize_parity.py use `cols` and, for this workload, the stepped window as well.  The class-S BAQ kernel
    #  is persistent: its tail and the latency-bound list kernels beside it are per window, and at 4 M columns they were 12 % of the step:
    #  profiles/r06_sessionZ_window_size.log)
    "mpileup30": dict(_wl("mpileup", 30, 4 << 20, ["mpileup", "-f", "{fa}", "{sam}"], baq=True), bench_cols=16 << 20),
    "mpileup30_B": dict(_wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-f", "{fa}", "{sam}"], flags_off=_REALN), bench_cols=16 << 20),
    # configs[3] shape (deep columns); --gpus N shards it like mpileup30
    "mpileup300": dict(_wl("mpileup", 300, 1 << 19, ["mpileup", "-f", "{fa}", "{sam}"], baq=True), bench_cols=1 << 21),
    "mpileup300_B": dict(_wl("mpileup", 300, 1 << 19, ["mpileup", "-B", "-f", "{fa}", "{sam}"], flags_off=_REALN), bench_cols=1 << 21),
    # between the two emit kernels' home grounds
    "mpileup100": dict(_wl("mpileup", 100, 1 << 20, ["mpileup", "-f", "{fa}", "{sam}"], baq=True), bench_cols=4 << 20),
    "mpileup100_B": dict(_wl("mpileup", 100, 1 << 20, ["mpileup", "-B", "-f", "{fa}", "{sam}"], flags_off=_REALN), bench_cols=4 << 20),
    # configs[4]: -E -A, BAQ recomputed + mate-overlap resolution on 30x PAIRED reads (99/147 + 83/163, insert ~N(300,30): about a
    # third of the pairs overlap); + 2 B per overlapping base of quality read-modify-write
    "mpileup30_EA_pairs": _wl("mpileup", 30, 4 << 20, ["mpileup", "-E", "-A", "-f", "{fa}", "{sam}"], baq=True, gen={"paired": True},
                              flags_on=_REDO_BAQ, flags_off=_NO_ORPHAN),
    "mpileup30_B_pairs": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-A", "-f", "{fa}", "{sam}"], gen={"paired": True}, flags_off=_REALN | _NO_ORPHAN),
    # the deep-amplicon shape inside an ordinary window: 30x + one 10 000x amplicon of 300 bp (-d raised well above the 20 000 reads that start within one read length: the cap's conservative detector stays quiet and the exact serial replay, 0.7 s here, is not what is measured)
    "mpileup30_hotspot": _wl("mpileup", 30, 4 << 20, ["mpileup", "-d", "100000", "-f", "{fa}", "{sam}"], baq=True, gen={"hotspot": (300, 10000)}, max_depth=100000),
    "mpileup30_B_hotspot": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-d", "100000", "-f", "{fa}", "{sam}"], gen={"hotspot": (300, 10000)}, flags_off=_REALN, max_depth=100000),
    # BAQ with a real indel spectrum: 5 % of the reads carry a 1-3 bp insertion or deletion (band-8 / general-band kernels under load)
    "mpileup30_indel": dict(_wl("mpileup", 30, 4 << 20, ["mpileup", "-f", "{fa}", "{sam}"], baq=True, gen={"indel_rate": 0.05}), bench_cols=16 << 20),
    # trimmed reads: half of the reads lose 1..50 bases at one end, i.e. fifty-one read lengths side by side (adapter / quality trimming of
    # real data).  Round 4's class-S grouping (64 consecutive reads of ONE length) sent nearly all of them through the list kernels; the
    # per-length dense groups of round 5 keep them in k_baq7s
    "mpileup30_trim": dict(_wl("mpileup", 30, 4 << 20, ["mpileup", "-f", "{fa}", "{sam}"], baq=True, gen={"trim_rate": 0.5}), bench_cols=16 << 20),
    # three input files, 10x each (the shape of test/dat/mpileup.out.1): the per-file column groups of bam_plcmd.c:669-857 at bench size
    "mpileup30_3files": _wl("mpileup", 30, 4 << 20, ["mpileup", "-f", "{fa}", "{sam}"], baq=True, files=3),
    "mpileup30_B_3files": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-f", "{fa}", "{sam}"], flags_off=_REALN, files=3),
    # the generic column walkers (k_mplp_len / k_mplp_emit): -s -O and --output-extra columns (bam_plcmd.c:727-855); the generated
    # reads carry no NM tag: that column is "*"
    "mpileup30_B_sOx": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-s", "-O", "--output-extra", "QNAME,NM", "-f", "{fa}", "{sam}"],
                           flags_on=_PRINT_MAPQ_CHAR | _PRINT_QPOS | _PRINT_QNAME, flags_off=_REALN, n_tags=1),
    # -s alone: the mapping-quality column rides in the tile kernels (round 4); three files and the 300x shape (read-major kernel) as well
    "mpileup30_B_s": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-s", "-f", "{fa}", "{sam}"], flags_on=_PRINT_MAPQ_CHAR, flags_off=_REALN),
    "mpileup300_B_s": _wl("mpileup", 300, 1 << 19, ["mpileup", "-B", "-s", "-f", "{fa}", "{sam}"], flags_on=_PRINT_MAPQ_CHAR, flags_off=_REALN),
    "mpileup30_B_s_3files": _wl("mpileup", 30, 4 << 20, ["mpileup", "-B", "-s", "-f", "{fa}", "{sam}"], flags_on=_PRINT_MAPQ_CHAR, flags_off=_REALN, files=3),
    # configs[1]
    "depth30": dict(_wl("depth", 30, 8 << 20, ["depth", "-a", "{sam}"], bpb=0.21), bench_cols=32 << 20),
    # rows widened into after the pileup path; single GPU, results stay on the device
    "glf30": _wl("glf", 30, 4 << 20, ["glf", "-f", "{fa}", "{sam}"], bpb=1.5 + 128.0 / 30.0),
    "calmd30": _wl("calmd", 30, 4 << 20, ["calmd", "-r", "{sam}", "{fa}"], bpb=4.0),
    # `samtools consensus` columns (call + quality per column stay on the device).  Algorithmic bytes per piled
    # base: seq 0.5 + qual 1 + per-read header ~0.2 + per-column result 12 B / depth 30 = 0.4
    "consensus30": _wl("consensus", 30, 4 << 20, ["consensus", "-f", "fastq", "{sam}"], bpb=2.1),
    "consensus30_simple": _wl("consensus", 30, 4 << 20, ["consensus", "-m", "simple", "-f", "fastq", "{sam}"], bpb=2.1),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec); ~6.3 TB/s is what a streaming copy reaches
# fp64 vector ALU without fused multiply-add (BAQ must round like the CPU: -ffp-contract=off): 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz
FP64_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# BAQ arithmetic per band cell (kernels_baq.hip): forward M 6 + I 4 + D 3 + row sum 3 + scaling 3 = 19 fp64 operations,
# backward M 6 + I 3 + D 4 + scaling 2 + MAP 6 = 21; 2*7+1 = 15 band cells per query base
BAQ_FP64_OPS_PER_BASE = {"baq_fwd": 19 * 15, "baq_bwd": 21 * 15, "baq_s": (19 + 21) * 15}     # baq_s: both passes in one launch (class S, baq_band7s.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node; without WORLD_SIZE in the environment bench.py launches itself under "
                         "torch.distributed.run with that many processes (default 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default mpileup30 (the metric's configuration); at N > 1 the default run also measures mpileup300 "
                         "(BASELINE.json configs[3], the north star's 8-GPU shape) and reports it inside the same JSON line")
    ap.add_argument("--no-pair", action="store_true", help="N > 1: do not add the mpileup300 measurement to the default run")
    ap.add_argument("--cols", type=int, default=0, help="window columns per GPU per step (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cols", type=int, default=0)
    ap.add_argument("--verify", action="store_true",
                    help="hash the text of the timed window and compare it with the oracle's text for the same seeds (adds ~20-40 s of CPU work)")
    ap.add_argument("--no-e2e", action="store_true", help="do not time the file -> text CLI runs (the `e2e` object of the default run)")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run one step under rocprofv3 --pmc for roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def upload_reads(torch, np, sa, rd, dev, keep):
    """numpy SoA -> device tensors -> sta_reads descriptor (STA_MEM_DEVICE)."""
    def up(name):
        arr = rd[name]
        if arr.dtype == np.uint32: arr = arr.view(np.int32)
        elif arr.dtype == np.uint16: arr = arr.view(np.int16)
        elif arr.dtype == np.uint64: arr = arr.view(np.int64)
        t = torch.from_numpy(np.ascontiguousarray(arr).copy()).to(dev)
        keep.append(t)
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
    return reads


def build_window(torch, np, sa, rd, n_cols, dev, origin=0, col_beg=0, col_end=None, tlen=None, star_tags=0):
    """rd: one generated read set, or a list of them (one per input file).  star_tags: every read gets that many "*" text columns
    (aux-tag columns of --output-extra for reads that carry no such tag)."""
    rds = rd if isinstance(rd, (list, tuple)) else [rd]
    keep = []
    files = (sa.Reads * len(rds))(*[upload_reads(torch, np, sa, r, dev, keep) for r in rds])
    if star_tags:
        for f, r in enumerate(rds):
            n = int(r["n"]) * star_tags
            xo = torch.arange(n + 1, dtype=torch.int32, device=dev); xt = torch.full((n + 1,), ord("*"), dtype=torch.uint8, device=dev)
            keep += [xo, xt]
            files[f].n_xcols = star_tags; files[f].xcol_off = xo.data_ptr(); files[f].xcol_text = xt.data_ptr(); files[f].n_xcol_bytes = n
    w = sa.Window()
    w.tid = 0; w.origin = origin; w.col_beg = col_beg; w.col_end = n_cols if col_end is None else col_end
    w.tname = b"chrS"; w.tlen = n_cols if tlen is None else tlen
    w.n_files = len(rds); w.files = files; w.mem = 1
    w.has_bed = 0; w.has_reg = 0
    keep.append(files)
    in_bytes = sum(int(r[f].nbytes) for r in rds for f in ("pos", "flag", "mapq", "aux", "l_qseq", "cig_off", "base_off8", "mtid",
                                                          "mpos", "isize", "name_off", "cigar", "seq", "qual")) + (w.col_end - w.col_beg)
    return w, keep, in_bytes


def n_reads_of(rd):
    return sum(int(r["n"]) for r in rd) if isinstance(rd, (list, tuple)) else int(rd["n"])


def slice_reads(np, rd, lo, hi, origin):
    """reads [lo, hi) of a generated set as their own sta_reads arrays, positions relative to `origin` (a rank's block + halo)."""
    out = {"n": hi - lo, "L": rd["L"]}
    for f in ("flag", "mapq", "aux", "l_qseq", "mtid", "mpos", "isize"):
        out[f] = rd[f][lo:hi]
    out["pos"] = (rd["_abs_pos"][lo:hi] - origin).astype(np.int32)
    c0, c1 = int(rd["cig_off"][lo]), int(rd["cig_off"][hi])
    out["cig_off"] = (rd["cig_off"][lo:hi + 1] - c0).astype(np.uint32)
    out["cigar"] = rd["cigar"][c0:c1]
    b0 = int(rd["base_off8"][lo]) if hi > lo else 0
    b1 = (int(rd["base_off8"][hi]) if hi < rd["n"] else len(rd["qual"]) // 8) if hi > lo else 0
    out["base_off8"] = (rd["base_off8"][lo:hi] - b0).astype(np.uint32)
    out["qual"] = rd["qual"][b0 * 8:b1 * 8]
    out["seq"] = rd["seq"][b0 * 4:b1 * 4]
    n0, n1 = int(rd["name_off"][lo]), int(rd["name_off"][hi])
    out["name_off"] = (rd["name_off"][lo:hi + 1] - n0).astype(np.uint32)
    out["names"] = rd["names"][n0:n1]
    return out


def oracle_text_hash(wl, n_cols, seed_ref=1, seed_reads=42, save_to=None, inputs=None, chunk_cols=None):
    """The checker: the oracle's text for the synthetic window (same generator, same seeds) -> sha256, bytes, wall time.
    inputs: optional dict from synth_inputs() so that several workloads of one shape share the generated reads and SAM text."""
    argv = WORKLOADS[wl]["argv"]
    oracle = os.path.join(REPO, "oracle", "_build", "oracle_samtools")
    if not os.path.exists(oracle):
        return None
    own = inputs is None
    if own:
        inputs = synth_inputs(wl, n_cols, seed_ref, seed_reads, chunk_cols)
    try:
        rd = inputs["rd"]
        args = []
        for a_ in argv:
            if a_ == "{sam}": args += list(inputs["sams"])
            else: args.append(a_.format(fa=inputs["fa"]))
        h = hashlib.sha256()
        n = 0
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as dn:
            p = subprocess.Popen([oracle] + args, stdout=subprocess.PIPE, stderr=dn)
            fh = open(save_to, "wb") if save_to else None
            while True:
                b = p.stdout.read(1 << 22)
                if not b:
                    break
                h.update(b); n += len(b)
                if fh: fh.write(b)
            if fh: fh.close()
            if p.wait() != 0:
                raise RuntimeError("oracle failed on the %s sample" % wl)
        dt = time.perf_counter() - t0
        return {"sha256": h.hexdigest(), "bytes": n, "seconds": dt, "n_reads": n_reads_of(rd), "bases": n_reads_of(rd) * 150,
                "argv": " ".join(x for x in argv if x != "{sam}").replace("{fa}", "ref.fa"), "ref": inputs["ref"], "rd": rd}
    finally:
        if own:
            shutil.rmtree(inputs["dir"], ignore_errors=True)


def make_reads(wl, ref, chunk_cols, seed_reads=42, chunks=None):
    """the workload's synthetic reads over `ref` (SURVEY.md 8d generator, tests/synth.py): ONE definition for the engine's
    arrays, the oracle's SAM text and the parity tests"""
    from synth import synth_chunked, synth_reads, synth_hotspot
    spec = WORKLOADS[wl]
    g = spec["gen"]
    if g.get("paired"):
        rd = synth_reads(ref, depth=spec["depth"], read_len=150, seed=seed_reads, paired=True)
        rd["_abs_pos"] = rd["_abs_pos"].copy()
        return rd
    kw = {k: g[k] for k in ("indel_rate", "trim_rate", "trim_max") if k in g}
    # a window above the workload's parity-test size (`cols`; `bench_cols` is a multiple of it) is assembled from pieces of that size, built by
    # a few threads; `chunks` counts in units of chunk_cols (a rank's window)
    piece = spec["cols"]
    if chunk_cols > piece and chunk_cols % piece == 0:
        m = chunk_cols // piece
        if chunks is not None:
            # (a rank's own window = all of its pieces; of its neighbours' windows only the adjacent piece: reads reach a few hundred columns)
            cs = sorted(chunks)
            mid = cs[len(cs) // 2]
            chunks = [c * m + j for c in cs for j in (range(m) if c == mid else (m - 1,) if c < mid else (0,))]
        chunk_cols = piece
    kw["procs"] = max(1, min(4, (os.cpu_count() or 2) // 2))
    if spec["files"] > 1:
        # one read set per input file, depth / files each, its own seed (single-GPU workloads)
        return [synth_chunked(ref, chunk_cols, depth=spec["depth"] // spec["files"], read_len=150, seed=seed_reads + 1000 * k, chunks=chunks, **kw)
                for k in range(spec["files"])]
    rd = synth_chunked(ref, chunk_cols, depth=spec["depth"], read_len=150, seed=seed_reads, chunks=chunks, **kw)
    if g.get("hotspot"):
        hl, hd = g["hotspot"]
        rd = synth_hotspot(ref, rd, hot_start=len(ref) // 4, hot_len=hl, hot_depth=hd, seed=seed_reads + 1000)
    return rd


def synth_inputs(wl, n_cols, seed_ref=1, seed_reads=42, chunk_cols=None):
    """a workload's synthetic window as numpy arrays AND as the SAM / FASTA files the oracle reads (caller removes ['dir']).
    chunk_cols: the per-GPU window of a sharded run (the input is then assembled from pieces, tests/synth.py synth_chunked)."""
    from synth import synth_ref, write_sam, write_fasta
    ref = synth_ref(n_cols, seed=seed_ref)
    rd = make_reads(wl, ref, chunk_cols or n_cols, seed_reads)
    d = tempfile.mkdtemp(prefix="sta_bench_")
    fa = os.path.join(d, "s.fa")
    sams = []
    for k, r in enumerate(rd if isinstance(rd, list) else [rd]):
        sams.append(os.path.join(d, "s%d.sam" % k if k else "s.sam"))
        write_sam(sams[-1], r, "chrS", n_cols)
    write_fasta(fa, "chrS", ref)
    return {"ref": ref, "rd": rd, "sam": sams[0], "sams": sams, "fa": fa, "dir": d}


def e2e_file_to_text(inputs, sample_cols, oracle_sha_of_sample, oracle_text_path, copies=8):
    """File in, text out, through the product's command drivers: what a user of `samtools mpileup` / `depth` waits for.
    The CPU-baseline sample (SAM text of `sample_cols` columns at 30x) is replicated onto `copies` contigs (the same reads under
    another contig name: >= 0.25 Gbases in all), written as BAM (level 1) with the product's own writer, and `samtools-amd` is run on it
    as a subprocess with its text sent to /dev/null: process start, HIP context, BGZF inflate, staging, PCIe both ways and formatting
    are all inside the wall time.  The text of `mpileup -f` is also hashed once and compared with the oracle's text for the
    sample replicated the same way."""
    from samtools_amd import _capi
    d = inputs["dir"]
    exe = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
    if not os.path.exists(exe):
        return None
    names = ["chrS%d" % k for k in range(copies)]
    body = open(inputs["sam"]).read()
    lines_start = body.index("\n", body.index("@SQ")) + 1
    reads = body[lines_start:]
    big_sam = os.path.join(d, "e2e.sam")
    with open(big_sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, sample_cols) for nm in names))
        for nm in names:
            fh.write(reads.replace("\tchrS\t", "\t%s\t" % nm))
    n_reads = reads.count("\n")
    fa_txt = open(inputs["fa"]).read()
    big_fa = os.path.join(d, "e2e.fa")
    with open(big_fa, "w") as fh:
        for nm in names:
            fh.write(fa_txt.replace(">chrS\n", ">%s\n" % nm, 1))
    bam = os.path.join(d, "e2e.bam")
    _capi.io_write_bam(big_sam, bam, 1)
    os.remove(big_sam)
    mbases = n_reads * copies * 150 / 1e6
    out = {"input": "%d contigs x %d columns, 30x 150 bp: %d reads, %.0f Mbases, BAM level 1 (%.0f MB) + FASTA"
                    % (copies, sample_cols, n_reads * copies, mbases, os.path.getsize(bam) / 1e6),
           "includes": "process start, HIP context creation, BGZF inflate, staging, PCIe both ways, text written to /dev/null",
           "commands": {}}

    def timed(args, reps=3):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                p = subprocess.run([exe] + args, stdout=dn, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return None, p.stderr.decode()[-300:]
            best = dt if best is None else min(best, dt)
        return best, None

    # what a run costs before its first window: process start + HIP context + engine (the same binary on a one-read input)
    tiny = os.path.join(d, "e2e_tiny.sam")
    with open(tiny, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrS0\tLN:%d\n" % sample_cols + reads[:reads.index("\n") + 1].replace("\tchrS\t", "\tchrS0\t"))
    t_start, _ = timed(["depth", tiny], reps=3)
    out["startup_s"] = t_start
    for label, args in (("mpileup -f", ["mpileup", "-f", big_fa, bam]), ("mpileup -B -f", ["mpileup", "-B", "-f", big_fa, bam]), ("depth -a", ["depth", "-a", bam])):
        t, err = timed(args)
        out["commands"][label] = {"wall_s": t, "mbases_per_s": mbases / t if t else None,
                                  "mbases_per_s_net_of_startup": mbases / max(t - t_start, 1e-3) if (t and t_start) else None, "error": err}
    # parity of ALL THREE: the whole text against the oracle's text of the sample, replicated the same way (round 5: `mpileup -B -f` and
    # `depth -a` used to be timed but not checked).  The oracle runs once per command on the one-contig sample; its wall time is the
    # same-boundary CPU figure (text input, text output, one thread), scaled to the replicated input.
    def engine_sha(args):
        p = subprocess.Popen([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        got = hashlib.sha256()
        while True:
            b = p.stdout.read(1 << 22)
            if not b:
                break
            got.update(b)
        p.wait()
        return got.hexdigest() if p.returncode == 0 else None

    def replicated_sha(txt):
        want = hashlib.sha256()
        for nm in names:
            want.update(txt.replace(b"chrS\t", nm.encode() + b"\t"))
        return want.hexdigest()

    oracle_exe = os.path.join(REPO, "oracle", "_build", "oracle_samtools")
    checks = {}
    if oracle_text_path and os.path.exists(oracle_text_path):
        checks["mpileup -f"] = (replicated_sha(open(oracle_text_path, "rb").read()), ["mpileup", "-f", big_fa, bam], None)
    if os.path.exists(oracle_exe):
        for label, o_args, e_args in (("mpileup -B -f", ["mpileup", "-B", "-f", inputs["fa"], inputs["sam"]], ["mpileup", "-B", "-f", big_fa, bam]),
                                      ("depth -a", ["depth", "-a", inputs["sam"]], ["depth", "-a", bam])):
            t0 = time.perf_counter()
            pr = subprocess.run([oracle_exe] + o_args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            t_or = time.perf_counter() - t0
            if pr.returncode == 0:
                checks[label] = (replicated_sha(pr.stdout), e_args, t_or)
    out["identical_to_oracle"] = {}
    for label, (want, e_args, t_or) in checks.items():
        got = engine_sha(e_args)
        out["identical_to_oracle"][label] = bool(got is not None and got == want)
        if t_or and out["commands"].get(label, {}).get("wall_s"):
            # same boundary on both sides: file in, text out (the oracle reads the SAM text of ONE contig: x copies)
            out["commands"][label]["oracle_1_thread_mbases_per_s"] = mbases / copies / t_or
            out["commands"][label]["vs_oracle_same_boundary"] = out["commands"][label]["mbases_per_s"] / (mbases / copies / t_or)
    out["identical_to_oracle"]["all"] = bool(checks) and all(v for k, v in out["identical_to_oracle"].items())
    return out


def collect_pmc(a, wlname, kernels):
    """roofline.traffic measured on THIS box: one step of the same workload under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes: both do not fit the TCC slots; MI355X_MICROARCH.md 'rocprofv3 PMC slots').
    Returns {kernel: {"FETCH_SIZE": KB per launch, "WRITE_SIZE": KB per launch}} or None."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sta_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [rocprof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", wlname, "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--no-pmc"] + (["--cols", str(a.cols)] if a.cols else [])
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            agg = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    k = row["Kernel_Name"].split("(")[0]
                    e = agg.setdefault(k, [0, 0.0])
                    e[0] += 1; e[1] += float(row["Counter_Value"])
            for k, (n, v) in agg.items():
                out.setdefault(k, {})[ctr] = v / max(n, 1)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out or None


# engine kernel label -> prefix of the rocprofv3 kernel name
KNAME = {"baq_s": "void k_baq7s", "baq_fwd": "void k_baq_fwd<7", "baq_bwd": "void k_baq_bwd<7", "mplp_emit": "k_mplp_emit_tile", "mplp_emit_deep": "k_mplp_emit_deep", "mplp_len": "k_mplp_len_rm",
         "depth_fused": "k_depth_fused", "glf_cols": "k_glf_cols", "cons_col": "k_cons_col", "cons_walk": "k_cons_walk", "cons_read_a": "k_cons_read_a"}
# gfx950 calibration (scripts/ubench/pmc_calib.hip, profiles/r04_pmc_calibration.md): kernels that read exactly 4 GiB from HBM
# with 16 / 8 / 4 / 1 bytes per lane, temporal and non-temporal, all count FETCH_SIZE = 2 GiB (TCC_EA0_RDREQ = one request per 128 B
# of it, none of them 32 B), and kernels that write 4 GiB count WRITE_SIZE = 4 GiB: on this chip FETCH_SIZE is half the bytes
# fetched whatever the access width, WRITE_SIZE is exact.  The factor is applied to every kernel.
FETCH_SCALE = 2.0
WRITE_SCALE = 1.0


def self_launch(a):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: become N ranks.  One process per GPU under
    torch.distributed.run on 127.0.0.1 (the form the driver uses), same arguments; never returns."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, STA_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush(); sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus is None:
        a.gpus = int(env_world) if env_world else 1
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is None and a.gpus > 1 and not a.pmc_child:
        self_launch(a)
    if env_world is not None and int(env_world) != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (a.gpus, env_world))
    import numpy as np
    import torch
    import samtools_amd as sa

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for the 1-GPU box: STA_BENCH_ONE_DEVICE=1 puts every rank on device 0, STA_BENCH_BACKEND=gloo replaces RCCL
    if os.environ.get("STA_BENCH_ONE_DEVICE"):
        local = 0
    backend = os.environ.get("STA_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU fallback)")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs device %d but only %d are visible (one process per GPU; "
                         "STA_BENCH_ONE_DEVICE=1 is the one-GPU test hook)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = world == 1 and bool(os.environ.get("STA_BENCH_FORCE_DIST")) and not a.pmc_child
    if world > 1 or force_dist:
        import torch.distributed as dist
        kw = {}
        if force_dist and "MASTER_ADDR" not in os.environ:
            # test hook for the 1-GPU box: a process group of ONE rank, so that the RCCL branch of everything below (communicator set-up,
            # the size all-gather, reductions and barriers on device tensors) executes on real hardware; only the send / receive pair of
            # the text gather needs a second GPU
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            kw = {"init_method": "tcp://127.0.0.1:%d" % port, "world_size": 1, "rank": 0}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)
    ctx = {"np": np, "torch": torch, "sa": sa, "dist": dist, "rank": rank, "world": world, "local": local, "dev": dev, "backend": backend}
    primary = a.workload or "mpileup30"
    res = run_workload(a, primary, ctx)
    if world > 1 and a.workload is None and not a.no_pair and not a.pmc_child:
        # the north star's second number: the 300x deep-amplicon shape sharded the same way (BASELINE.json configs[3])
        second = run_workload(a, "mpileup300", ctx, secondary=True)
        if rank == 0 and res is not None and second is not None:
            res["mpileup300"] = {k: second[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype",
                                                         "config", "gather", "per_rank", "distributed", "kernels_ms_per_step", "output_sha256", "verify") if k in second}
    if rank == 0 and res is not None:
        print(json.dumps(res))
        bad = [r for r in (res, res.get("mpileup300") or {}) if (r.get("parity_check") and not r["parity_check"]["identical"]) or (r.get("verify") and not r["verify"]["identical"])]
        e2e_id = (res.get("e2e") or {}).get("identical_to_oracle")
        if bad or (isinstance(e2e_id, dict) and e2e_id.get("all") is False):
            raise SystemExit("bench.py: the engine's text differs from the oracle's")
    if dist is not None:
        dist.destroy_process_group()


def run_workload(a, wlname, ctx, secondary=False):
    """warm up, time K steps of one workload on this process group, return the result record on rank 0 (None elsewhere)"""
    np, torch, sa, dist = ctx["np"], ctx["torch"], ctx["sa"], ctx["dist"]
    rank, world, local, dev, backend = ctx["rank"], ctx["world"], ctx["local"], ctx["dev"], ctx["backend"]
    from synth import synth_ref

    spec = WORKLOADS[wlname]
    kind, depth, def_cols, alg_bpb = spec["kind"], spec["depth"], spec.get("bench_cols", spec["cols"]), spec["bpb"]
    if world > 1 and (spec["gen"].get("paired") or spec["gen"].get("hotspot") or spec["files"] > 1):
        raise SystemExit("workload %s is a single-GPU measurement (its generator is not built piecewise)" % wlname)
    cols_per_gpu = a.cols or def_cols
    n_cols = cols_per_gpu * world
    from samtools_amd import shard
    # ONE input for the whole job, the same on every rank: piece k = the reads starting in the k-th window of cols_per_gpu columns
    # (seed 42 + k; they reach into the next window).  Rank r owns columns [blk_beg, blk_end) and only builds the pieces around them.
    ref = synth_ref(n_cols, seed=1)
    rd_all = make_reads(wlname, ref, cols_per_gpu, 42, chunks=(rank - 1, rank, rank + 1) if world > 1 else None)
    blk_beg, blk_end = shard.block_of(rank, world, n_cols)
    if world > 1:
        # reads that can touch the block plus the mate halo (reads starting up to 2 x the longest span before it)
        lo, hi = shard.read_range(rd_all["_abs_pos"], blk_beg, blk_end, halo=shard.halo_columns(150 + 3))
        origin = max(0, blk_beg - shard.halo_columns(150 + 3))
        rd = slice_reads(np, rd_all, lo, hi, origin)
    else:
        origin, rd = 0, rd_all
    stream = torch.cuda.current_stream().cuda_stream
    eng = ctx.get("eng")
    if eng is None:
        eng = ctx["eng"] = sa.Engine(local, stream)
    ref_t = torch.from_numpy(ref.copy()).to(dev)
    eng.set_reference(0, ref_t.data_ptr(), n_cols, 1)
    w, keep, in_bytes = build_window(torch, np, sa, rd, n_cols, dev, origin=origin, col_beg=blk_beg - origin, col_end=blk_end - origin, tlen=n_cols,
                                     star_tags=spec["n_tags"])
    if kind == "mpileup":
        par = sa.MplpParams.defaults()
        par.has_fai = 1
        par.flag = (par.flag | spec["flags_on"]) & ~spec["flags_off"]
        if spec["max_depth"]:
            par.max_depth = spec["max_depth"]
        par.n_tags = spec["n_tags"]
    elif kind == "depth":
        par = sa.DepthParams.defaults()
        par.all_pos = 1
    elif world > 1:
        raise SystemExit("workload %s is a single-GPU measurement" % wlname)

    cons_par = None
    if kind == "consensus":
        cons_par = sa.ConsParams.defaults()
        if wlname.endswith("_simple"):
            cons_par.mode = 0
        else:   # the Bayesian mode reads MD:Z from text column 0; the generated reads carry no tag ("*")
            nrd = int(rd["n"])
            xo = torch.arange(nrd + 1, dtype=torch.int32, device=dev); xt = torch.full((nrd + 1,), ord("*"), dtype=torch.uint8, device=dev)
            keep += [xo, xt]
            w.files[0].n_xcols = 1; w.files[0].xcol_off = xo.data_ptr(); w.files[0].xcol_text = xt.data_ptr(); w.files[0].n_xcol_bytes = nrd

    class _ConsInfo:
        out_bytes = 0; piled_bases = 0

    def plan():
        eng.stage_window(w)
        if kind == "consensus":
            ci = eng.consensus_run(cons_par)
            r = _ConsInfo(); r.out_bytes = int(ci.n_cols) * 12; r.piled_bases = int(ci.n_entries)
            return r
        if kind == "glf":
            return eng.glf_plan()
        if kind == "calmd":
            return eng.calmd_plan(flag=1)          # -r: BAQ (plain mode) + tag, then MD / NM
        return eng.mpileup_plan(par) if kind == "mpileup" else eng.depth_plan(par)

    info = plan()
    out_bytes = int(info.out_bytes)
    piled = int(info.piled_bases) or n_reads_of(rd) * 150      # (the calmd plan reports no pileup counters: every base is aligned)
    # (piled_bases counts only the columns this rank owns: k_prep_reads clips every read to [col_beg, col_end))
    sizes = None
    cap = out_bytes + 4096
    if dist is not None:
        sizes = shard.exchange_sizes(out_bytes, dev if backend == "nccl" else torch.device("cpu"))      # once: the synthetic window is the same every step
    out_t = torch.empty(cap, dtype=torch.uint8, device=dev)

    # N > 1: the text of step k is gathered on rank 0 (ONE variable-size RCCL gather per step) while step k+1 computes:
    # two output buffers, the gather runs asynchronously on RCCL's stream
    n_buf = 2 if dist is not None else 1
    out_bufs = [out_t] + [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n_buf - 1)]
    pending = [None] * n_buf
    recv = None
    if dist is not None and rank == 0:
        recv = [torch.empty(sum(sizes), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
    step_no = [0]
    t_wait = [0.0]                         # host time spent waiting for gathers (this rank)

    def step():
        i = step_no[0] % n_buf
        step_no[0] += 1
        if pending[i] is not None:
            tw = time.perf_counter()
            shard.wait_all(pending[i])     # the buffer's previous gather must have left it
            t_wait[0] += time.perf_counter() - tw
            pending[i] = None
        if kind == "mpileup":
            eng.stage_window(w)
            eng.mpileup_run(par, out_bufs[i].data_ptr(), cap)       # plan + emit: the text lands in the caller's buffer
        elif kind == "depth":
            eng.stage_window(w)
            eng.depth_run(par, out_bufs[i].data_ptr(), cap)
        else:
            plan()
        if dist is not None:
            # the single collective of the path: per-block column text -> rank 0 over RCCL/xGMI, true sizes (samtools_amd/shard.py)
            pending[i] = shard.gather_text_v(out_bufs[i], out_bytes, dst=0, sizes=sizes, recv=recv[i] if recv else None)

    def drain():
        for i in range(n_buf):
            if pending[i] is not None:
                tw = time.perf_counter()
                shard.wait_all(pending[i])
                t_wait[0] += time.perf_counter() - tw
                pending[i] = None

    if a.pmc_child:
        step(); torch.cuda.synchronize()
        return None
    # Per-kernel HIP events: a pair costs the stream 3-6 us and a step has a dozen scopes -- 0.07 ms of a 6.7 ms mpileup30 step, 0.035 of a
    # 0.70 ms -B step (profiles/r06_sessionR_profile_events_cost.log).  The warm-up steps run with every launch bracketed and name the
    # dominant kernel; the TIMED steps bracket that kernel only (the roofline's live duration, on the kernel's own stream); the table of
    # the other kernels comes from a pass of its own behind the timed region (kernels_ms_source).  STA_BENCH_PROFILE_ALL=1: as before.
    prof_all = bool(os.environ.get("STA_BENCH_PROFILE_ALL")) or a.warmup < 1 or not hasattr(eng, "profile_only")
    dom_timed = None
    if not prof_all:
        eng.profile_only(None); eng.profile(True); eng.profile_reset()
    for _ in range(a.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if not prof_all:
        wp = {k: v for k, v in eng.profile_get().items() if not k.startswith(("baq8", "baq7l", "baq7_list", "baq8_list"))}
        dom_timed = max(wp.items(), key=lambda kv: kv[1][1])[0] if wp else None
        if dom_timed is None:
            prof_all = True
    if dist is not None:
        dist.barrier()
    eng.profile_only(None if prof_all else dom_timed) if hasattr(eng, "profile_only") else None
    eng.profile(True)
    eng.profile_reset()
    torch.cuda.synchronize()
    t_wait[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = time.perf_counter() - t0     # this rank's own steps issued (kernels may still run, gathers may be in flight)
    drain()                                # every gather of the timed steps has completed inside the timed region
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0       # this rank done (text gathered / sent)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = eng.profile_get()
    kernels_ms_source = "HIP events around every launch inside the timed steps"
    if not prof_all:
        # the per-kernel table: its own pass, every launch bracketed, scaled to the timed step count; the dominant kernel keeps its timed-region figure
        n_tab = max(1, min(a.steps, 5))
        eng.profile_only(None); eng.profile_reset()
        for _ in range(n_tab):
            step()
        drain()
        torch.cuda.synchronize()
        tab = eng.profile_get()
        scale = a.steps / n_tab
        merged = {k: (int(round(v[0] * scale)), v[1] * scale) for k, v in tab.items()}
        merged.update(prof)
        prof = merged
        kernels_ms_source = ("'%s': HIP events inside the timed steps (the only launch bracketed there); the others: a pass of %d steps behind the timed region "
                             "with every launch bracketed" % (dom_timed, n_tab))
    eng.profile(False)
    per_rank = None
    gather_ms = None
    if dist is not None:
        # the collective on its own (outside the timed region): the same variable-size gather of the last step's text, three
        # times, nothing else running -- inside the steps it overlaps the next step's kernels and shows only as gather_wait_ms
        i = (step_no[0] - 1) % n_buf
        gts = []
        for _ in range(3):
            dist.barrier(); torch.cuda.synchronize()
            tg = time.perf_counter()
            shard.wait_all(shard.gather_text_v(out_bufs[i], out_bytes, dst=0, sizes=sizes, recv=recv[i] if recv else None))
            torch.cuda.synchronize()
            gts.append(time.perf_counter() - tg)
        gt = torch.tensor([min(gts)], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_ms = float(gt[0].item()) * 1e3
        # where every rank's time went (ms per step): its kernels by HIP events, host time waiting for gathers, time until its
        # own work was done, and the barrier slack up to the slowest rank
        dp = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": local, "device_name": dp.name,
                "device_uuid": str(getattr(dp, "uuid", "")), "pci_bus_id": getattr(dp, "pci_bus_id", None), "pid": os.getpid(),
                "kernels_ms": sum(v[1] for v in prof.values()) / a.steps, "gather_wait_ms": t_wait[0] / a.steps * 1e3,
                "issue_ms": t_issue / a.steps * 1e3, "own_ms": t_own / a.steps * 1e3, "barrier_slack_ms": (dt - t_own) / a.steps * 1e3,
                "out_bytes": out_bytes, "piled_bases": piled}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    tt = torch.tensor([dt, float(piled)], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, piled_all = float(tmax[0].item()), float(tsum[1].item())
    else:
        dt_all, piled_all = dt, float(piled)

    if rank == 0:
        last = (step_no[0] - 1) % n_buf
        timed_sha = None
        if kind in ("mpileup", "depth"):
            # what was just timed: the whole text (all ranks' blocks at N > 1), hashed
            src = recv[last][:sum(sizes)] if recv else out_bufs[last][:out_bytes]
            timed_sha = hashlib.sha256(src.cpu().numpy().tobytes()).hexdigest()
        value = piled_all * a.steps / dt_all / 1e6
        pmc, pmc_src = None, None
        if world == 1 and not a.no_pmc and not secondary:
            pmc = collect_pmc(a, wlname, prof)
            pmc_src = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) of one step of this command on this GPU"
        if pmc is None and not a.cols:
            for tag in ("r02", "r01"):
                pth = os.path.join(REPO, "profiles", "%s_%s_pmc_traffic.json" % (tag, wlname))
                if os.path.exists(pth):
                    raw = json.load(open(pth))
                    pmc = {k: {c: v[c]["per_launch"] for c in v} for k, v in raw.items()}
                    pmc_src = "recorded: profiles/%s (not measured in this run)" % os.path.basename(pth)
                    break

        def traffic_of(name):
            pre = KNAME.get(name, name)
            ent = next((v for k, v in (pmc or {}).items() if k.startswith(pre)), None)
            if not ent or "FETCH_SIZE" not in ent or "WRITE_SIZE" not in ent:
                return None
            return (ent["FETCH_SIZE"] * FETCH_SCALE + ent["WRITE_SIZE"] * WRITE_SCALE) * 1024.0

        def roof(name, bpb=None):
            """SURVEY.md 8(d) roofline of one kernel: the path's algorithmic bytes per piled base (each staged byte read once, each
            output byte written once) x the bases one launch processes / the kernel's average launch time, against 8 TB/s.
            bpb: the text kernels are priced with the pileup path's own bytes (no BAQ reference window)."""
            alg_bpb = bpb if bpb is not None else spec["bpb"]
            launches, ms = prof[name]
            per_step = max(1, launches // max(1, a.steps))
            avg_ms = ms / max(1, launches)
            units = piled / per_step
            ach = alg_bpb * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            tr = traffic_of(name)
            r = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                 "traffic": tr, "traffic_ratio": (tr / (alg_bpb * units)) if (tr is not None and units) else None,
                 "traffic_source": (pmc_src + "; FETCH_SIZE x %.1f (gfx950 calibration)" % FETCH_SCALE) if tr is not None else None,
                 "alg_bytes_per_unit": alg_bpb, "units_per_launch": units, "avg_launch_ms": avg_ms, "launches_per_step": per_step}
            if name in BAQ_FP64_OPS_PER_BASE:
                # BAQ is fp64 work ("flops, not bytes, then dominate BAQ (report separately)")
                ops = BAQ_FP64_OPS_PER_BASE[name] * units
                tops = ops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
                r["fp64"] = {"ops_per_unit": BAQ_FP64_OPS_PER_BASE[name], "achieved": tops, "peak": FP64_PEAK_TOPS, "unit": "Tflop/s (no FMA)",
                             "frac": tops / FP64_PEAK_TOPS,
                             "counts": "the reference's fp64 operations only (probaln_glocal: 19 + 21 per band cell x 15 cells); the rows the kernel "
                                       "re-evaluates instead of storing, selects, integer and address work are excluded -- useful work, not pipe utilisation"}
                # the forward-row scratch stream the kernel pair moves through HBM: implementation traffic, reported as DRAM utilisation
                # (the fused class-S kernel writes AND reads it inside one launch: twice the bytes per unit)
                sb = float(os.environ.get("STA_BAQ_STREAM_BPB", "0")) or (2.0 * eng_baq7s_stream_bpb if name == "baq_s" else eng_baq_stream_bpb)
                r["dram_util"] = {"stream_bytes_per_unit": sb, "achieved": sb * units / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s"}
                r["dram_util"]["frac"] = r["dram_util"]["achieved"] / HBM_PEAK_GBS
            return r

        # forward rows streamed per query base by the BAQ pair (2 doubles per band cell and stored row, see kernels_baq.hip)
        eng_baq_stream_bpb = float(sa.baq_stream_bytes_per_base()) if hasattr(sa, "baq_stream_bytes_per_base") else 240.0
        eng_baq7s_stream_bpb = float(sa.baq7s_stream_bytes_per_base()) if hasattr(sa, "baq7s_stream_bytes_per_base") else 80.0
        # kernels on the side stream (band-8 BAQ groups) overlap the main ones: they cannot be "the" dominant kernel
        main_k = {k: v for k, v in prof.items() if not k.startswith(("baq8", "baq7l"))}
        dom_name = max(main_k.items(), key=lambda kv: kv[1][1])[0] if main_k else None
        res = {
            "metric": "Mbases piled/s (mpileup, 30x 150bp)" if wlname == "mpileup30" else "Mbases piled/s (%s: %s, %dx 150bp)" % (wlname, " ".join(spec["argv"][:-2] if kind == "mpileup" else spec["argv"][:-1]), depth),
            "value": value, "unit": "Mbases/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_all / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/f64" if (spec["baq"] or kind in ("glf", "calmd") or wlname == "consensus30") else "u8",
            "data": "synthetic",
            "config": {"workload": wlname, "command": " ".join(x for x in spec["argv"] if x != "{sam}").replace("{fa}", "ref.fa"),
                       "read_len": 150, "depth": depth, "window_cols_per_gpu": cols_per_gpu, "input_cols": n_cols, "reads_per_gpu": n_reads_of(rd), "input_files": spec["files"],
                       "piled_bases_per_gpu_step": piled, "out_bytes_per_gpu_step": out_bytes,
                       "staged_in_bytes_per_gpu": in_bytes,
                       "parallelism": "one sorted input, reference columns sharded x%d (+ mate halo), 1 variable-size RCCL gather" % world},
            "roofline": roof(dom_name, spec["bpb_pileup"] if dom_name and dom_name.startswith("mplp_") else None) if dom_name else None,
            "kernels_ms_per_step": {k: v[1] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
            "kernels_ms_source": kernels_ms_source,
        }
        emit_name = next((k for k in ({"mpileup": ["mplp_fused", "mplp_emit_deep", "mplp_emit"], "depth": ["depth_fused", "depth_emit"], "glf": ["glf_cols"], "calmd": ["md_emit"], "consensus": ["cons_col"]}[kind]) if k in prof), None)
        if emit_name and emit_name != dom_name:
            res["roofline_pileup"] = roof(emit_name, spec["bpb_pileup"])
        # whole step against the same roof: every kernel of the step, SURVEY.md 8d bytes
        res["roofline_step"] = {"bound": "hbm", "achieved": alg_bpb * piled_all / (dt_all / a.steps) / 1e9 / max(1, world), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s per GPU", "alg_bytes_per_unit": alg_bpb}
        res["roofline_step"]["frac"] = res["roofline_step"]["achieved"] / HBM_PEAK_GBS
        res["output_sha256"] = timed_sha
        if per_rank:
            res["per_rank"] = per_rank
            # what the collective actually ran on: the process group's own answers, not this script's arguments
            try:
                nccl_v = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
            except Exception:
                nccl_v = None
            res["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": nccl_v,
                                  "distinct_devices": len({(r.get("device_uuid") or r["device"], r.get("pci_bus_id")) for r in per_rank}),
                                  "launcher": "bench.py self-launch" if os.environ.get("STA_BENCH_SELF_LAUNCHED") else "external (torch.distributed.run)",
                                  "one_device_test_hook": bool(os.environ.get("STA_BENCH_ONE_DEVICE"))}
            moved = sum(sizes) - sizes[0]
            res["gather"] = {"what": "one 8-byte size all-gather (once) + ONE variable-size gather of the ranks' text per step "
                                     "(dist.batch_isend_irecv = ncclGroupStart / ncclSend / ncclRecv on RCCL), overlapped with the next step",
                             "backend": backend, "bytes_total": sum(sizes), "bytes_over_links": moved,
                             "ms_isolated": gather_ms, "gbs_isolated": moved / (gather_ms * 1e-3) / 1e9 if gather_ms else None,
                             "wait_ms_per_step_rank0": per_rank[0]["gather_wait_ms"]}
            # The ceiling of the design: every rank's text crosses ITS OWN xGMI link into rank 0 (point to point,
            # 7 links, 76.8 GB/s per direction at the link's peak = the "~153 GB/s" per link of both directions), all links at once, so
            # one step's gather takes the largest peer block / 76.8 GB/s at best.  A step that computes faster than that is gather-bound:
            # the N-GPU factor over one GPU is then N x (one-GPU step) / (gather time), not N.
            XGMI_GBS_PER_DIR = 76.8
            peer_max = max(sizes[1:]) if len(sizes) > 1 else 0
            pred_ms = peer_max / (XGMI_GBS_PER_DIR * 1e9) * 1e3
            res["gather"]["predicted"] = {"xgmi_link_gbs_per_direction_peak": XGMI_GBS_PER_DIR, "largest_peer_block_bytes": peer_max,
                                          "ms_at_link_peak": pred_ms, "measured_over_predicted": (gather_ms / pred_ms) if gather_ms and pred_ms else None,
                                          "step_is_gather_bound_at_link_peak": bool(pred_ms > res["ms_per_step"])}
        if kind in ("mpileup", "depth") and a.verify:
            # the timed window itself, byte for byte (hash of the whole text) against the oracle on the same seeds
            o = oracle_text_hash(wlname, n_cols, chunk_cols=cols_per_gpu)
            res["verify"] = {"oracle_sha256": o["sha256"] if o else None, "identical": bool(o and o["sha256"] == timed_sha),
                             "bytes": o["bytes"] if o else None, "oracle_seconds": o["seconds"] if o else None}
        if world == 1 and not a.no_cpu_baseline and not secondary:
            # bounded sample: BAQ runs at ~6 Mbases/s on one core, so 2 M columns (400 k reads, 60 Mbases) is ~10 s of CPU work;
            # the other workloads keep the same sample (the oracle needs 0.2-0.7 s there: generating and writing the SAM text
            # in Python costs far more than the run, so a bigger sample would only slow the bench down)
            sample = a.cpu_sample_cols or 2000000
            if depth >= 300:
                sample //= 10
            elif depth >= 100:
                sample //= 4
            want_e2e = wlname == "mpileup30" and not a.no_e2e
            inp_s = synth_inputs(wlname, sample)
            o_txt = os.path.join(inp_s["dir"], "oracle.txt") if want_e2e else None
            try:
                o = oracle_text_hash(wlname, sample, inputs=inp_s, save_to=o_txt)
                if want_e2e and o:
                    try:
                        res["e2e"] = e2e_file_to_text(inp_s, sample, o["sha256"], o_txt)
                    except Exception as ex:      # the bench line must not be lost to a failure of the file lane
                        res["e2e"] = {"error": repr(ex)}
            finally:
                shutil.rmtree(inp_s["dir"], ignore_errors=True)
            if o:
                res["cpu_baseline"] = {
                    "value": o["bases"] / o["seconds"] / 1e6, "unit": "Mbases/s", "cores": 1, "kind": "port",
                    "sample": "%s on %d synthetic reads (%d Mbases, %d columns, SAM text input, text hashed not stored), %.1f s wall, "
                              "oracle restatement (not the upstream binary: HTSlib is absent)" % (o["argv"], o["n_reads"], o["bases"] // 1000000, sample, o["seconds"])}
                if kind in ("mpileup", "depth"):
                    # the same sample through the engine: byte parity of every default bench run (the oracle is only the checker)
                    ref_s = torch.from_numpy(o["ref"].copy()).to(dev)
                    eng.set_reference(0, ref_s.data_ptr(), sample, 1)
                    ws, keep_s, _ = build_window(torch, np, sa, o["rd"], sample, dev, star_tags=spec["n_tags"])
                    eng.stage_window(ws)
                    inf = eng.mpileup_plan(par) if kind == "mpileup" else eng.depth_plan(par)
                    (eng.mpileup_emit if kind == "mpileup" else eng.depth_emit)()
                    got = eng.fetch_output(int(inf.out_bytes))
                    res["parity_check"] = {"sample_cols": sample, "bytes": len(got), "engine_sha256": hashlib.sha256(got).hexdigest(),
                                           "oracle_sha256": o["sha256"], "identical": hashlib.sha256(got).hexdigest() == o["sha256"]}
                if kind == "consensus":
                    # the same sample through the bulk entry: every column's (position, nth, depth, call, quality) against the rows
                    # of the oracle's `-f pileup` output for the same options (the oracle is only the checker)
                    ws, keep_s, _ = build_window(torch, np, sa, o["rd"], sample, dev)
                    nrs = int(o["rd"]["n"])
                    if cons_par.mode != 0:
                        xo_s = torch.arange(nrs + 1, dtype=torch.int32, device=dev); xt_s = torch.full((nrs + 1,), ord("*"), dtype=torch.uint8, device=dev)
                        keep_s += [xo_s, xt_s]
                        ws.files[0].n_xcols = 1; ws.files[0].xcol_off = xo_s.data_ptr(); ws.files[0].xcol_text = xt_s.data_ptr(); ws.files[0].n_xcol_bytes = nrs
                    eng.stage_window(ws)
                    ci = eng.consensus_run(cons_par)
                    ins, cols, _, _, _ = eng.fetch_consensus(sample, ci)
                    ins = np.frombuffer(ins, dtype=np.int32, count=sample)
                    cv = np.frombuffer(cols, dtype=np.int32, count=int(ci.n_cols) * 3).reshape(-1, 3)
                    pos = np.repeat(np.arange(1, sample + 1, dtype=np.int64), ins + 1)
                    first = np.cumsum(ins + 1) - (ins + 1)
                    nth = np.arange(int(ci.n_cols), dtype=np.int64) - np.repeat(first, ins + 1)
                    keepc = (cv[:, 0] > 0) & (cv[:, 1] != ord("*"))
                    got_rows = np.stack([pos[keepc], nth[keepc], cv[keepc, 0], cv[keepc, 1], cv[keepc, 2]], axis=1)
                    pargs = [x for x in spec["argv"] if x not in ("-f", "fastq")]
                    inp = synth_inputs(wlname, sample)
                    try:
                        pr = subprocess.run([os.path.join(REPO, "oracle", "_build", "oracle_samtools")] + [x.format(sam=inp["sam"], fa=inp["fa"]) for x in pargs[:-1]] + ["-f", "pileup", inp["sam"]],
                                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
                    finally:
                        shutil.rmtree(inp["dir"], ignore_errors=True)
                    want_rows = np.array([[int(f[1]), int(f[2]), int(f[3]), ord(f[4]), int(f[5])] for f in (l.split("\t") for l in pr.stdout.decode().split("\n") if l)], dtype=np.int64)
                    same = got_rows.shape == want_rows.shape and bool((got_rows == want_rows).all())
                    res["parity_check"] = {"sample_cols": sample, "columns": int(got_rows.shape[0]), "what": "(position, nth, depth, call, quality) of every column vs the oracle's -f pileup rows", "identical": same}
        return res
    return None


if __name__ == "__main__":
    main()
