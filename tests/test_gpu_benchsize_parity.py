"""Byte parity at the configurations bench.py measures.

The engine runs the exact window bench.py times (same generator, same seeds, device-resident arrays through the
C-ABI) and the sha256 of its text must equal the oracle's for the same reads:
  * mpileup30     4 194 304 columns, 838 860 reads, BAQ on  (~20-40 s of oracle)
  * mpileup300    524 288 columns, 1 048 576 reads, BAQ on  (the deep-column emit path + BAQ together)
  * depth30       8 388 608 columns, -a
  * (round 3) mpileup100[_B] 1 048 576 columns x 100; mpileup30_EA_pairs = BASELINE.json configs[4] (`-E -A`, proper pairs, mate
    overlaps under load) and its -B form; mpileup30[_B]_hotspot (30x + one 10 000x amplicon of 300 bp: the per-wave deep / fast
    emit choice); mpileup30_indel (5 % of the reads with an indel: band-8 and general-band BAQ under load)
  * (round 4) mpileup30[_B]_3files: three inputs of 10x each (the dat/mpileup.out.1 shape: per-file column groups, bam_plcmd.c:669-857);
    mpileup30_B_sOx: `-s -O --output-extra QNAME,NM` through the generic walkers k_mplp_len / k_mplp_emit (bam_plcmd.c:727-855)
  * engine paths that only an environment variable reaches: chunked BAQ slab (STA_BAQ_SLAB_GIB=1), no side stream
    for the band-8 groups (STA_BAQ_NO_SIDE_STREAM=1), at 786 432 columns
  * BASELINE.json configs[0]: examples/ex1.sam.gz (headerless, @SQ from the FASTA) + ex1.fa, SAM and BAM input.
Needs a GPU: -m gpu.  The oracle is only the checker."""
import gzip
import hashlib
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


_inputs = {}


def _shape(wl, n_cols):
    """generated once per (depth, columns): the numpy arrays for the engine, the SAM / FASTA for the oracle"""
    import atexit
    import shutil
    import bench
    spec = bench.WORKLOADS[wl]
    key = (spec["depth"], spec["files"], repr(sorted(spec["gen"].items())), n_cols)       # workloads of one shape share the generated input
    if key not in _inputs:
        for k in list(_inputs):                       # one shape at a time: these are hundreds of MB each
            shutil.rmtree(_inputs.pop(k)["dir"], ignore_errors=True)
        _inputs[key] = bench.synth_inputs(wl, n_cols)
        atexit.register(shutil.rmtree, _inputs[key]["dir"], True)
    return _inputs[key]


def _engine_sha(wl, n_cols, env=None):
    """sha256 of the engine's text for bench.py's window of workload `wl` (seeds 1 / 42)."""
    import numpy as np
    import torch
    import samtools_amd as sa
    import bench
    spec = bench.WORKLOADS[wl]
    kind = spec["kind"]
    dev = torch.device("cuda", 0)
    inp = _shape(wl, n_cols)
    ref, rd = inp["ref"], inp["rd"]
    saved = {}
    for k, v in (env or {}).items():
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        eng = sa.Engine(0, torch.cuda.current_stream().cuda_stream)
        ref_t = torch.from_numpy(ref.copy()).to(dev)
        eng.set_reference(0, ref_t.data_ptr(), n_cols, 1)
        w, keep, _ = bench.build_window(torch, np, sa, rd, n_cols, dev, star_tags=spec["n_tags"])
        eng.stage_window(w)
        if kind == "mpileup":
            par = sa.MplpParams.defaults(); par.has_fai = 1
            par.flag = (par.flag | spec["flags_on"]) & ~spec["flags_off"]
            if spec["max_depth"]:
                par.max_depth = spec["max_depth"]
            par.n_tags = spec["n_tags"]
            info = eng.mpileup_plan(par)
            out = torch.empty(int(info.out_bytes) + 64, dtype=torch.uint8, device=dev)
            eng.mpileup_emit(out.data_ptr(), out.numel())
        else:
            par = sa.DepthParams.defaults(); par.all_pos = 1
            info = eng.depth_plan(par)
            out = torch.empty(int(info.out_bytes) + 64, dtype=torch.uint8, device=dev)
            eng.depth_emit(out.data_ptr(), out.numel())
        eng.sync()
        text = out[:int(info.out_bytes)].cpu().numpy().tobytes()
        eng.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return hashlib.sha256(text).hexdigest(), len(text), int(info.piled_bases)


_oracle_cache = {}


def _oracle(wl, n_cols):
    import bench
    key = (wl, n_cols)
    if key not in _oracle_cache:
        o = bench.oracle_text_hash(wl, n_cols, inputs=_shape(wl, n_cols))
        assert o is not None, "oracle binary missing (make -C oracle)"
        _oracle_cache[key] = (o["sha256"], o["bytes"], o["bases"])
    return _oracle_cache[key]


# Workloads of one input shape share a worker (pytest.ini: --dist loadgroup), so the generated input is built once per shape and at
# most four of these windows (25 GiB of BAQ scratch each at 30x) are on the GPU at a time.
_BENCH_WL = [("mpileup30", "a"), ("mpileup30_B", "a"), ("mpileup300", "b"), ("mpileup300_B", "b"), ("mpileup100", "b"), ("mpileup100_B", "b"),
             ("mpileup30_EA_pairs", "c"), ("mpileup30_B_pairs", "c"), ("mpileup30_hotspot", "d"), ("mpileup30_B_hotspot", "d"),
             ("mpileup30_indel", "a"), ("depth30", "c"), ("mpileup30_trim", "f"),
             # round 4: three input files (per-file column groups of the tile kernels), the generic walkers (-s -O --output-extra)
             ("mpileup30_3files", "e"), ("mpileup30_B_3files", "e"), ("mpileup30_B_sOx", "a"),
             # -s on the tile path: tile kernel, read-major kernel (300x), three files
             ("mpileup30_B_s", "a"), ("mpileup300_B_s", "b"), ("mpileup30_B_s_3files", "e")]


@pytest.mark.parametrize("wl", [pytest.param(w, marks=pytest.mark.xdist_group("benchsize_" + g)) for w, g in _BENCH_WL])
def test_bench_window_text_is_byte_identical_to_the_oracle(wl):
    import bench
    n_cols = bench.WORKLOADS[wl]["cols"]
    want_sha, want_n, _ = _oracle(wl, n_cols)
    got_sha, got_n, piled = _engine_sha(wl, n_cols)
    assert got_n == want_n
    assert got_sha == want_sha
    assert piled > 0


@pytest.mark.xdist_group("benchsize_g")
def test_headline_window_text_is_byte_identical_to_the_oracle():
    """The window `python bench.py` steps by default (WORKLOADS["mpileup30"]["bench_cols"]: 16 M columns, 503 Mbases, 1.28 GB of text, assembled from
    4 M-column pieces): the whole text against the oracle's, by sha256 -- the oracle needs ~90 s for it."""
    import bench
    n_cols = bench.WORKLOADS["mpileup30"].get("bench_cols", bench.WORKLOADS["mpileup30"]["cols"])
    want_sha, want_n, _ = _oracle("mpileup30", n_cols)
    got_sha, got_n, piled = _engine_sha("mpileup30", n_cols)
    assert got_n == want_n
    assert got_sha == want_sha
    assert piled >= 30 * 0.99 * n_cols


@pytest.mark.xdist_group("benchsize_d")
@pytest.mark.parametrize("env", [{"STA_BAQ_SLAB_GIB": "1"}, {"STA_BAQ_NO_SIDE_STREAM": "1"}, {"STA_BAQ_SLAB_GIB": "1", "STA_BAQ_NO_SIDE_STREAM": "1"},
                                 # round 5: the builds of the class-S kernel (baq_band7s.h: 16 = M_LOGTAB, the default; 0 = the MAP quality from the
                                 # formula; 1 / 17 = plain instead of non-temporal row stream), the emit kernels without the XCD-aware tile mapping
                                 {"STA_BAQ7S_MODE": "0"}, {"STA_BAQ7S_MODE": "16"}, {"STA_BAQ7S_MODE": "1"}, {"STA_BAQ7S_MODE": "17"}, {"STA_XCD_MAP": "0"},
                                 # round 6: the list kernels held to a few resident workgroups that walk the list with the grid's stride
                                 {"STA_BAQ_LIST_MAX_WG": "3"}],
                         ids=["slab1g", "noside", "slab1g_noside", "baq7s_m0", "baq7s_m16", "baq7s_m1", "baq7s_m17", "no_xcd_map", "list_wg3"])
def test_env_only_engine_paths(env):
    n_cols = 3 << 18      # 786 432 columns: 157 286 reads, a 6.4 GB one-launch slab -> 7 chunks under STA_BAQ_SLAB_GIB=1
    want_sha, want_n, _ = _oracle("mpileup30", n_cols)
    got_sha, got_n, _ = _engine_sha("mpileup30", n_cols, env)
    assert (got_n, got_sha) == (want_n, want_sha)


def _ex1(tmp_path):
    """configs[0]: examples/ex1.sam.gz has no header; `samtools view -bt ex1.fa.fai` takes @SQ from the FASTA index."""
    from bamio import sam_to_bam
    gold = os.path.join(HERE, "golden", "examples")
    fa = os.path.join(gold, "ex1.fa")
    names, lens = [], []
    for line in open(fa):
        if line.startswith(">"):
            names.append(line[1:].split()[0]); lens.append(0)
        else:
            lens[-1] += len(line.strip())
    sam = str(tmp_path / "ex1.sam")
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\tSO:coordinate\n")
        for n, l in zip(names, lens):
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (n, l))
        fh.write(gzip.open(os.path.join(gold, "ex1.sam.gz"), "rt").read())
    bam = str(tmp_path / "ex1.bam")
    sam_to_bam(sam, bam)
    return sam, bam, fa


@pytest.mark.parametrize("opts", [[], ["-B"], ["-E", "-A"], ["-a"], ["-r", "seq2:100-900"]], ids=["default", "B", "EA", "a", "region"])
def test_config0_ex1(tmp_path, oracle_bin, product_bin, opts):
    sam, bam, fa = _ex1(tmp_path)
    want = subprocess.run([oracle_bin, "mpileup"] + opts + ["-f", fa, sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 500
    for inp in (sam, bam):
        got = subprocess.run([product_bin, "mpileup"] + opts + ["-f", fa, inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, got.stderr.decode()[-400:]
        assert got.stdout == want
    d_want = subprocess.run([oracle_bin, "depth", "-a", sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    d_got = subprocess.run([product_bin, "depth", "-a", bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert d_got.returncode == 0 and d_got.stdout == d_want
