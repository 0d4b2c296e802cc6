"""The chunked BAM reader's device-inflate lane (host_chunk.cpp work_gpu_feeder / work_gpu_parse) without a GPU: a stand-in decoder
(tests/cpu/gpu_inflate_stub.cpp, STA_FAKE_GPU_INFLATE) inflates every batch on the host -- and, in mode 2, reports one job in seven as
given up by the device -- so the feeder's batching, the redo of refused blocks, the CRC check on the parser threads, the hand-over of
records that straddle groups and the end of the stream are exercised on the CPU.  Every mode must stage exactly what the reader's own
inflate stages.  What the lane replaces: bgzf.c inflate_block() under sam_read1 (bam_plcmd.c:409, bam2depth.c:541-543)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)


@pytest.fixture(scope="module")
def bench(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sb") / "stage_bench")
    subprocess.run(["bash", os.path.join(REPO, "scripts", "stage_bench.sh"), exe], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return exe


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    from synth import synth_ref, synth_reads, write_sam
    from bamio import sam_to_bam
    d = tmp_path_factory.mktemp("feeder")
    ref = synth_ref(200000, seed=3)
    rd = synth_reads(ref, depth=12, read_len=150, seed=8)
    sam = str(d / "in.sam"); write_sam(sam, rd, "chr1", len(ref))
    out = str(d / "in.bam"); sam_to_bam(sam, out, level=1, block=0xff00 - 77)      # (records straddle BGZF blocks and groups)
    return out


def _run(exe, bam, env, threads="4"):
    e = dict(os.environ, STA_NO_PINNED="1"); e.update(env)
    p = subprocess.run([exe, bam, threads, "65536", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    import re
    line = p.stdout.decode().strip().split("|")[-1]
    line = re.sub(r"pair_staged [0-9.]+ s = [0-9.]+ ns per read, ", "pair_staged ", line)      # (a timing line of the overlap-name bookkeeping)
    return line.split("wall")[0] + line.split("(checksum")[1]


def test_feeder_lane_stages_what_the_readers_own_inflate_stages(bench, bam):
    want = _run(bench, bam, {"STA_CHUNK_MAP": "0"})
    assert _run(bench, bam, {}) == want
    for mode in ("1", "2"):
        for batch in ("1", "3", "48"):
            for thr in ("1", "4"):
                assert _run(bench, bam, {"STA_FAKE_GPU_INFLATE": mode, "STA_GPU_INFLATE_BATCH": batch}, thr) == want, (mode, batch, thr)
    # the decoder comes up late (the reader starts with its own inflate and switches over mid-file), or not at all
    for delay in ("2", "15", "60"):
        assert _run(bench, bam, {"STA_FAKE_GPU_INFLATE": "2", "STA_GPU_INFLATE_BATCH": "3", "STA_FAKE_GPU_INFLATE_DELAY_MS": delay}) == want, delay
    assert _run(bench, bam, {"STA_FAKE_GPU_INFLATE": "3", "STA_FAKE_GPU_INFLATE_DELAY_MS": "5"}) == want
