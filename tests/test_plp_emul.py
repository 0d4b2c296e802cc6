"""The step functions of the mpileup tile kernels (samtools_amd/csrc/plp_tile.h: k_mplp_len_rm, k_mplp_emit_tile) run on the CPU by
tests/cpu/plp_emul.cpp -- the same source the device executes per thread, in loops over the lanes -- and their text is diffed
against the oracle's `mpileup -B` for seeded reads with messy CIGARs.  Test infrastructure: the product has no CPU path.
Also: the packed-byte helpers against plain per-byte loops (tests/cpu/plp_swar_test.cpp)."""
import os
import random
import subprocess

import numpy as np
import pytest

from synth import synth_ref, synth_reads, write_sam, write_fasta
from synth_rich import _cigar_for

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(REPO, "samtools_amd", "csrc")
OPS = "MIDNSHP=XB"


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    d = tmp_path_factory.mktemp("plp_emul")
    exes = []
    for mode in (0, 1):
        exe = str(d / ("plp_emul%d" % mode))
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DPLP_EMUL_ANY=%d" % mode, "-I/opt/rocm/include",
                        os.path.join(HERE, "cpu", "plp_emul.cpp"), "-o", exe], check=True)
        exes.append(exe)
    return exes


def _messy_reads(n_cols, depth, seed, frac_messy=0.08):
    """synth.py's reads with a share of the CIGARs replaced by clipped / spliced / padded ones (same query length)."""
    ref = synth_ref(n_cols, seed=seed)
    rd = synth_reads(ref[:n_cols - 1200], depth=depth, read_len=150, seed=seed + 1, indel_rate=0.04, max_indel=9)
    rng = random.Random(seed)
    n = rd["n"]
    cig = [[(int(c) >> 4, int(c) & 15) for c in rd["cigar"][int(rd["cig_off"][r]):int(rd["cig_off"][r + 1])]] for r in range(n)]
    for r in range(n):
        if rng.random() < frac_messy:
            _, _, ops = _cigar_for(rng, 150)
            cig[r] = [(l, OPS.index(o)) for l, o in ops]
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(c) for c in cig])
    rd["cig_off"] = off
    rd["cigar"] = np.array([(l << 4) | o for c in cig for l, o in c], dtype=np.uint32)
    rd["mapq"] = np.array([rng.choice((0, 13, 60, 60, 60, 93, 94, 200, 255)) for _ in range(n)], dtype=np.uint8)
    span = np.array([sum(l for l, o in c if o in (0, 2, 3, 7, 8)) for c in cig], dtype=np.int64)
    simple = np.array([len(c) == 1 and c[0][1] in (0, 7, 8) for c in cig])
    return ref, rd, span, simple


def _dump(tmp, ref, rd, span, simple, n_cols):
    d = str(tmp)
    keep = span > 0
    info = (np.where(keep, 3, 1) | np.where(simple, 4, 0) | np.where((rd["flag"] & 16) != 0, 8, 0) | (rd["mapq"].astype(np.uint32) << 8)).astype(np.uint32)
    for name, arr in (("pos.i32", rd["_abs_pos"].astype(np.int32)), ("end.i32", (rd["_abs_pos"] + span).astype(np.int32)), ("info.u32", info),
                      ("lq.i32", rd["l_qseq"]), ("cig_off.u32", rd["cig_off"]), ("b8.u32", rd["base_off8"]), ("cigar.u32", rd["cigar"]),
                      ("seq.u8", rd["seq"]), ("qual.u8", rd["qual"]), ("ref.u8", ref)):
        np.ascontiguousarray(arr).tofile(os.path.join(d, name))
    open(os.path.join(d, "meta.txt"), "w").write("%d %d chrS\n" % (rd["n"], n_cols))
    sam, fa = os.path.join(d, "x.sam"), os.path.join(d, "x.fa")
    write_sam(sam, rd, "chrS", n_cols)
    write_fasta(fa, "chrS", ref)
    return d, sam, fa


CASES = [
    # n_cols, depth, seed, min_baseQ, tile_cap, no_ends, all
    (20000, 30, 11, 13, 12288, 0, 0),
    (12000, 60, 12, 0, 12288, 0, 0),
    (12000, 30, 13, 30, 12288, 1, 0),
    (9000, 220, 14, 13, 12288, 0, 0),        # more than one batch of reads per measuring tile; every wave beyond the slice
    (9000, 90, 15, 13, 6144, 0, 0),          # rows around the slice size: both routes in one window
    (30000, 1, 16, 13, 12288, 0, 1),         # -a: zero-depth rows between the reads
    (30000, 1, 17, 13, 1024, 0, 0),
    # -s: the mapping-quality column as a third string of the tile kernels
    (20000, 30, 18, 13, 12288, 0, 0, 1),
    (9000, 90, 19, 20, 6144, 0, 0, 1),
    (30000, 1, 20, 13, 12288, 0, 1, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_tile_step_functions_reproduce_the_oracle_text(tmp_path, emul, oracle_bin, case):
    n_cols, depth, seed, minq, cap, no_ends, all_ = case[:7]
    mq_col = case[7] if len(case) > 7 else 0
    ref, rd, span, simple = _messy_reads(n_cols, depth, seed)
    d, sam, fa = _dump(tmp_path, ref, rd, span, simple, n_cols)
    args = ["mpileup", "-B", "-Q", str(minq), "-d", "1000000", "-f", fa]
    if no_ends:
        args.append("--no-output-ends")
    if all_:
        args.append("-a")
    if mq_col:
        args.append("-s")
    want = subprocess.run([oracle_bin] + args + [sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert want.count(b"\n") > 1000
    for exe in emul:
        got = subprocess.run([exe, d, str(minq), str(cap), str(no_ends), str(all_), str(mq_col)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert got.returncode == 0, got.stderr.decode()[-600:]
        if got.stdout != want:
            g, w = got.stdout.split(b"\n"), want.split(b"\n")
            for i, (a, b) in enumerate(zip(g, w)):
                if a != b:
                    raise AssertionError("%s: first difference at row %d:\n got  %r\n want %r" % (os.path.basename(exe), i, a[:300], b[:300]))
            raise AssertionError("%s: %d rows vs %d" % (os.path.basename(exe), len(g), len(w)))


def test_packed_byte_helpers_match_plain_loops(tmp_path):
    exe = str(tmp_path / "plp_swar_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(HERE, "cpu", "plp_swar_test.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0 and b"plp_swar_test OK" in p.stdout, p.stdout.decode()[-600:]
