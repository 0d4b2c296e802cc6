"""The wave functions of the device's BGZF decoder (samtools_amd/csrc/bgzf_inflate_dev.h, what every wave of k_bgzf_inflate runs)
on the CPU against zlib: tests/cpu/inflate_emul.cpp -- stored / fixed / dynamic blocks from every zlib level and strategy, sizes
0 .. 65 280, damaged streams.  Reference: RFC 1951; HTSlib bgzf.c inflate_block() is what the decoder stands for."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def test_wave_inflate_functions_match_zlib(tmp_path):
    exe = str(tmp_path / "inflate_emul")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(REPO, "samtools_amd", "csrc"), os.path.join(HERE, "cpu", "inflate_emul.cpp"), "-lz", "-o", exe], check=True)
    for seed in (1, 2):
        p = subprocess.run([exe, "600", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b" 0 wrong" in p.stdout
