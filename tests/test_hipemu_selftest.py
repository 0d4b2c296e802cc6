"""The CPU emulation of the kernels (tests/cpu/hipemu, test infrastructure for containers without a GPU) checked on kernels small
enough to work out by hand: wave operations on full and partial waves, an `if` with wave operations inside a persistent ticket loop
(the lanes that skip it wait at the loop top, which has the lower address), a loop with lane-dependent trip count and a full-wave
shuffle behind it, an inner loop only some lanes enter, decoupled look-back between workgroups.  Every expected value is what a wave64
machine that reconverges at the immediate post-dominator computes (tests/cpu/hipemu/selftest.cpp)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang++ (host compile)")
def test_hipemu_semantics_on_hand_checked_kernels():
    d = os.path.join(HERE, "cpu", "hipemu")
    p = subprocess.run(["make", "-C", d, "selftest"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "hipemu selftest: ok" in out, out[-3000:]
