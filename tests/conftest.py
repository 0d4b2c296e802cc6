import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from product_paths import hipemu, product_exe, product_root  # noqa: E402

if hipemu():
    # STA_HIPEMU=1 / =asan, set by hand: the same tests against the CPU emulation build of the library's own sources
    # (tests/cpu/hipemu, test infrastructure for containers without a GPU).  `import samtools_amd` then finds the links to the package's
    # Python files that sit next to the emulated library.
    if not os.path.exists(product_exe()):
        raise SystemExit("STA_HIPEMU is set but %s is missing: make -C tests/cpu/hipemu%s" % (product_exe(), {"asan": " SAN=1", "ubsan": " SAN=ub", "trace": " TRACE=1"}.get(hipemu(), "")))
    sys.path.insert(0, product_root())


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# tests that need torch device memory, RCCL or bench-size windows: not for the CPU emulation (STA_HIPEMU)
_NOT_UNDER_HIPEMU = ("test_bench_launch.py", "test_gpu_fullsize.py", "test_gpu_benchsize_parity.py", "test_bulk_entry_through_the_c_abi",
                     "test_two_processes_one_gather", "test_three_processes_write_their_blocks_in_place", "test_one_process_over_rccl_with_device_capture",
                     "test_four_processes_unequal_blocks_against_the_oracle", "test_device_capture_is_the_host_capture",
                     "test_throughput_on_bam_like_blocks")


def pytest_collection_modifyitems(config, items):
    if not hipemu():
        return
    skip = pytest.mark.skip(reason="STA_HIPEMU: needs torch device memory / RCCL / a bench-size window")
    for it in items:
        if any(k in it.nodeid for k in _NOT_UNDER_HIPEMU):
            it.add_marker(skip)


def pytest_sessionstart(session):
    """The C-ABI library is built in-tree by __graft_entry__.build(); a fresh checkout that runs the tests first gets it built
    here (hipcc cross-compiles gfx950 without a GPU).  Nothing is built when the artefacts are already there."""
    if hipemu():
        return
    lib = os.path.join(REPO, "samtools_amd", "lib", "libsamtools_amd.so")
    exe = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
    if os.path.exists(lib) and os.path.exists(exe):
        # the controlling process maps the in-tree library too (the xdist workers and the CLI sub-processes do the work): a
        # missing or unloadable library fails the session here, not test by test
        import samtools_amd  # noqa: F401
        return
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        return          # the tests that need the library will say so
    subprocess.run(["make", "-j8", "-C", os.path.join(REPO, "samtools_amd", "csrc")], check=False, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_bin():
    """CPU oracle (test infrastructure only); built on demand with gcc."""
    exe = os.path.join(REPO, "oracle", "_build", "oracle_samtools")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, stdout=subprocess.DEVNULL)
    return exe


@pytest.fixture(scope="session")
def product_bin():
    """samtools-amd CLI (HIP engine).  Must already be built in-tree (see __graft_entry__.build)."""
    exe = product_exe()
    if not os.path.exists(exe):
        pytest.fail("samtools_amd/bin/samtools-amd is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return exe
