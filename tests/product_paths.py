"""Where the library and the CLI under test live.

Normally that is the package itself: samtools_amd/lib/libsamtools_amd.so and samtools_amd/bin/samtools-amd, the HIP build.
With STA_HIPEMU=1 (or =asan / =ubsan / =trace) set BY HAND the same tests run against the CPU emulation build of the same sources
(tests/cpu/hipemu: test infrastructure for containers without a GPU, never a fallback -- nothing in the package, bench.py or
__graft_entry__ knows about it, and the driver's `-m gpu` run never sets the variable)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def hipemu():
    v = os.environ.get("STA_HIPEMU", "")
    return v if v in ("1", "asan", "ubsan", "trace") else ""


def product_root():
    """The directory that holds samtools_amd/{lib,bin} (and, for the emulation, links to the package's Python files)."""
    v = hipemu()
    if v:
        return os.path.join(HERE, "cpu", "hipemu", "_build", {"asan": "asan", "ubsan": "ubsan", "trace": "plain_trace"}.get(v, "plain"))
    return REPO


def lib_dir():
    return os.path.join(product_root(), "samtools_amd", "lib")


def product_exe():
    return os.path.join(product_root(), "samtools_amd", "bin", "samtools-amd")
