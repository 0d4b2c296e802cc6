"""The C-ABI library must load and export every function include/*.h declares, and the headers must be
valid plain C (the reference is C; a maintainer would include them from bam_plcmd.c).  No compute here."""
import ctypes
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(REPO, "include")
LIB = os.path.join(REPO, "samtools_amd", "lib", "libsamtools_amd.so")


def declared_functions(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("#"))
    names = set()
    for m in re.finditer(r"\b(sta_\w+)\s*\(", text):
        name = m.group(1)
        # skip function-pointer typedefs: "(*sta_name)(" never matches because of the ')' before '('
        names.add(name)
    return sorted(names)


def test_every_declared_entry_point_is_exported():
    assert os.path.exists(LIB), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(LIB)
    missing = []
    total = 0
    for h in sorted(os.listdir(INC)):
        if not h.endswith(".h"):
            continue
        for fn in declared_functions(os.path.join(INC, h)):
            total += 1
            if not hasattr(lib, fn):
                missing.append("%s:%s" % (h, fn))
    assert total >= 50
    assert not missing, "declared but not exported: %s" % ", ".join(missing)


def test_headers_compile_as_c(tmp_path):
    src = tmp_path / "inc.c"
    src.write_text('#include "samtools_amd.h"\n#define STA_PLP_DROPIN\n#include "samtools_amd_plp.h"\n'
                   'int use(void) { bam_plp_t it = bam_plp_init(0, 0); bam_plp_destroy(it); return (int)sizeof(bam_pileup1_t) + (int)sizeof(sta_window); }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", INC, str(src)], check=True)


def test_no_device_fails_loudly():
    """There is no CPU fallback: without a HIP device engine creation reports STA_ERR_NO_DEVICE."""
    lib = ctypes.CDLL(LIB)
    lib.sta_device_count.restype = ctypes.c_int
    if lib.sta_device_count() > 0:
        return
    h = ctypes.c_void_p()
    assert lib.sta_engine_create(ctypes.byref(h), 0, None) == -2
    assert not h.value
