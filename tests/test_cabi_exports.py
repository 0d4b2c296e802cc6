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


def test_iterator_structs_have_htslib_layout(tmp_path):
    """The drop-in header declares bam1_t / bam_pileup1_t itself when htslib/sam.h is not included first: the layouts must
    be HTSlib's (>= 1.10, LP64) or a caller compiled against HTSlib would read garbage."""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stddef.h>
#include "samtools_amd.h"
#include "samtools_amd_plp.h"
_Static_assert(sizeof(hts_pos_t) == 8, "hts_pos_t");
_Static_assert(sizeof(bam1_core_t) == 48, "bam1_core_t size");
_Static_assert(offsetof(bam1_core_t, pos) == 0 && offsetof(bam1_core_t, tid) == 8 && offsetof(bam1_core_t, bin) == 12, "core head");
_Static_assert(offsetof(bam1_core_t, qual) == 14 && offsetof(bam1_core_t, l_extranul) == 15 && offsetof(bam1_core_t, flag) == 16, "core mid");
_Static_assert(offsetof(bam1_core_t, l_qname) == 18 && offsetof(bam1_core_t, n_cigar) == 20 && offsetof(bam1_core_t, l_qseq) == 24, "core lens");
_Static_assert(offsetof(bam1_core_t, mtid) == 28 && offsetof(bam1_core_t, mpos) == 32 && offsetof(bam1_core_t, isize) == 40, "core mate");
_Static_assert(sizeof(bam1_t) == 80, "bam1_t size");
_Static_assert(offsetof(bam1_t, core) == 0 && offsetof(bam1_t, id) == 48 && offsetof(bam1_t, data) == 56, "bam1_t head");
_Static_assert(offsetof(bam1_t, l_data) == 64 && offsetof(bam1_t, m_data) == 68, "bam1_t tail");
_Static_assert(sizeof(bam_pileup_cd) == 8, "bam_pileup_cd");
_Static_assert(sizeof(bam_pileup1_t) == 40, "bam_pileup1_t size");
_Static_assert(offsetof(bam_pileup1_t, b) == 0 && offsetof(bam_pileup1_t, qpos) == 8 && offsetof(bam_pileup1_t, indel) == 12, "plp head");
_Static_assert(offsetof(bam_pileup1_t, level) == 16 && offsetof(bam_pileup1_t, cd) == 24 && offsetof(bam_pileup1_t, cigar_ind) == 32, "plp tail");
_Static_assert(sizeof(kstring_t) == 24, "kstring_t");
_Static_assert(sizeof(sta_plp_entry) == 16, "sta_plp_entry");
_Static_assert(sizeof(sta_glf_col) == 128, "sta_glf_col");
int main(void) { return 0; }
''')
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", INC, str(src)], check=True)


def test_ctypes_mirror_matches_the_header(tmp_path):
    """samtools_amd/_capi.py restates the header's structures by hand: sizes and a few field offsets must agree with what a C
    compiler sees (catches a forgotten field on either side without needing a GPU)."""
    import sys
    sys.path.insert(0, REPO)
    from samtools_amd import _capi
    names = [("sta_reads", _capi.Reads), ("sta_window", _capi.Window), ("sta_mplp_params", _capi.MplpParams),
             ("sta_depth_params", _capi.DepthParams), ("sta_plan_info", _capi.PlanInfo), ("sta_kernel_time", _capi.KernelTime),
             ("sta_glf_params", _capi.GlfParams), ("sta_glf_col", _capi.GlfCol), ("sta_calmd_params", _capi.CalmdParams),
             ("sta_cons_params", _capi.ConsParams), ("sta_cons_col", _capi.ConsCol), ("sta_cons_info", _capi.ConsInfo)]
    src = tmp_path / "sz.c"
    body = "".join('printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n, _ in names)
    body += 'printf("off_window_files %zu\\n", offsetof(sta_window, files));\nprintf("off_mplp_flag %zu\\n", offsetof(sta_mplp_params, flag));\n'
    body += 'printf("off_reads_xcol_off %zu\\n", offsetof(sta_reads, xcol_off));\n'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "samtools_amd.h"\nint main(void) {\n' + body + "return 0; }\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", INC, str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n") if l)
    for n, cls in names:
        assert int(out[n]) == ctypes.sizeof(cls), n
    assert int(out["off_window_files"]) == _capi.Window.files.offset
    assert int(out["off_mplp_flag"]) == _capi.MplpParams.flag.offset
    assert int(out["off_reads_xcol_off"]) == _capi.Reads.xcol_off.offset


def test_every_subcommand_refuses_to_compute_without_a_device(tmp_path):
    """No CPU fallback anywhere: on a machine without a HIP device every sub-command of the CLI must exit non-zero, print
    nothing on stdout and say why.  (Skipped on the GPU box, where the -m gpu tests run the same commands for real.)"""
    lib = ctypes.CDLL(LIB)
    lib.sta_device_count.restype = ctypes.c_int
    if lib.sta_device_count() > 0:
        import pytest
        pytest.skip("a HIP device is present")
    exe = os.path.join(REPO, "samtools_amd", "bin", "samtools-amd")
    gold = os.path.join(REPO, "tests", "golden")
    sam, fa = os.path.join(gold, "mpileup", "mp_D.sam"), os.path.join(gold, "mpileup", "mp.fa")
    cases = [["mpileup", sam], ["mpileup", "-f", fa, sam], ["depth", "-a", sam], ["plpdump", sam], ["coverage", sam], ["glf", sam],
             ["calmd", "-r", sam, fa], ["consensus", sam], ["consensus", "-m", "simple", "-f", "pileup", sam], ["bedcov", os.path.join(gold, "bedcov", "bedcov.bed"), os.path.join(gold, "bedcov", "bedcov.bam")]]
    for env_extra in ({}, {"STA_COV_ITERATOR": "1"}):
        for args in cases:
            p = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env_extra))
            assert p.returncode != 0, args
            assert p.stdout == b"", args
            assert b"HIP device" in p.stderr, (args, p.stderr[-200:])


def test_product_sources_never_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under samtools_amd/ or include/ may include, link, import or execute it
    (only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() use it, as the checker)."""
    import re
    bad = []
    roots = [os.path.join(REPO, "samtools_amd"), INC]
    for root in roots:
        for dp, _, fs in os.walk(root):
            for fn in fs:
                if not fn.endswith((".cpp", ".hip", ".h", ".py", "Makefile")):
                    continue
                text = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"oracle/|oracle_samtools|o_plp\.h|o_common\.h|import\s+oracle|from\s+oracle", text):
                    bad.append(os.path.relpath(os.path.join(dp, fn), REPO))
    assert not bad, bad
    # and the library does not link it
    out = subprocess.run(["ldd", LIB], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "oracle" not in out


def test_external_c99_client_builds_against_the_public_header_only(tmp_path):
    """tests/cabi/plp_client.c includes nothing of ours but include/samtools_amd_plp.h, uses the unprefixed HTSlib names
    (STA_PLP_DROPIN) and links -lsamtools_amd: it must build as pedantic C99 and, without a device, fail loudly."""
    from cabi_client import build_client
    exe = build_client(tmp_path)
    src = open(os.path.join(REPO, "tests", "cabi", "plp_client.c")).read()
    incs = re.findall(r'#include\s+"([^"]+)"', src)
    assert incs == ["samtools_amd_plp.h"], incs
    needed = subprocess.run(["readelf", "-d", exe], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "libsamtools_amd.so" in needed
    lib = ctypes.CDLL(LIB)
    lib.sta_device_count.restype = ctypes.c_int
    if lib.sta_device_count() > 0:
        return
    for mode in ([], ["-p"]):
        p = subprocess.run([exe] + mode + [os.path.join(REPO, "tests", "golden", "mpileup", "mp_D.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode != 0 and p.stdout == b"" and b"HIP device" in p.stderr


def test_external_consensus_client_builds_against_the_public_header_only(tmp_path):
    """tests/cabi/cons_client.c: a pedantic-C99 client of pileup_loop() under the reference's own names (STA_CONS_DROPIN)"""
    from cabi_client import build_cons_client
    exe = build_cons_client(tmp_path)
    src = open(os.path.join(REPO, "tests", "cabi", "cons_client.c")).read()
    assert re.findall(r'#include\s+"([^"]+)"', src) == ["samtools_amd_cons.h"]
    lib = ctypes.CDLL(LIB)
    lib.sta_device_count.restype = ctypes.c_int
    if lib.sta_device_count() > 0:
        return
    p = subprocess.run([exe, os.path.join(REPO, "tests", "golden", "consensus", "consen1.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stdout == b"" and b"HIP device" in p.stderr


def test_deep_strip_alignment_arithmetic_matches_plain_indexing(tmp_path):
    """samtools_amd/csrc/deep_strip.h (the per-block shift of k_mplp_emit_deep) against the straightforward per-column indexing,
    compiled for the host."""
    import subprocess
    exe = str(tmp_path / "deep_strip_test")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(REPO, "tests", "cpu", "deep_strip_test.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"deep_strip_test OK" in p.stdout, p.stdout.decode()[-400:]
