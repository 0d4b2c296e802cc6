"""Every long and short option the reference's `mpileup` and `depth` accept is accepted by the engine's command line (CPU: options are
read before the device is asked for anything).

The reference splices SAM_OPT_GLOBAL_OPTIONS(...) into each command's getopt table (sam_opts.h:63-71; bam_plcmd.c:1098 enables
--input-fmt-option, --reference, --write-index, --verbosity for mpileup; bam2depth.c:765 those and --threads / -@ for depth) and
leaves the disabled ones in the table with the value '?', so that naming them prints the usage.  Round 5's getopt tables lacked them:
`depth: unrecognized option '--verbosity'` -- a script that runs under samtools failed here.

The option lists below are the reference's tables as data (names + whether they take an argument); where /root/reference is readable
(this container, not the GPU box) the tables are also read from the sources and compared with these lists."""
import os
import re
import subprocess

import pytest

from product_paths import product_exe

REF = "/root/reference"

# (long name, takes an argument) -- bam_plcmd.c:1099-1147
MPILEUP_LONG = [
    ("rf", 1), ("ff", 1), ("incl-flags", 1), ("excl-flags", 1), ("output", 1), ("output-QNAME", 0), ("output-qname", 0), ("illumina1.3+", 0),
    ("count-orphans", 0), ("bam-list", 1), ("no-BAQ", 0), ("no-baq", 0), ("adjust-MQ", 1), ("adjust-mq", 1), ("max-depth", 1), ("redo-BAQ", 0),
    ("redo-baq", 0), ("fasta-ref", 1), ("exclude-RG", 1), ("exclude-rg", 1), ("positions", 1), ("region", 1), ("ignore-RG", 0), ("ignore-rg", 0),
    ("min-MQ", 1), ("min-mq", 1), ("min-BQ", 1), ("min-bq", 1), ("ignore-overlaps-removal", 0), ("disable-overlap-removal", 0), ("output-mods", 0),
    ("output-BP", 0), ("output-bp", 0), ("output-BP-5", 0), ("output-bp-5", 0), ("output-MQ", 0), ("output-mq", 0), ("customized-index", 0),
    ("reverse-del", 0), ("output-extra", 1), ("output-sep", 1), ("output-empty", 1), ("no-output-ins", 0), ("no-output-ins-mods", 0),
    ("no-output-del", 0), ("no-output-ends", 0),
]
MPILEUP_SHORT = "Af:r:l:q:Q:RC:Bd:b:o:EG:6OsxXaM"               # bam_plcmd.c:1150
# bam2depth.c:757-763
DEPTH_LONG = [("min-MQ", 1), ("min-mq", 1), ("min-BQ", 1), ("min-bq", 1), ("excl-flags", 1), ("incl-flags", 1), ("require-flags", 1)]
DEPTH_SHORT = "@:q:Q:JHd:m:l:g:G:o:ar:Xf:b:s"                   # bam2depth.c:768
# SAM_OPT_GLOBAL_OPTIONS (sam_opts.h:63-71), in its order: input-fmt, input-fmt-option, output-fmt, output-fmt-option, reference, threads;
# write-index and verbosity are always on.  '-' = disabled (the entry stays and returns '?': usage), 0 = long only, c = short option c
GLOBAL = ["input-fmt", "input-fmt-option", "output-fmt", "output-fmt-option", "reference", "threads"]
GLOBAL_ARGS = {"mpileup": ("-", 0, "-", "-", 0, "-"), "depth": ("-", 0, "-", "-", 0, "@")}        # bam_plcmd.c:1098, bam2depth.c:765


def _value_for(name, tmp):
    """an argument the option accepts"""
    if name in ("rf", "ff", "incl-flags", "excl-flags", "require-flags", "g", "G"): return "UNMAP,DUP"
    if name in ("fasta-ref", "reference", "f"): return tmp["fa"]
    if name in ("positions", "l", "b"): return tmp["bed"]
    if name in ("exclude-RG", "exclude-rg"): return tmp["list"]
    if name in ("bam-list",): return tmp["list"]
    if name in ("region", "r"): return "chr1:1-10"
    if name in ("output", "o"): return os.path.join(tmp["dir"], "out.txt")
    if name in ("output-extra",): return "QNAME,NM"
    if name in ("output-sep", "output-empty"): return "x"
    if name in ("input-fmt-option", "output-fmt-option"): return "nthreads=2"
    if name in ("input-fmt", "output-fmt"): return "bam"
    return "7"


@pytest.fixture(scope="module")
def tmp(tmp_path_factory):
    d = tmp_path_factory.mktemp("gopt")
    fa = d / "r.fa"; fa.write_text(">chr1\nACGTACGTACGTACGTACGT\n")
    bed = d / "r.bed"; bed.write_text("chr1\t1\t10\n")
    lst = d / "names.txt"; lst.write_text("grp1\n")
    sam = d / "in.sam"; sam.write_text("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:20\nr1\t0\tchr1\t2\t60\t5M\t*\t0\t0\tCGTAC\tIIIII\n")
    return {"dir": str(d), "fa": str(fa), "bed": str(bed), "list": str(lst), "sam": str(sam)}


def _run(cmd, opts, tmp, infile=None):
    exe = product_exe()
    if not os.path.exists(exe):
        pytest.fail("samtools_amd/bin/samtools-amd is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    p = subprocess.run([exe, cmd] + opts + [infile or os.path.join(tmp["dir"], "no-such-input.bam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, STA_NO_DEVICE_WAIT="1"), timeout=120)
    return p.returncode, p.stderr.decode(errors="replace")


def _rejected(err):
    return "Usage: samtools" in err or "unrecognized option" in err or "invalid option" in err or "requires an argument" in err


def _cases(cmd):
    longs = MPILEUP_LONG if cmd == "mpileup" else DEPTH_LONG
    short = MPILEUP_SHORT if cmd == "mpileup" else DEPTH_SHORT
    out = [("--" + n, a, n) for n, a in longs]
    for m in re.finditer(r"([A-Za-z0-9@])(:?)", short):
        out.append(("-" + m.group(1), 1 if m.group(2) else 0, m.group(1)))
    for name, en in zip(GLOBAL, GLOBAL_ARGS[cmd]):
        if en != "-":
            out.append(("--" + name, 1, name))
    out += [("--write-index", 0, "write-index"), ("--verbosity", 1, "verbosity")]
    return out


@pytest.mark.parametrize("cmd", ["mpileup", "depth"])
def test_no_option_of_the_reference_table_is_rejected(cmd, tmp):
    bad = []
    for opt, has_arg, name in _cases(cmd):
        if cmd == "mpileup" and opt in ("-E", "--redo-BAQ", "--redo-baq"):
            opts = [opt]                                # (alone it is fine; with -B the reference refuses the pair, as does the engine)
        else:
            opts = [opt] + ([_value_for(name, tmp)] if has_arg else [])
        if cmd == "depth" and opt == "-X":
            continue                                    # (-X halves the file list: covered by tests/test_bai_index.py)
        rc, err = _run(cmd, opts, tmp)
        if _rejected(err):
            bad.append((opt, err.strip().splitlines()[:2]))
    assert not bad, bad


@pytest.mark.parametrize("cmd", ["mpileup", "depth"])
def test_disabled_global_options_print_the_usage_as_in_the_reference(cmd, tmp):
    for name, en in zip(GLOBAL, GLOBAL_ARGS[cmd]):
        if en != "-":
            continue
        rc, err = _run(cmd, ["--" + name, _value_for(name, tmp)], tmp)
        assert rc == 1 and "Usage: samtools " + cmd in err, (name, err)


@pytest.mark.parametrize("cmd", ["mpileup", "depth"])
def test_global_option_values_are_checked_like_parse_sam_global_opt(cmd, tmp):
    rc, err = _run(cmd, ["--verbosity", "loud"], tmp)                  # sam_opts.c:147-152
    assert rc == 1 and "Invalid verbosity value." in err and "Usage:" in err
    rc, err = _run(cmd, ["--input-fmt-option", "no_such_key=1"], tmp)  # hts_opt_add: "Unknown option"
    assert rc == 1 and "Unknown option 'no_such_key'" in err and "Usage:" in err
    for ok in ("nthreads=4", "NTHREADS=4", "required_fields=0x1ff", "decode_md=0", "level=5"):
        rc, err = _run(cmd, ["--input-fmt-option", ok], tmp)
        assert not _rejected(err), (ok, err)
    if cmd == "depth":
        rc, err = _run(cmd, ["--threads", "many"], tmp)                # sam_opts.c:137-142
        assert rc == 1 and "Invalid threads value." in err
        rc, err = _run(cmd, ["-@", "2", "--reference", tmp["fa"]], tmp)
        assert not _rejected(err), err


def test_mpileup_reference_option_names_the_fasta_when_f_did_not(tmp):
    """bam_plcmd.c:1223-1227: `--reference FILE` is the global option; the FASTA is loaded from it unless -f / --fasta-ref gave one"""
    rc, err = _run("mpileup", ["--reference", os.path.join(tmp["dir"], "missing.fa")], tmp)
    assert rc == 1 and "fai_load" in err and "missing.fa" in err
    rc, err = _run("mpileup", ["-f", tmp["fa"], "--reference", os.path.join(tmp["dir"], "missing.fa")], tmp)
    assert "fai_load" not in err and not _rejected(err)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not here (GPU box)")
def test_the_lists_above_are_the_reference_tables():
    def table(path, start_pat):
        src = open(os.path.join(REF, path)).read()
        body = src[src.index(start_pat):]
        body = body[:body.index("{NULL, 0, NULL, 0}")]
        names = [(m.group(1), 1 if m.group(2) == "required_argument" else 0) for m in re.finditer(r'\{"([^"]+)",\s*(required_argument|no_argument)', body)]
        glob = re.search(r"SAM_OPT_GLOBAL_OPTIONS\(([^)]*)\)", body).group(1)
        short = re.search(r'getopt_long\(argc, argv, "([^"]+)"', src[src.index(start_pat):]).group(1)
        return names, tuple(0 if a.strip() == "0" else a.strip().strip("'") for a in glob.split(",")), short
    names, glob, short = table("bam_plcmd.c", "static const struct option lopts[] =\n    {\n        SAM_OPT_GLOBAL_OPTIONS")
    assert names == MPILEUP_LONG and glob == GLOBAL_ARGS["mpileup"] and short == MPILEUP_SHORT
    names, glob, short = table("bam2depth.c", 'static const struct option lopts[] = {\n        {"min-MQ"')
    assert names == DEPTH_LONG and glob == GLOBAL_ARGS["depth"] and short == DEPTH_SHORT
    hdr = open(os.path.join(REF, "sam_opts.h")).read()
    macro = hdr[hdr.index("#define SAM_OPT_GLOBAL_OPTIONS"):]
    assert re.findall(r'\{"([a-z-]+)",', macro[:macro.index("/*")]) == GLOBAL + ["write-index", "verbosity"]
