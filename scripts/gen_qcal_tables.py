#!/usr/bin/env python3
"""Extracts the platform quality-calibration tables of `samtools consensus` as DATA.

bam_consensus.c:446-662 holds `static_qcal[6]`: per platform three maps (substitution, undercall, overcall) from a reported quality
0..99 to a calibrated one -- constants of the reference's error model, tuned by its authors by hand ("manually tuned to work in
conjunction with other command line parameters used in the machine profiles").  They cannot be derived: an engine that is to give the
reference's answers for `-X hifi|hiseq|r10.4_sup|r10.4_dup|ultima` and `-t :name` has to carry the same numbers.  This script reads
them out of the reference tree and writes them, as numbers only, for the product (samtools_amd/csrc/cons_qcal_tables.inc) and for the
oracle (oracle/o_qcal_tables.inc).  Run once in the build container (the reference tree is not on the GPU box); both outputs are
committed.

    python scripts/gen_qcal_tables.py [/root/reference/bam_consensus.c]
"""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["flat", "hifi", "hiseq", "r10.4_sup", "r10.4_dup", "ultima"]


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/bam_consensus.c"
    text = open(src).read()
    a = text.index("static qcal_t static_qcal[6]")
    b = text.index("int set_qcal(", a)
    body = re.sub(r"//[^\n]*", "", text[a:b])
    nums = [int(x) for x in re.findall(r"-?\d+", body[body.index("{"):])]
    assert len(nums) == 6 * 3 * 100, len(nums)
    tables = [[nums[(t * 3 + m) * 100:(t * 3 + m + 1) * 100] + [0] for m in range(3)] for t in range(6)]      # element 100: zero-initialised in the C struct
    assert tables[0][0][:100] == list(range(100))
    head = ("// GENERATED DATA (scripts/gen_qcal_tables.py): the platform quality-calibration tables of `samtools consensus`\n"
            "// (bam_consensus.c:446-662, static_qcal[6]: substitution / undercall / overcall map per platform, 101 entries each, the last one\n"
            "// zero as in the C struct).  Numbers only; the order is flat, hifi, hiseq, r10.4_sup, r10.4_dup, ultima.\n")
    for path, ty in ((os.path.join(REPO, "samtools_amd", "csrc", "cons_qcal_tables.inc"), "int32_t"), (os.path.join(REPO, "oracle", "o_qcal_tables.inc"), "int")):
        with open(path, "w") as fh:
            fh.write(head)
            fh.write("static const char *const QCAL_NAMES[6] = { %s };\n" % ", ".join('"%s"' % n for n in NAMES))
            fh.write("static const %s QCAL_TABLES[6][3][101] = {\n" % ty)
            for t in range(6):
                fh.write("    {   // %s\n" % NAMES[t])
                for m in range(3):
                    rows = [", ".join("%2d" % v for v in tables[t][m][k:k + 20]) for k in range(0, 101, 20)]
                    fh.write("        { " + ",\n          ".join(rows) + " },\n")
                fh.write("    },\n")
            fh.write("};\n")
        print("wrote", path)


if __name__ == "__main__":
    main()
