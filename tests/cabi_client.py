"""Builds tests/cabi/plp_client.c -- a plain-C99 client that sees nothing but include/samtools_amd_plp.h and the shared
library -- exactly the way a samtools source file would be switched over: STA_PLP_DROPIN + -lsamtools_amd."""
import os
import subprocess

from product_paths import lib_dir

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def build_client(outdir):
    exe = os.path.join(str(outdir), "plp_client")
    lib = lib_dir()
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-DSTA_PLP_DROPIN",
                    "-I", os.path.join(REPO, "include"), os.path.join(HERE, "cabi", "plp_client.c"),
                    "-L", lib, "-lsamtools_amd", "-Wl,-rpath," + lib, "-o", exe], check=True)
    return exe


def build_cons_client(outdir):
    """tests/cabi/cons_client.c: the same for the consensus iterator (include/samtools_amd_cons.h, STA_CONS_DROPIN: pileup_loop)"""
    exe = os.path.join(str(outdir), "cons_client")
    lib = lib_dir()
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-DSTA_CONS_DROPIN",
                    "-I", os.path.join(REPO, "include"), os.path.join(HERE, "cabi", "cons_client.c"),
                    "-L", lib, "-lsamtools_amd", "-Wl,-rpath," + lib, "-o", exe], check=True)
    return exe
