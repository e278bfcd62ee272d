"""Build libbcehip.so and the bce CLI in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libbcehip.so")
BIN = os.path.join(HERE, "bin", "bce")


def build(jobs: int = 8, verbose: bool = False) -> str:
    """make -C bce_amd/csrc; returns the path of libbcehip.so."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError("build finished but %s is missing" % LIB)
    return LIB
