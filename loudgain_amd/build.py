"""Explicit, in-tree build of the HIP shared library (gfx950)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libloudscan_hip.so")


def build_all(force=False, verbose=False):
    cmd = ["make", "-C", CSRC] + (["-B"] if force else []) + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError("build did not produce " + LIB)
    return LIB
