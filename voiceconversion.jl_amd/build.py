"""In-tree build of libvcmi.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs=4, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    so = os.path.join(_HERE, "libvcmi.so")
    if not os.path.exists(so):
        raise RuntimeError("libvcmi.so was not produced")
    return so
