"""GPU: the C-ABI exercised from a plain C program (no Python/torch in the process) -- the path a Julia `ccall`
host takes.  Compiles tests/c/abi_smoke.c with gcc against include/vcmi.h and runs it."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_host_program(tmp_path):
    libdir = os.path.join(ROOT, "voiceconversion.jl_amd")
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                           "-L", libdir, "-lvcmi", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout
