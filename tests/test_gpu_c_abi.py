"""The boundary used from plain C (gcc, no C++/Python in the way): compile tests/c/abi_smoke.c
against include/ and the shared library, run it on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(ROOT, "rust_robotics_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-L", lib, "-lrust_robotics_amd", f"-Wl,-rpath,{lib}", "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_OK" in r.stdout, (r.returncode, r.stdout, r.stderr)
