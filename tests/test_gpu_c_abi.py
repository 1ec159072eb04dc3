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


def test_cpp_wrapper_consumer(tmp_path):
    """include/rust_robotics.hpp (the reference-named C++ classes over the C ABI) compiled with g++ and RUN:
    step loops, error kinds / messages, adaptive MCL, FastSLAM 1.0 / 2.0 through the wrapper"""
    exe = str(tmp_path / "hpp_smoke")
    lib = os.path.join(ROOT, "rust_robotics_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "hpp_smoke.cpp"),
           "-L", lib, "-lrust_robotics_amd", f"-Wl,-rpath,{lib}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "HPP_OK" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_example_ports_run(tmp_path):
    """the ports of the reference's examples (headless_localizers.rs, render_gif_particle_filter.rs, render_gif_slam.rs call
    patterns) through the C++ wrapper and the Python mirror"""
    import sys

    exe = str(tmp_path / "headless")
    lib = os.path.join(ROOT, "rust_robotics_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "cpp", "headless_localizers.cpp"),
           "-L", lib, "-lrust_robotics_amd", f"-Wl,-rpath,{lib}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "HEADLESS_OK" in r.stdout, (r.returncode, r.stdout, r.stderr)
    for script, token in (("headless_localizers.py", "HEADLESS_OK"), ("render_particle_filter.py", "RENDER_PF_OK"), ("fastslam_demo.py", None),
                          ("particle_filter_localization.py", None), ("adaptive_mcl.py", None)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PYTHONPATH=ROOT))
        assert r.returncode == 0 and (token is None or token in r.stdout), (script, r.returncode, r.stdout[-800:], r.stderr[-1500:])
