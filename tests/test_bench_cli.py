"""bench.py's launcher behaviour without a GPU: `python bench.py --gpus N` from a bare shell re-executes itself under
torch.distributed.run (one rank per GPU), every rank gets as far as the engine's "no HIP device" check and says so, and
the process ends with a non-zero status instead of hanging or printing a number."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def has_gpu():
    from rust_robotics_amd import _ffi

    return int(_ffi.lib().rr_device_count()) > 0


@pytest.mark.skipif(has_gpu(), reason="exercises the no-device path")
def test_self_launch_reaches_the_device_check():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "no launcher detected" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.stderr.count("no HIP device available for rank") == 2, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no JSON line may be printed without a measurement"


@pytest.mark.skipif(has_gpu(), reason="exercises the no-device path")
def test_single_gpu_run_fails_loudly_without_a_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no HIP device available" in r.stderr
