"""The sharded transports across two PHYSICAL GPUs (one process per device).  Skipped on a one-GPU box; on the
first multi-GPU box these are the tests that exercise what everything else can only simulate on one device:
system-scope mailbox records and fine-grained inbox stores crossing xGMI, hipIpc mappings of another device's
memory, RCCL all-reduce / all-gather / send-recv between devices -- each bit-identical to the unsharded filter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_devices():
    from rust_robotics_amd import _ffi

    return int(_ffi.lib().rr_device_count())


@pytest.mark.parametrize("what,kind,port", [("mcl", "p2p", 29741), ("mcl", "rccl", 29742), ("fastslam", "p2p", 29743), ("fastslam", "rccl", 29744)])
def test_two_ranks_on_two_devices(what, kind, port):
    if n_devices() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gpu_two_device_worker.py"), what, kind]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.count("TWO_DEVICE_OK") == 2, (r.stdout[-2000:], r.stderr[-4000:])
