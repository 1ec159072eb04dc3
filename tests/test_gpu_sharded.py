"""The real sharded path (HipShard + RCCL through torch.distributed) at the world size the
gpurun box offers (1): every phase kernel, the stream hand-over to torch and the collectives
run; the result must equal the unsharded engine bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_nccl_equals_unsharded():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29715", os.path.join(ROOT, "tests", "_gpu_sharded_worker.py"), "20000", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
