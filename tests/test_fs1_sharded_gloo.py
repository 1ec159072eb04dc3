"""Sharded FastSLAM 1.0 on CPU (world sizes 2, 3 and 8 -- the BASELINE's rank count --, gloo): the protocol the HIP engine runs
between GPUs -- contiguous particle blocks with their whole maps, global weight maximum, integer
sums of every shard, gate + systematic plan from the global totals, whole particles moved along
the segment matrix -- executed with the D-spec oracle standing in for the kernels must reproduce
the single-shard D-spec trajectory bit for bit.  (The GPU side of the same statement is
tests/test_gpu_fs1_sharded.py.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from oracle import dp, u32p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 91


def scenario(n, L, steps):
    """model, gate threshold, initial state (maps pre-initialised: EKF branch) and observations"""
    det = oracle.det()
    md = oracle.det_fs1_model()
    rng = np.random.default_rng(7)
    lms = rng.uniform(-12.0, 12.0, size=(L, 2))
    poses = np.column_stack([np.full(n, 1.0 / n), rng.normal(0, 0.2, n), rng.normal(0, 0.2, n), rng.normal(0, 0.05, n)])
    maps = np.zeros((n, L, 6))
    maps[:, :, 0] = lms[None, :, 0] + rng.normal(0, 0.5, (n, L))
    maps[:, :, 1] = lms[None, :, 1] + rng.normal(0, 0.5, (n, L))
    maps[:, :, 2] = maps[:, :, 5] = 0.5
    xt = np.zeros(3)
    zs = []
    for t in range(steps):
        xt = np.array([xt[0] + 0.1 * np.cos(xt[2]), xt[1] + 0.1 * np.sin(xt[2]), xt[2] + 0.01])
        out = np.empty((L, 3))
        cnt = det.det_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), L, 20.0, md.r00, md.r11, SEED, t, dp(out))
        z = np.ascontiguousarray(out[:cnt])
        zs.append(z[:0] if t == 2 else z)  # one step without observations
    return md, n / 1.5, poses, maps, zs


def single_shard(n, L, steps, chunks):
    det = oracle.det()
    md, nth, poses, maps, zs = scenario(n, L, steps)
    pw, px, py, pyaw = (np.ascontiguousarray(poses[:, k]) for k in range(4))
    planes = oracle.maps_aos_to_planes(maps, n, L)
    idx = np.empty(n, np.uint32)
    fired = []
    for t, z in enumerate(zs):
        fired.append(int(det.det_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), 1.0, 0.1,
                                            dp(z) if len(z) else None, len(z), C.byref(md), nth, SEED, t, t, chunks, u32p(idx))))
    return pw, px, py, pyaw, planes.reshape(L * 6, n), fired


@pytest.mark.parametrize("world,chunks,port", [(2, 1, 29641), (3, 2, 29642), (8, 2, 29643)])
def test_sharded_fastslam_equals_single_shard(tmp_path, world, chunks, port):
    n_local, L, steps = 300, 5, 9
    n = n_local * world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_fs1_sharded_worker.py"), str(tmp_path),
           str(n_local), str(L), str(steps), str(chunks)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    pw, px, py, pyaw, planes, fired = single_shard(n, L, steps, chunks)
    assert any(fired) and not all(fired), fired
    moved = 0
    for g in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{g}.npz"))
        assert d["fired"].tolist() == fired
        sl = slice(g * n_local, (g + 1) * n_local)
        for name, e in (("pw", pw), ("px", px), ("py", py), ("pyaw", pyaw)):
            assert np.array_equal(d[name].view(np.uint64), e[sl].view(np.uint64)), f"rank {g}: {name}"
        assert np.array_equal(d["planes"].reshape(L * 6, n_local).view(np.uint64), planes[:, sl].view(np.uint64)), f"rank {g}: maps"
        moved += int(d["moved"])
    assert moved > 0, "expected whole particles to migrate between ranks"
