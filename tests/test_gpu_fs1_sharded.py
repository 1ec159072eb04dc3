"""Sharded FastSLAM 1.0 (BASELINE config 4 shape) over the peer-to-peer transport.
Contiguous particle blocks, each with its whole map; the resample stores every served slot's
3 + 6L planes straight into the owning shard's slab.  Every shard's poses, weights and maps must
equal the unsharded engine's bit for bit (same seed, same obs_chunks), over steps where the
N_eff gate fires and steps where it does not."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 77


def engines(variant):
    """(unsharded class, sharded class, params with the gate at n / 1.5) of FastSLAM 1.0 or 2.0"""
    from rust_robotics_amd.slam import fastslam1 as fs
    from rust_robotics_amd.slam import fastslam2 as fs2

    if variant == 2:
        return fs2.FastSlam2, fs2.ShardedFastSlam2, fs2.default_params()
    return fs.FastSlam1, fs.ShardedFastSlam1, fs.default_params()


def scenario(n, L, steps, variant=1):
    """initial state, controls and observations shared by the sharded and unsharded runs"""
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests.test_gpu_fs1_parity import make_state, scene

    lms = scene(L, 5)
    poses, maps = make_state(n, L, lms, 6)
    prm = engines(variant)[2]
    base = prm.base if variant == 2 else prm
    base.nth = n / 1.5
    base.initial_weight = 1.0 / n
    xt = np.zeros(3)
    zs = []
    for t in range(steps):
        xt = np.array([xt[0] + 0.1 * np.cos(xt[2]), xt[1] + 0.1 * np.sin(xt[2]), xt[2] + 0.01])
        z = np.array(fs.get_observations(xt, [tuple(p) for p in lms], seed=SEED, step=t)).reshape(-1, 3)
        if t == 3:
            z = z[:0]  # a step without observations
        zs.append(z)
    return prm, poses, maps, zs


def unsharded(n, L, steps, chunks, variant=1):
    prm, poses, maps, zs = scenario(n, L, steps, variant)
    f = engines(variant)[0](n, L, params=prm, seed=SEED, obs_chunks=chunks)
    f.set_state(poses, maps)
    fired = []
    for z in zs:
        f.update([1.0, 0.1], z)
        fired.append(f.last_resample_fired())
    return f.get_state(), fired


def check(shard_states, n_local, L, steps, chunks, variant=1):
    (ep, em), fired = unsharded(n_local * len(shard_states), L, steps, chunks, variant)
    assert any(fired) and not all(fired), fired  # both branches of the gate were exercised
    for g, (p, m) in enumerate(shard_states):
        sl = slice(g * n_local, (g + 1) * n_local)
        assert np.array_equal(p.view(np.uint64), ep[sl].view(np.uint64)), f"rank {g}: poses / weights differ"
        assert np.array_equal(m.view(np.uint64), em[sl].view(np.uint64)), f"rank {g}: maps differ"


def run_in_process(world, n_local, L=7, steps=8, chunks=2, variant=1):
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1

    n = world * n_local
    prm, poses, maps, zs = scenario(n, L, steps, variant)
    shards = [engines(variant)[1](g, world, n_local, L, params=prm, seed=SEED, obs_chunks=chunks) for g in range(world)]
    for g, s in enumerate(shards):
        s.set_state(poses[g * n_local:(g + 1) * n_local], maps[g * n_local:(g + 1) * n_local])
    ShardedFastSlam1.link_local(shards)
    for t, z in enumerate(zs):
        for s in shards:  # only enqueued; the device-side waits pair the shards up
            s.update_async([1.0, 0.1], z)
        if t == 5:  # an accessor in mid-run has to make the pending (lazy) resample real
            for s in shards:
                s.poses()
    states = []
    for g, s in enumerate(shards):
        assert not s.timed_out(), f"rank {g}: a peer wait timed out"
        states.append(s.get_state())
    check(states, n_local, L, steps, chunks, variant)
    print("FS1_P2P_LOCAL_OK")


@pytest.mark.parametrize("world,n_local,variant", [(1, 3000, 1), (2, 2500, 1), (3, 1300, 1), (2, 2500, 2), (3, 1300, 2)])
def test_in_process_shards_equal_unsharded(world, n_local, variant):
    """fresh interpreter with enough hardware queues: see tests/test_gpu_p2p.py; variant 2 = FastSLAM 2.0
    (the proposal kernel reads the first observation's landmark through the lazy resample indices)"""
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_fs1_sharded import run_in_process; "
            f"run_in_process({world}, {n_local}, variant={variant})")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and "FS1_P2P_LOCAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_two_processes_over_ipc_handles():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "tests", "_gpu_fs1_p2p_worker.py"), "3000", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and r.stdout.count("FS1_P2P_OK") == 2, (r.stdout[-2000:], r.stderr[-4000:])


def test_shard_geometry_is_checked():
    from rust_robotics_amd.core import RoboticsError
    from rust_robotics_amd.slam.fastslam1 import FastSlam1, ShardedFastSlam1

    with pytest.raises(RoboticsError):
        FastSlam1(100, 2, first_global_index=50, n_global=120)  # block sticks out of the global range
    s = ShardedFastSlam1(0, 2, 100, 2)
    with pytest.raises(RoboticsError):
        s.update([1.0, 0.0], [])  # not connected
    with pytest.raises(RoboticsError):
        ShardedFastSlam1.link_local([s])  # world 2 declared, one shard linked
