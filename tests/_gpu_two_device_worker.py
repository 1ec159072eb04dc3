"""worker of tests/test_gpu_two_devices.py: one process per GPU (rank r drives device r), the four sharded
transports across two PHYSICAL devices -- the hand-off no single-GPU box can exercise: fine-grained inbox
stores and system-scope mailbox records crossing xGMI (peer-to-peer), RCCL send/recv of whole particles."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mcl(kind, dist, rank, world, n, steps):
    from rust_robotics_amd.sharded import NativeShard, P2PShard, gloo_allgather, gloo_exchange
    from tests import helpers as H
    from tests.test_gpu_p2p import unsharded

    kw = dict(seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    if kind == "p2p":
        shard = P2PShard(rank, world, rank, n, **kw)
        shard.connect_ipc(gloo_allgather(dist))
    else:
        shard = NativeShard(rank, world, rank, n, gloo_exchange(dist), **kw)
    dist.barrier()
    rng = np.random.default_rng(43)
    if kind == "p2p":
        shard.want_estimate(True)  # every step leaves this shard's part of the mean try_step returns
    for t in range(steps):
        shard.step([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng))
    if kind == "p2p":
        assert not shard.timed_out(), "a peer wait timed out"
        # ... the shards' sums over N = the mean of the resampled set, whose slots came from both devices
        import torch

        sums, den = shard.estimate_sums()
        tot = torch.tensor(sums, dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        whole = unsharded(n * world, steps)
        np.testing.assert_allclose((tot / den).numpy(), whole[:, :4].mean(axis=0), rtol=1e-10, atol=1e-10)
    got = shard.particles()
    exp = unsharded(n * world, steps)[rank * n:(rank + 1) * n]
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), f"{kind} shard differs from the unsharded engine"
    dist.barrier()
    shard.close()


def fastslam(kind, dist, rank, world, n_local, steps):
    from rust_robotics_amd.sharded import gloo_allgather, gloo_exchange
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1
    from tests.test_gpu_fs1_sharded import SEED, check, scenario

    L, chunks = 7, 2
    prm, poses, maps, zs = scenario(n_local * world, L, steps)
    sl = slice(rank * n_local, (rank + 1) * n_local)
    shard = ShardedFastSlam1(rank, world, n_local, L, device=rank, params=prm, seed=SEED, obs_chunks=chunks)
    shard.set_state(poses[sl], maps[sl])
    if kind == "p2p":
        shard.connect_ipc(gloo_allgather(dist))
    else:
        shard.connect_rccl(gloo_exchange(dist))
    dist.barrier()
    moved = 0
    for z in zs:
        shard.update_async([1.0, 0.1], z)
        moved += shard.migrated() if kind == "rccl" else 0
    assert not shard.timed_out(), "a peer wait timed out"
    got = shard.get_state()
    states = [None] * world
    dist.all_gather_object(states, got)
    if rank == 0:
        check(states, n_local, L, steps, chunks)
        assert kind == "p2p" or moved > 0, "expected whole particles to cross devices"
    dist.barrier()
    shard.close()


def main():
    import torch.distributed as dist

    what, kind = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if what == "mcl":
        mcl(kind, dist, rank, world, 8000, 8)
    else:
        fastslam(kind, dist, rank, world, 3000, 8)
    dist.destroy_process_group()
    print("TWO_DEVICE_OK")


if __name__ == "__main__":
    main()
