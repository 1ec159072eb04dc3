"""worker of tests/test_gpu_p2p.py::test_two_processes_over_ipc_handles: one process per shard,
both on device 0, peers mapped through hipIpc handles exchanged over gloo."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from rust_robotics_amd.sharded import P2PShard, gloo_allgather
    from tests import helpers as H
    from tests.test_gpu_p2p import unsharded

    n, steps = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    shard = P2PShard(rank, world, 0, n, seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    shard.connect_ipc(gloo_allgather(dist))
    dist.barrier()
    rng = np.random.default_rng(43)
    for t in range(steps):
        shard.step([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng))
    assert not shard.timed_out(), "a peer wait timed out"
    got = shard.particles()
    exp = unsharded(n * world, steps)[rank * n:(rank + 1) * n]
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), "p2p shard differs from the unsharded engine"
    dist.barrier()
    shard.close()
    dist.destroy_process_group()
    print("P2P_OK")


if __name__ == "__main__":
    main()
