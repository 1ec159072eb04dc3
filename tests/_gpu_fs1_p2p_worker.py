"""worker of tests/test_gpu_fs1_sharded.py::test_two_processes_over_ipc_handles: one process per
shard, both on device 0, peers mapped through hipIpc handles exchanged over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from rust_robotics_amd.sharded import gloo_allgather
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1
    from tests.test_gpu_fs1_sharded import SEED, check, scenario

    n_local, steps = int(sys.argv[1]), int(sys.argv[2])
    L, chunks = 7, 2
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    prm, poses, maps, zs = scenario(n_local * world, L, steps)
    sl = slice(rank * n_local, (rank + 1) * n_local)
    shard = ShardedFastSlam1(rank, world, n_local, L, params=prm, seed=SEED, obs_chunks=chunks)
    shard.set_state(poses[sl], maps[sl])
    shard.connect_ipc(gloo_allgather(dist))
    dist.barrier()
    for z in zs:
        shard.update_async([1.0, 0.1], z)
    assert not shard.timed_out(), "a peer wait timed out"
    got = shard.get_state()
    states = [None] * world
    dist.all_gather_object(states, got)
    if rank == 0:
        check(states, n_local, L, steps, chunks)
    dist.barrier()
    del shard
    dist.destroy_process_group()
    print("FS1_P2P_OK")


if __name__ == "__main__":
    main()
