"""worker of tests/test_gpu_p2p.py::test_two_processes_over_ipc_handles and tests/test_gpu_world8.py: one process per shard,
all on device 0, peers mapped through hipIpc handles exchanged over gloo.
    argv: n_local steps.  RR_WORKER_PEAKED=<L>: the bench scene (L landmarks on the seeded grid, ParticleFilterConfig's
    default noises, initial state (0, 0, 0, 1)) instead of the reference's 4-landmark scene."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard, gloo_allgather
    from tests import helpers as H
    from tests.test_gpu_p2p import unsharded

    n, steps = int(sys.argv[1]), int(sys.argv[2])
    peaked = int(os.environ.get("RR_WORKER_PEAKED", "0"))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if peaked:
        lms, sigma = H.landmarks_grid(peaked, 2 if peaked == 64 else 1), 0.2
        shard = P2PShard(rank, world, 0, n, seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
    else:
        lms, sigma = H.REF_SCENE_LANDMARKS, 0.5
        shard = P2PShard(rank, world, 0, n, seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    shard.connect_ipc(gloo_allgather(dist))
    dist.barrier()
    rng = np.random.default_rng(43)
    obs = [H.observations(lms, H.true_pose(t + 1), sigma, rng) for t in range(steps)]
    for t in range(steps):
        shard.step([1.0, 0.1], obs[t])
    assert not shard.timed_out(), "a peer wait timed out"
    got = shard.particles()
    if rank == 0:
        print("P2P_TOPOLOGY", shard.topology(), flush=True)
    if os.environ.get("RR_P2P_CU_PARTITION", "0") not in ("", "0") and world > 1:
        assert shard.topology()["cu_partition_cus"] > 0 and shard.topology()["last_step"] == "lazy", shard.topology()
    dist.barrier()  # every rank's shard is done before anybody puts the unsharded filter of ALL particles beside it
    if peaked:
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n * world, max_particles=n * world)
        whole = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
        for t in range(steps):
            whole.step_async([1.0, 0.1], obs[t])
        exp = whole.get_particles_array()[rank * n:(rank + 1) * n]
        del whole
    else:
        exp = unsharded(n * world, steps)[rank * n:(rank + 1) * n]
    assert np.array_equal(got.view(np.uint64), np.ascontiguousarray(exp).view(np.uint64)), "p2p shard differs from the unsharded engine"
    dist.barrier()
    shard.close()
    dist.destroy_process_group()
    print("P2P_OK")


if __name__ == "__main__":
    main()
