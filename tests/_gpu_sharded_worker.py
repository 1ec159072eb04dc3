"""worker of tests/test_gpu_sharded.py: NCCL world (size 1 on the gpurun box) stepping the real
HipShard through ShardedLocalizer; compares against the unsharded engine in the same process."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import HipShard, ShardedLocalizer
    from tests import helpers as H

    n, steps = int(sys.argv[1]), int(sys.argv[2])
    scheme = int(sys.argv[3]) if len(sys.argv) > 3 else _ffi.RR_RESAMPLE_SYSTEMATIC  # 0 = multinomial shards
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    rank, world = dist.get_rank(), dist.get_world_size()
    shard = HipShard(rank, world, 0, n, seed=42, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0), scheme=scheme)
    sl = ShardedLocalizer(shard, dist)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n * world, max_particles=n * world, range_noise=0.5, velocity_noise=0.3,
                                           yaw_rate_noise=math.radians(5.0))
    ref = loc.MonteCarloLocalizer(cfg, seed=42, resample_scheme=scheme)
    rng = np.random.default_rng(43)
    for t in range(steps):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        assert sl.step([1.0, 0.1], obs)
        ref.step([1.0, 0.1], obs)
    got = shard.particles()
    exp = ref.get_particles_array()[rank * n:(rank + 1) * n]
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), "sharded particles differ from the unsharded engine"
    est, cov = sl.estimate()
    np.testing.assert_allclose(est, ref.estimate(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cov, ref.calc_covariance(), rtol=1e-7, atol=1e-9)
    shard.close()
    # the native step (RCCL called from inside the library) must give the same particles
    from rust_robotics_amd.sharded import NativeShard, gloo_exchange

    nat = NativeShard(rank, world, 0, n, gloo_exchange(dist), seed=42, range_noise=0.5, velocity_noise=0.3,
                      yaw_rate_noise=math.radians(5.0), scheme=scheme)
    rng = np.random.default_rng(43)
    for t in range(steps):
        nat.step([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng))
    got = nat.particles()
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), "native sharded particles differ from the unsharded engine"
    e2, c2 = nat.estimate()
    np.testing.assert_allclose(e2, ref.estimate(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(c2, ref.calc_covariance(), rtol=1e-7, atol=1e-9)
    nat.close()
    dist.destroy_process_group()
    print("SHARDED_OK")


if __name__ == "__main__":
    main()
