"""worker of tests/test_sharded_gloo.py: one rank of a gloo world stepping the sharded
localizer over the CPU stand-in backend; writes its final particle block to argv[1]."""
import math
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rust_robotics_amd.sharded import ShardedLocalizer  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.sharded_cpu_backend import CpuShard  # noqa: E402


def main():
    out_dir, n_local, steps, gate_always = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    scheme = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    shard = CpuShard(rank, world, n_local, seed=42, sigma=0.5, sigma_v=0.3, sigma_w=math.radians(5.0),
                     gate_always=bool(gate_always), threshold=0.9, scheme=scheme)
    loc = ShardedLocalizer(shard, dist)
    rng = np.random.default_rng(43)
    fired, moved = [], 0
    for t in range(steps):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        f = loc.step([1.0, 0.1], obs)
        fired.append(int(f))
        if f:
            moved += int(loc.last_matrix.sum() - np.trace(loc.last_matrix))
    est, cov = loc.estimate()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=shard.x, y=shard.y, yaw=shard.yaw, v=shard.v,
             w=shard.w, uniform=shard.uniform, fired=np.array(fired), moved=moved, est=est, cov=cov)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
