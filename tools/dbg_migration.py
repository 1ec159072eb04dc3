"""How many particles cross a shard boundary per step in the bench configuration (loopback seam, one GPU)?"""
import sys
sys.path.insert(0, '.')
import numpy as np
from rust_robotics_amd.sharded import LocalWindowShards
from tests import helpers as H
world, n_local, steps, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sh = LocalWindowShards(world, n_local, seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
lms = H.landmarks_grid(L, 1)
rng = np.random.default_rng(2)
mig = []
for t in range(steps):
    sh.step([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
    mig.append(sh.migrated())
mig = np.array(mig)
print(f"world {world} n_local {n_local} L {L}: migrated per step: first 5 {mig[:5].tolist()}, median {int(np.median(mig))}, p90 {int(np.percentile(mig, 90))}, max {mig.max()}  (of {world * n_local})")
