"""debug: step time of in-process p2p shards (bench parameters)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from tests import helpers as H
from rust_robotics_amd.sharded import P2PShard
world, n_local, steps, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = dict(seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
shards = [P2PShard(g, world, 0, n_local, **kw) for g in range(world)]
P2PShard.link_local(shards)
lms = H.landmarks_grid(L, 1)
rng = np.random.default_rng(2)
obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(steps + 50)]
for t in range(50):
    for s in shards: s.step([1.0, 0.1], obs[t])
    if t % 5 == 4 or t < 5:
        try:
            for s in shards: s.synchronize()
        except Exception as e:
            print(f"world {world} n_local {n_local}: failed in warm-up step {t}: {str(e)[:80]}", [s.timed_out() for s in shards])
            sys.exit(0)
t0 = time.perf_counter()
for t in range(50, 50 + steps):
    for s in shards: s.step([1.0, 0.1], obs[t])
for s in shards: s.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e6
print(f"in-process world {world} n_local {n_local}: {dt:.1f} us/step (all shards on one device)", [s.timed_out() for s in shards])
