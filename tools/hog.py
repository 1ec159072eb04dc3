"""debug load: an unsharded 1e6 x 32 MCL filter stepping for argv[1] seconds; prints steps and plan statistics"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import rust_robotics_amd.localization as loc
from tests import helpers as H
n, L = 1_000_000, 32
cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
lms = H.landmarks_grid(L, 1)
rng = np.random.default_rng(2)
obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(50)]
t0 = time.time()
k = 0
while time.time() - t0 < float(sys.argv[1]):
    for o in obs:
        pf.step_async([1.0, 0.1], o)
    pf.synchronize()
    k += 50
print("hog", sys.argv[2], "steps", k, "plan stats", pf.plan_stats(), flush=True)
