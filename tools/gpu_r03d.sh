cd $GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, time, math, os
sys.path.insert(0, '.')
import numpy as np
import rust_robotics_amd.localization as loc
from tests import helpers as H
n, L, K = 1000, 4, 2000
cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
rng = np.random.default_rng(42)
obs = np.stack([H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng) for t in range(K)])
u = np.tile([1.0, 0.1], (K, 1))
def run(est):
    pf = loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 0.0], cfg, seed=42)
    pf.step_many(u[:64], obs[:64], estimates=est)
    pf.synchronize()
    t0 = time.perf_counter()
    pf.step_many(u, obs, estimates=est)
    pf.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
print(os.environ.get("RR_PF_SMALL_DBG"), "no-est-arg", run(False), "est-arg", run(True), run(True))
PY
python /tmp/t.py
RR_PF_SMALL_DBG=1 python /tmp/t.py
