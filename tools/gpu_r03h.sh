cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kld_adaptive.py tests/test_gpu_c_abi.py tests/test_golden.py -q 2>&1 | tail -15
python - <<'PY'
import sys, time, math
sys.path.insert(0, '.')
import numpy as np
import rust_robotics_amd.localization as loc
from tests import helpers as H
cfg = loc.MonteCarloLocalizationConfig(min_particles=100, max_particles=5000)
rng = np.random.default_rng(1)
obs = [H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.2, rng) for t in range(400)]
for name in ("step_async", "try_step"):
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=3)
    f = getattr(pf, name)
    for t in range(100): f([1.0, 0.1], obs[t])
    pf.synchronize()
    t0 = time.perf_counter()
    for t in range(100, 400): f([1.0, 0.1], obs[t])
    pf.synchronize()
    print("adaptive MCL 100..5000 particles,", name, "us/step %.1f" % ((time.perf_counter() - t0) / 300 * 1e6), "count", pf.particle_count())
PY
