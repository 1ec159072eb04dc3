"""Soak of the two multinomial searches (run on a GPU box: python tools/soak_mn.py [steps] [particles]): two MCL filters
and two gated particle filters of the same seed, one searching through the guide table (default), one through the
coarse table of the CDF (RR_MN_GUIDE=0, read at a filter's first multinomial resample), stepped `steps` times back to
back; the particle sets and the last resample's indices must be identical bit for bit at several checkpoints."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402
import rust_robotics_amd.localization as loc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lms = H.landmarks_grid(32, 5)


def make(kind, guide):
    os.environ["RR_MN_GUIDE"] = "1" if guide else "0"
    kw = dict(seed=5, resample_scheme=0, record_indices=True)
    if kind == "mcl":
        f = loc.MonteCarloLocalizer(loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5), **kw)
    else:
        f = loc.ParticleFilterLocalizer(loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, resample_threshold=0.5), **kw)
    rng = np.random.default_rng(6)
    f.step_async([1.0, 0.1], H.observations(lms, H.true_pose(1), 0.5, rng))  # the first resample reads the switch
    f.synchronize()
    return f, rng


for kind in ("mcl", "pf"):
    (a, ra), (b, rb) = make(kind, True), make(kind, False)
    fired = 0
    for t in range(1, steps):
        pose = H.true_pose(t + 1)
        a.step_async([1.0, 0.1], H.observations(lms, pose, 0.5, ra))
        b.step_async([1.0, 0.1], H.observations(lms, pose, 0.5, rb))
        if t % (steps // 5) == 0 or t == steps - 1:
            fa, fb = a.last_resample_fired(), b.last_resample_fired()
            assert fa == fb, (kind, t)
            fired += int(fa)
            pa, pb = a.get_particles_array(), b.get_particles_array()
            assert np.array_equal(pa.view(np.uint64), pb.view(np.uint64)), (kind, t, "particles differ")
            if fa:
                assert np.array_equal(a.last_resample_indices(), b.last_resample_indices()), (kind, t, "indices differ")
    print(f"{kind}: {steps} steps of {n} particles identical at every checkpoint ({fired} of them right after a resample)", flush=True)
