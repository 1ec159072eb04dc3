"""The lazy multinomial resample of a single-GPU filter (monte_carlo_localization.rs:322-365, particle_filter.rs:441-473): the
plan kernels leave the CDF and the guide table; the draws and their search run inside the NEXT step's k_step_lazy<kSrcDraw>
when steps follow each other, or in k_resample_guide_mn when an accessor comes first.  Both routes, and the route with the
search always a launch of its own (RR_MN_DEFER=0), must give the same particles, weights and source indices, bit for bit."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def make(loc, n, gated, record):
    kw = dict(seed=11, resample_scheme=0, record_indices=record)
    if gated:
        cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0), resample_threshold=0.5)
        return loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, **kw)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    return loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, **kw)


def run(n, L, T, gated, read_every, record):
    import rust_robotics_amd.localization as loc

    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(5)
    pf = make(loc, n, gated, record)
    out = []
    for t in range(T):
        pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.5, rng))
        if (t + 1) % read_every == 0:
            out.append((t, pf.get_particles_array().copy(), pf.last_resample_indices().copy() if record else None))
    out.append((T, pf.get_particles_array().copy(), None))
    return out


@pytest.mark.parametrize("n,L", [(5_000, 6), (200_000, 16)])
@pytest.mark.parametrize("gated", [False, True])
def test_search_inside_the_next_step_equals_the_search_kernel(n, L, gated):
    T = 24
    burst = run(n, L, T, gated, read_every=8, record=True)   # 8 steps in a row: the step kernel draws for itself
    single = run(n, L, T, gated, read_every=1, record=True)  # an accessor after every step: k_resample_guide_mn
    by_step = {t: (p, i) for t, p, i in single}
    for t, p, i in burst:
        ps, isg = by_step[t]
        assert np.array_equal(bits(p), bits(ps)), f"particles differ after step {t}"
        if i is not None and isg is not None:
            assert np.array_equal(i, isg), f"source indices differ after step {t}"
    # a filter that does not record indices takes the same route
    plain = run(n, L, T, gated, read_every=8, record=False)
    assert np.array_equal(bits(plain[-1][1]), bits(burst[-1][1]))


def test_deferral_can_be_switched_off():
    """RR_MN_DEFER=0 (read when the handle is created): every resample's search is a launch of its own; same bits."""
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
            "from tests.test_gpu_multinomial_lazy import run\n"
            "r = run(60_000, 8, 20, False, 5, True)\n"
            "h = hashlib.sha256()\n"
            "for t, p, i in r: h.update(np.ascontiguousarray(p).tobytes()); h.update(b'' if i is None else np.ascontiguousarray(i).tobytes())\n"
            "print(h.hexdigest())\n") % ROOT
    digests = []
    for defer in ("1", "0"):
        env = dict(os.environ, RR_MN_DEFER=defer)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(r.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1]
