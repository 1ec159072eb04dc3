#!/usr/bin/env python3
"""Soak of the resident service: N steps of a PF, an adaptive MCL and a FastSLAM filter through their resident kernels with random
pauses (idle exits, relaunches), accessors thrown in, each shadowed by a launched twin; every estimate compared bit for bit.
    python tools/soak_resident.py [steps]"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_robotics_amd.localization as loc  # noqa: E402
from rust_robotics_amd.slam import fastslam1 as fs  # noqa: E402
from tests import helpers as H  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(1)
    cfg = loc.ParticleFilterConfig(n_particles=300, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    pf = [loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=3) for _ in range(2)]
    mc = [loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(), seed=4) for _ in range(2)]
    prm = fs.default_params()
    prm.first_obs_cov = 0.5
    sl = [fs.FastSlam1(100, 8, seed=5, params=prm) for _ in range(2)]
    pf[0].set_resident(700.0)
    mc[0].set_resident(700.0)
    sl[0].set_resident(700.0)
    lms = np.random.default_rng(3).uniform(-13, 13, size=(8, 2))
    t0 = time.time()
    for t in range(steps):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t % 400 + 1), 0.5, rng)
        a, b = pf[0].step([1.0, 0.1], obs), pf[1].step([1.0, 0.1], obs)
        assert np.array_equal(bits(a), bits(b)), f"PF estimate differs at step {t}"
        a, b = mc[0].try_step([1.0, 0.1], obs), mc[1].try_step([1.0, 0.1], obs)
        assert np.array_equal(bits(a), bits(b)), f"adaptive MCL estimate differs at step {t}"
        z = np.ascontiguousarray(np.array(fs.get_observations(H.true_pose(t % 400 + 1, v=0.5), [tuple(p) for p in lms], seed=5, step=t)).reshape(-1, 3))
        for f in sl:
            f.update([0.5, 0.1], z)
        ba, bb = sl[0].best_particle(), sl[1].best_particle()
        assert ba[2] == bb[2] and np.array_equal(bits(ba[0]), bits(bb[0])), f"FastSLAM best particle differs at update {t}"
        r = rng.random()
        if r < 0.01:
            time.sleep(0.002)  # longer than the idle time: the kernels leave
        elif r < 0.015:
            assert np.array_equal(bits(pf[0].get_particles_array()), bits(pf[1].get_particles_array()))
            assert mc[0].particle_count() == mc[1].particle_count()
            assert np.array_equal(bits(sl[0].get_state()[1]), bits(sl[1].get_state()[1]))
    print({"steps": steps, "seconds": round(time.time() - t0, 1), "pf": pf[0].resident_stats(), "mcl": mc[0].resident_stats(), "fastslam": sl[0].resident_stats()})


if __name__ == "__main__":
    main()
