#!/usr/bin/env python3
"""Soak of the in-step estimate in all its forms (run on a GPU box: python tools/soak_estimate.py [steps] [particles]).
For each scheme (systematic in-plan, systematic deferred, multinomial) and each filter kind (gated PF, fixed-N MCL): filter A takes
a random mix of plain asynchronous steps, estimate-producing asynchronous steps (read at once, read a step later, never read),
synchronous try_steps and accessors in between; filter B only ever takes plain asynchronous steps.  The particle sets must be
identical bit for bit at every checkpoint, every estimate A reports must agree with the accessor's over the same set (1e-11), and
a synchronous try_step must return the bits the deferred form reports for the same step (twin filter C)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402
import rust_robotics_amd.localization as loc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lms = H.landmarks_grid(16, 4)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def make(kind, scheme, seed):
    if kind == "mcl":
        return loc.MonteCarloLocalizer(loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5), seed=seed, resample_scheme=scheme)
    return loc.ParticleFilterLocalizer(loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, resample_threshold=0.5), seed=seed, resample_scheme=scheme)


def run(kind, scheme, defer):
    os.environ["RR_PF_EST_DEFER"] = "1" if defer else "0"
    a, b, c = make(kind, scheme, 5), make(kind, scheme, 5), make(kind, scheme, 5)
    rng, pick = np.random.default_rng(6), np.random.default_rng(7)
    pending_read = False
    n_sync = n_now = n_later = 0
    for t in range(steps):
        obs = H.observations(lms, H.true_pose(t + 1), 0.5, rng)
        u = [1.0, 0.1]
        b.step_async(u, obs)
        what = pick.integers(0, 6)
        if pending_read:  # the estimate of the step before, read after one more step has been enqueued
            pending_read = False
        if what == 0:
            a.step_async(u, obs)
            c.step_async(u, obs)
        elif what == 1:  # read at once
            a.step_async_estimate(u, obs)
            e = np.array(a.last_step_estimate())
            np.testing.assert_allclose(e, a.estimate(), rtol=1e-11, atol=1e-11)
            c.step_async(u, obs)
            n_now += 1
        elif what == 2:  # read a step later (the next iteration's step consumes the resample and sums on the way)
            a.step_async_estimate(u, obs)
            c.step_async_estimate(u, obs)
            ec = np.array(c.last_step_estimate())  # twin: read at once
            obs2 = H.observations(lms, H.true_pose(t + 1), 0.5, np.random.default_rng(10_000 + t))
            a.step_async(u, obs2)
            b.step_async(u, obs2)
            c.step_async(u, obs2)
            ea = np.array(a.last_step_estimate())
            assert np.array_equal(bits(ea), bits(ec)), f"{kind}/{scheme}/{defer} step {t}: later {ea} != at once {ec}"
            n_later += 1
        elif what == 3:  # never read
            a.step_async_estimate(u, obs)
            c.step_async(u, obs)
        elif what == 4:  # synchronous try_step against the asynchronous estimate of the twin
            ea = np.array(a.step(u, obs))
            c.step_async_estimate(u, obs)
            ec = np.array(c.last_step_estimate())
            if scheme == 0 or not defer:  # (systematic deferred: try_step uses the in-plan form, another order of summation)
                assert np.array_equal(bits(ea), bits(ec)), f"{kind}/{scheme}/{defer} step {t}: try_step {ea} != async {ec}"
            else:
                np.testing.assert_allclose(ea, ec, rtol=1e-11, atol=1e-11)
            n_sync += 1
        else:  # an accessor in between
            a.step_async_estimate(u, obs)
            a.calc_covariance()
            c.step_async(u, obs)
        if t % 97 == 96 or t == steps - 1:
            pa, pb, pc = a.get_particles_array(), b.get_particles_array(), c.get_particles_array()
            for k in range(5):
                assert np.array_equal(bits(pa[:, k]), bits(pb[:, k])) and np.array_equal(bits(pc[:, k]), bits(pb[:, k])), f"{kind}/{scheme}/{defer}: particles differ at step {t}"
    print(f"{kind:3s} scheme {scheme} deferred {int(defer)}: {steps} steps ok ({n_now} read at once, {n_later} a step later, {n_sync} synchronous)")


for kind in ("pf", "mcl"):
    for scheme, defer in ((1, False), (1, True), (0, True)):
        run(kind, scheme, defer)
print("SOAK_ESTIMATE_OK")
