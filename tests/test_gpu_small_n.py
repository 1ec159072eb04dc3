"""Small particle sets (<= 2048 particles -- every caller in the reference runs 100 - 1200): the whole step, and K of them,
in ONE launch of one workgroup (k_step_small, rr_pf_step_many).  Must be bit-identical to the D-spec, hence to the large
kernels, for both resamplers, both gates and both likelihood forms, at every register layout (R = 1, 2, 4)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dp
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def make(loc, n, scheme, gated, lik=0, seed=9, record=True):
    kw = dict(seed=seed, resample_scheme=scheme, record_indices=record, likelihood_mode=lik)
    if gated:
        cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0), resample_threshold=0.5)
        return loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, **kw)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    return loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, **kw)


@pytest.mark.parametrize("n", [1, 7, 64, 100, 150, 512, 513, 1000, 1024, 1025, 2047, 2048])
@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("gated", [False, True])
def test_small_steps_bit_exact_vs_det(det, n, scheme, gated):
    import rust_robotics_amd.localization as loc

    pf = make(loc, n, scheme, gated)
    x, y, yaw, v = (np.empty(n) for _ in range(4))
    det.det_pf_init(n, 9, 0, dp(np.array([0.0, 0.0, 0.0, 1.0])), dp(x), dp(y), dp(yaw), dp(v))
    d = H.DetPF(det, x, y, yaw, v, dt=0.1, sigma=0.5, sigma_v=0.3, sigma_w=math.radians(5.0), threshold=0.5 if gated else 1.0,
                gate=0 if gated else 1, scheme=scheme, lik=0, seed=9)
    rng = np.random.default_rng(10)
    fired_any = False
    for t in range(8):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        if t % 3 == 0:
            est = pf.step([1.0, 0.1], obs)
        elif t % 3 == 1:
            pf.step_async([1.0, 0.1], obs)
            est = None
        else:
            pf.step_async_estimate([1.0, 0.1], obs)
            est = pf.last_step_estimate()
        fired = d.step([1.0, 0.1], obs)
        fired_any |= fired
        assert pf.last_resample_fired() == fired, f"gate differs at step {t}"
        if fired:
            assert np.array_equal(pf.last_resample_indices(), d.idx), f"indices differ at step {t}"
        if est is not None:
            if fired:
                want = np.array([d.x.mean(), d.y.mean(), d.yaw.mean(), d.v.mean()])
            else:
                wn = d.normalized_weights()
                want = np.array([(wn * a).sum() for a in (d.x, d.y, d.yaw, d.v)])
            np.testing.assert_allclose(est, want, rtol=1e-10, atol=1e-10)
        got = pf.get_particles_array()
        for k, e in enumerate((d.x, d.y, d.yaw, d.v)):
            assert np.array_equal(bits(got[:, k]), bits(e)), f"n={n} step {t} column {k}"
        if not fired:
            assert np.array_equal(bits(pf.raw_weights()), bits(d.w)), f"raw weights differ at step {t}"
    assert fired_any or gated


@pytest.mark.parametrize("n,L,scheme,lik", [(200, 4, 0, 0), (1000, 4, 1, 0), (1500, 40, 0, 1), (2048, 120, 1, 0), (300, 0, 1, 0)])
def test_step_many_equals_single_steps(n, L, scheme, lik):
    """K steps in one launch == K launches of one step == the large kernels (RR_PF_SMALL=0), bit for bit, estimates included
    (those to rounding: the large kernels add the offspring-weighted sum, the small one the resampled set)."""
    import rust_robotics_amd.localization as loc

    K = 23
    lms = H.landmarks_grid(max(L, 1), 3)[:L]
    rng = np.random.default_rng(4)
    obs = np.stack([H.observations(lms, H.true_pose(t + 1), 0.5, rng) if L else np.zeros((0, 3)) for t in range(K)])
    u = np.tile([1.0, 0.1], (K, 1))
    u[:, 0] += 0.01 * np.arange(K)  # the controls differ from step to step
    a = make(loc, n, scheme, gated=True, lik=lik)
    est_a = a.step_many(u, obs)
    b = make(loc, n, scheme, gated=True, lik=lik)
    est_b = np.array([b.step(u[t], obs[t]) for t in range(K)])
    os.environ["RR_PF_SMALL"] = "0"
    try:
        c = make(loc, n, scheme, gated=True, lik=lik)
    finally:
        del os.environ["RR_PF_SMALL"]
    for t in range(K):
        c.step_async(u[t], obs[t])
    pa, pb, pc = a.get_particles_array(), b.get_particles_array(), c.get_particles_array()
    assert np.array_equal(bits(pa), bits(pb)), "step_many differs from single small steps"
    assert np.array_equal(bits(pa[:, :4]), bits(pc[:, :4])), "the small kernel differs from the large kernels"
    np.testing.assert_allclose(pa[:, 4], pc[:, 4], rtol=0, atol=0)
    assert np.array_equal(bits(est_a), bits(est_b))
    assert a.counters() == b.counters() == c.counters() == (K, K)
    assert a.last_resample_fired() == c.last_resample_fired()
    if L:  # (without observations the weights stay uniform and the N_eff gate never opens: there are no indices)
        assert np.array_equal(a.last_resample_indices(), c.last_resample_indices())
    np.testing.assert_allclose(a.estimate(), c.estimate(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a.calc_covariance(), c.calc_covariance(), rtol=1e-9, atol=1e-12)
    # asynchronous form + accessors afterwards
    a.step_many(u[:5], obs[:5], estimates=False)
    for t in range(5):
        c.step_async(u[t], obs[t])
    assert np.array_equal(bits(a.get_particles_array()[:, :4]), bits(c.get_particles_array()[:, :4]))
    assert a.n_eff() == c.n_eff()


def test_step_many_validates_its_inputs():
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd.core import RoboticsError

    pf = make(loc, 100, 0, gated=True)
    obs = np.zeros((3, 2, 3))
    obs[:, :, 0] = 1.0
    with pytest.raises(RoboticsError):
        pf.step_many(np.array([[1.0, 0.1], [np.nan, 0.0], [1.0, 0.1]]), obs)
    obs[1, 0, 0] = -1.0
    with pytest.raises(RoboticsError):
        pf.step_many(np.tile([1.0, 0.1], (3, 1)), obs)
    assert pf.counters() == (0, 0)
    assert pf.step_many(np.zeros((0, 2)), np.zeros((0, 0, 3))).shape == (0, 4)


def test_reference_invariants_at_the_reference_sizes():
    """particle_filter.rs:611-623 (sum w = 1 +- 1e-3 after update), :626-636 (finite outputs), :639-646 (cov diag >= 0) through
    the small kernel at the sizes the reference's callers use."""
    import rust_robotics_amd.localization as loc

    for n in (100, 120, 150, 200):
        pf = make(loc, n, 0, gated=True)
        rng = np.random.default_rng(n)
        for t in range(40):
            e = pf.step([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng))
            assert np.all(np.isfinite(e))
        p = pf.get_particles_array()
        assert abs(p[:, 4].sum() - 1.0) < 1e-3 and np.all(np.isfinite(p))
        assert np.all(np.diag(pf.calc_covariance()) >= 0)
        assert np.hypot(*(pf.estimate()[:2] - H.true_pose(40)[:2])) < 2.0
