"""The literal restatement of the reference (oracle/ref_literal.c: libm, no FMA, the reference's own operation order)
against the GPU engine AT THE BASELINE SIZES -- not on a sample: every particle of 1e6 x 32 and 2e6 x 64 (MCL, BASELINE.json
configs[1] and the per-GPU shape of configs[4]) and every (particle, landmark) pair of 1e5 x 200 and 1.25e5 x 200 (FastSLAM 1.0, configs[2] and the per-GPU shape of configs[3]).

The bit-exact tests elsewhere compare the kernels with oracle/det_spec.c, which compiles the same arithmetic header as the
kernels: they prove the plumbing.  These compare the ARITHMETIC with code that shares nothing with the kernels, after a few
real steps of a tracking filter, with the device's own motion noise handed to the literal code (the noise stream is the
D-spec's Philox/Box-Muller, reproduced on the host by det_normal2_v and checked bit for bit in test_gpu_pf_parity.py):

  propagate       particle_filter.rs:279-296   /  fastslam1.rs:123-137   rtol = atol = 1e-6 (the reference's gate convention)
  range weights   particle_filter.rs:310-329,476-479 (one exp per PAIR in the literal code, one per particle on the device)
  2x2 EKF         fastslam1.rs:140-183: landmark mean, covariance (4 entries), likelihood-accumulated weight
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-6, atol=1e-6)


def host_threads(ref):
    import os

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return ref.ref_set_threads(max(1, min(avail, 16)))


@pytest.mark.parametrize("n,L", [(1_000_000, 32), (2_000_000, 64)])
def test_mcl_every_particle_matches_the_literal_reference(det, ref, n, L):
    import rust_robotics_amd.localization as loc

    seed, sv, sw, sigma, dt = 1, 2.0, math.radians(40.0), 0.2, 0.1  # the bench configuration (defaults of particle_filter.rs:67-78)
    lms = H.landmarks_grid(L, 1)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=seed, resample_scheme=1)
    rng = np.random.default_rng(2)
    u = [1.0, 0.1]
    for t in range(4):  # real steps: the cloud and its weights are those of a tracking filter
        pf.step_async(u, H.observations(lms, H.true_pose(t + 1), sigma, rng))
    before = pf.get_particles_array()
    step = pf.counters()[0]
    # ---- propagate, particle_filter.rs:279-296, with the device's own noise
    pf.predict_with_control(u)
    after = pf.get_particles_array()
    z0, z1 = np.empty(n), np.empty(n)
    det.det_normal2_v(seed, 3, step, 0, n, dp(z0), dp(z1))
    nv, nw = sv * z0, sw * z1
    x, y, yaw, v = (np.ascontiguousarray(before[:, k]) for k in range(4))
    host_threads(ref)
    try:
        ref.ref_pf_predict(n, dp(x), dp(y), dp(yaw), dp(v), u[0], u[1], dt, dp(nv), dp(nw))
        np.testing.assert_allclose(after[:, :4], np.column_stack([x, y, yaw, v]), **TOL)
        # ---- weights, particle_filter.rs:310-329: raw (one factor per pair in the literal code), then normalised
        obs = np.ascontiguousarray(H.observations(lms, H.true_pose(5), sigma, rng))
        pf.update_with_observations(obs)
        raw = pf.raw_weights()
        gx, gy = np.ascontiguousarray(after[:, 0]), np.ascontiguousarray(after[:, 1])
        wr = np.empty(n)
        ref.ref_pf_update_raw(n, dp(gx), dp(gy), dp(wr), dp(obs), L, sigma)
    finally:
        ref.ref_set_threads(1)
    # The literal running product loses bits of its own only while a partial product is subnormal, and 1e-6 of them only below
    # 2^-1054 = 5e-318; the factors that follow are at most 1/(sigma sqrt(2 pi)) = 1.995 each, so a final weight above
    # 5e-318 * 1.995^63 = 3.9e-299 (L = 64; L = 32: 1e-308) has lost less than that.  Compared at 1e-6 down to 1e-290 (until
    # round 6: 1e-250).
    big = wr > 1e-290
    assert np.count_nonzero(big) > n // 2
    np.testing.assert_allclose(raw[big], wr[big], rtol=1e-6, atol=0.0)
    assert np.all(raw[~big] <= 1e-289)
    # ... and below it, down to the edge of the normal range, the two still agree in the logarithm (no garbage hides in the mask)
    tiny = (~big) & (wr > 1e-300) & (raw > 0.0)
    if np.any(tiny):
        assert np.max(np.abs(np.log(raw[tiny]) - np.log(wr[tiny]))) < 1e-5, "masked weights disagree"
    ref.ref_pf_normalize(n, dp(wr))  # the reference's serial left-to-right sum (particle_filter.rs:426-439)
    got = pf.get_particles_array()[:, 4]
    np.testing.assert_allclose(got, wr, rtol=1e-6, atol=1e-12)
    assert abs(got.sum() - 1.0) < 1e-9
    np.testing.assert_allclose(pf.n_eff(), ref.ref_pf_neff(n, dp(wr)), rtol=1e-6)
    est = np.empty(4)
    ref.ref_pf_estimate(n, dp(gx), dp(gy), dp(np.ascontiguousarray(after[:, 2])), dp(np.ascontiguousarray(after[:, 3])), dp(wr), dp(est))
    np.testing.assert_allclose(pf.estimate(), est, **TOL)


@pytest.mark.parametrize("n", [100_000, 125_000])
def test_fastslam_every_pair_matches_the_literal_reference(det, ref, n):
    """BASELINE.json configs[2] (100 000 particles x 200 landmarks) and the per-GPU shape of configs[3] (10^6 particles over 8
    GPUs: 125 000 x 200): every landmark observed, the EKF branch for every pair."""
    from rust_robotics_amd.slam import fastslam1 as fs

    L, seed = 200, 2
    lms = np.random.default_rng(seed).uniform(-13.0, 13.0, size=(L, 2))
    prm = fs.default_params()
    prm.first_obs_cov = 0.5
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=seed)
    u = [0.5, 0.1]

    def z_at(t):
        return np.ascontiguousarray(np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=seed, step=t)).reshape(-1, 3))

    for t in range(4):  # the first update initialises the maps, the others are EKF updates with data-dependent resampling
        f.update_async(u, z_at(t))
    poses, maps = f.get_state()
    assert np.all(maps[:, :, 2] < 100.0), "every landmark of every particle is initialised"
    step = f.counters()[0]
    # ---- predict, fastslam1.rs:123-137 with the device's own unit normals
    f.predict(u)
    got = f.poses()
    z0, z1 = np.empty(n), np.empty(n)
    det.det_normal2_v(seed, 3, step, 0, n, dp(z0), dp(z1))
    px, py, pyaw = (np.array(poses[:, k], dtype=np.float64, order="C", copy=True) for k in (1, 2, 3))
    mr = oracle.ref_fs1_model()
    mr.init_cov = 0.5
    host_threads(ref)
    try:
        ref.ref_fs1_predict(n, dp(px), dp(py), dp(pyaw), u[0], u[1], dp(z0), dp(z1), C.byref(mr))
        np.testing.assert_allclose(got[:, 1:], np.column_stack([px, py, pyaw]), **TOL)
        # ---- the EKF loop, fastslam1.rs:250-256 over :140-183, from the device's own predicted poses
        z = z_at(4)
        assert len(z) == L
        f.observe(z)
        gp, gm = f.get_state()
        pw = np.array(poses[:, 0], copy=True)
        lm = maps.reshape(-1).copy()
        qx, qy, qyaw = (np.ascontiguousarray(got[:, k]) for k in (1, 2, 3))
        ref.ref_fs1_observe(n, L, dp(qx), dp(qy), dp(qyaw), dp(pw), dp(lm), dp(z), len(z), C.byref(mr))
    finally:
        ref.ref_set_threads(1)
    np.testing.assert_allclose(gm.reshape(-1), lm, **TOL)  # 1.2e8 numbers: mean and covariance of every landmark of every particle
    # (one update multiplies at most 200 factors of at most 1 / (2 pi sqrt(det R)) = 1.29 into a weight: a result above
    # 5e-318 * 1.29^200 = 6.6e-296 has not lost 1e-6 of itself in a subnormal partial product)
    big = pw > 1e-290
    assert np.count_nonzero(big) > n // 2
    np.testing.assert_allclose(gp[big, 0], pw[big], rtol=1e-6, atol=0.0)
    tiny = (~big) & (pw > 1e-300) & (gp[:, 0] > 0.0)
    if np.any(tiny):
        assert np.max(np.abs(np.log(gp[tiny, 0]) - np.log(pw[tiny]))) < 1e-5, "masked weights disagree"
    print(f"weights compared at 1e-6: {np.count_nonzero(big)} of {n}; in the logarithm: {np.count_nonzero(tiny)}")
