"""The two CPU oracles against each other: the deterministic spec (oracle/det_spec.c, what the GPU
must match bit for bit) stays within the reference's own tolerance (rtol = atol = 1e-6,
scripts/check_benchmark_gate.py:34-35) of the literal restatement (oracle/ref_literal.c), and its
integer CDF selects the same particles as the reference's serial float cumsum except for draws
that land within rounding distance of a CDF step -- counted here, never hidden."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p, u64p
from tests import helpers as H

TOL = dict(rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,L", [(1000, 4), (5000, 32), (2000, 64)])
def test_weights_agree(det, ref, n, L):
    lms = H.landmarks_grid(L, 3)
    pose = H.true_pose(7)
    obs = H.observations(lms, pose, 0.2, np.random.default_rng(4))
    x, y, yaw, v = H.cloud(n, 5, center=(pose[0], pose[1], pose[2], 1.0))
    wr, wf, wp = np.empty(n), np.empty(n), np.empty(n)
    ref.ref_pf_update_raw(n, dp(x), dp(y), dp(wr), dp(obs), L, 0.2)
    det.det_pf_weights(n, dp(x), dp(y), dp(wf), dp(obs), L, 0.2, 0)
    det.det_pf_weights(n, dp(x), dp(y), dp(wp), dp(obs), L, 0.2, 1)
    big = wr > 1e-280
    assert big.sum() > n // 4
    np.testing.assert_allclose(wf[big], wr[big], rtol=1e-10)
    np.testing.assert_allclose(wp[big], wr[big], rtol=1e-11)
    # normalised weights, N_eff and the mean
    fx = H.det_fixed(det, wf)
    s = det.det_fix_total_to_double(fx["total"], fx["shift"])
    ref.ref_pf_normalize(n, dp(wr))
    np.testing.assert_allclose(wf / s, wr, **TOL)
    np.testing.assert_allclose(det.det_fix_neff(fx["total"], fx["q2_hi"], fx["q2_lo"]), ref.ref_pf_neff(n, dp(wr)), rtol=1e-6)


@pytest.mark.parametrize("n", [10_000, 200_000, 1_000_000])
def test_resample_index_agreement_at_scale(det, ref, n):
    """identical weights and identical draws into both CDFs: mismatches are the draws within the
    float cumsum's own accumulated rounding error (~1e-13) of a step; expected << 1 at 1e4 and
    O(0.1-1) at 1e6."""
    rng = np.random.default_rng(n)
    w = rng.random(n) ** 4 * np.exp(-rng.random(n) * 12)
    wn = w.copy()
    ref.ref_pf_normalize(n, dp(wn))
    fx = H.det_fixed(det, w)
    cdf = H.det_cdf(det, w, fx)
    r = np.floor(rng.random(n) * 2**53) / 2**53
    a, b = np.empty(n, np.uint32), np.empty(n, np.uint32)
    det.det_indices_multinomial(n, u64p(cdf), fx["total"], 0, n, dp(r), 0, 0, u32p(a))
    ref.ref_mcl_resample_indices(n, dp(wn), dp(r), u32p(b))
    mism_m = int(np.count_nonzero(a != b))
    rho = float(np.floor(rng.random() * 2**53) / 2**53)
    det.det_indices_systematic(n, u64p(cdf), fx["total"], n, 0, n, rho, u32p(a))
    ref.ref_fs1_resample_indices(n, dp(w.copy()), rho / n, u32p(b))
    mism_s = int(np.count_nonzero(a != b))
    assert np.all(np.abs(a.astype(np.int64) - b.astype(np.int64)) <= 1), "a differing draw must pick the neighbouring particle"
    budget = 0 if n <= 10_000 else 8
    assert mism_m <= budget and mism_s <= budget, (mism_m, mism_s)


def test_fastslam_update_agreement(det, ref):
    n, L = 300, 8
    rng = np.random.default_rng(9)
    lms = rng.uniform(-10, 10, (L, 2))
    mr, md = oracle.ref_fs1_model(), oracle.det_fs1_model()
    mr.init_cov = md.init_cov = 1.5
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    dx, dy, dyaw, dw = px.copy(), py.copy(), pyaw.copy(), pw.copy()
    planes = oracle.maps_aos_to_planes(lm, n, L)
    ia, ib = np.empty(n, np.uint32), np.empty(n, np.uint32)
    for t in range(10):
        xt = H.true_pose(t + 1)
        z = np.empty((L, 3))
        cnt = det.det_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), L, 20.0, 0.5, 0.0305, 21, t, dp(z))
        z = np.ascontiguousarray(z[:cnt])
        z0, z1 = np.empty(n), np.empty(n)
        det.det_normal2_v(21, 3, t, 0, n, dp(z0), dp(z1))
        rho = det.det_resample_rho(21, t)
        fa = det.det_fs1_update(n, L, dp(dx), dp(dy), dp(dyaw), dp(dw), dp(planes), 1.0, 0.1, dp(z), cnt, C.byref(md), n / 1.5, 21, t, t, 1, u32p(ia))
        fb = ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(z0), dp(z1), dp(z), cnt, C.byref(mr),
                                n / 1.5, rho / n, u32p(ib))
        assert fa == fb
        if fa:
            assert np.array_equal(ia, ib)
        np.testing.assert_allclose(dw, pw, **TOL)
        np.testing.assert_allclose(np.column_stack([dx, dy, dyaw]), np.column_stack([px, py, pyaw]), **TOL)
        np.testing.assert_allclose(oracle.maps_planes_to_aos(planes, n, L), lm, **TOL)


def test_fastslam_ekf_agreement_at_the_edges(det, ref):
    """One EKF update (fastslam1.rs:140-183) of the D-spec against the literal restatement over geometries that sit on the
    arithmetic's edges: bearings and headings next to +-pi (both angle wraps), landmarks centimetres to kilometres from the
    particle (the Jacobian's 1/d and 1/d^2), covariances from collapsed to just under the first-observation threshold and
    exactly on it (the branch test is `>`), innovations from zero to many sigmas (weights down to the underflow range)."""
    rng = np.random.default_rng(31)
    n = 40_000
    mr, md = oracle.ref_fs1_model(), oracle.det_fs1_model()
    thr = mr.init_threshold
    yaw = np.where(rng.random(n) < 0.3, np.sign(rng.standard_normal(n)) * (np.pi - 10.0 ** rng.uniform(-12, -1, n)), rng.uniform(-np.pi, np.pi, n))
    dist = 10.0 ** rng.uniform(-2, 3, n)
    bearing = np.where(rng.random(n) < 0.3, np.sign(rng.standard_normal(n)) * (np.pi - 10.0 ** rng.uniform(-12, -1, n)), rng.uniform(-np.pi, np.pi, n))
    px, py = rng.uniform(-50, 50, n), rng.uniform(-50, 50, n)
    lx, ly = px + dist * np.cos(yaw + bearing), py + dist * np.sin(yaw + bearing)
    scale = 10.0 ** rng.uniform(-12, np.log10(thr) - 1e-3, n)
    scale[rng.random(n) < 0.02] = thr          # exactly on the threshold: still the EKF branch
    scale[rng.random(n) < 0.02] = 0.0          # collapsed covariance
    rho = rng.uniform(-0.95, 0.95, n)
    c00, c11 = scale, scale * 10.0 ** rng.uniform(-2, 0, n)
    c01 = rho * np.sqrt(c00 * c11)
    sig = 10.0 ** rng.uniform(-3, 1.2, n)      # innovation in units of sqrt(R)
    zd = np.maximum(dist + sig * np.sqrt(mr.r00) * rng.standard_normal(n), 0.0)
    za = bearing + sig * np.sqrt(mr.r11) * rng.standard_normal(n)
    za = np.where(rng.random(n) < 0.5, za, (za + np.pi) % (2 * np.pi) - np.pi)  # wrapped or not: the update wraps the innovation anyway
    e_ref = np.column_stack([lx, ly, c00, c01, c01, c11]).copy()
    w_ref = np.ones(n)
    for p in range(n):
        w = C.c_double(1.0)
        ref.ref_fs1_update_landmark(px[p], py[p], yaw[p], C.byref(w), zd[p], za[p], dp(e_ref[p]), C.byref(mr))
        w_ref[p] = w.value
    # D-spec: one landmark, n particles, every particle its own observation -> n single-particle calls on plane views
    e_det = np.column_stack([lx, ly, c00, c01, c01, c11]).copy()
    w_det = np.ones(n)
    for p in range(n):
        planes = np.ascontiguousarray(e_det[p])
        pw = np.array([1.0])
        det.det_fs1_observe(1, dp(np.array([px[p]])), dp(np.array([py[p]])), dp(np.array([yaw[p]])), dp(pw), dp(planes),
                            dp(np.array([zd[p], za[p], 0.0])), 1, C.byref(md), 1)
        e_det[p], w_det[p] = planes, pw[0]
    assert np.all(np.isfinite(w_ref)) and np.all(np.isfinite(e_ref))
    big = w_ref > 1e-280
    assert np.count_nonzero(big) > n // 2 and np.count_nonzero(~big) > 0
    # The literal code's own conditioning sets the bar: its Kalman gain divides by det(S), and where the covariance is collapsed
    # relative to R the posterior loses digits in BOTH codes.  Relative 1e-6 on the weights that are not underflowing, and
    # 1e-6 of the entry's scale (distance for the mean, prior covariance for the covariance) on the landmark.
    np.testing.assert_allclose(w_det[big], w_ref[big], rtol=1e-6, atol=0.0)
    assert np.all(w_det[~big] <= 1e-279)
    np.testing.assert_allclose(e_det[:, :2], e_ref[:, :2], rtol=1e-6, atol=1e-6)
    cov_scale = np.maximum(c00, c11)[:, None]
    assert np.all(np.abs(e_det[:, 2:] - e_ref[:, 2:]) <= 1e-6 * cov_scale + 1e-300)


def test_pf_arithmetic_agreement_at_the_edges(det, ref):
    """particle_filter.rs:279-296 and :310-329 of the D-spec against the literal restatement where the elementary functions are
    stretched: headings that have grown for a long run (the reference never wraps a particle filter's yaw: |yaw| up to 2^30 rad,
    the bound up to which the D-spec's sine / cosine reduction is exact to the last bits -- include/rr_detmath.h; a filter turning at
    0.1 rad/s and stepping at 10 Hz gets there in 1e11 steps, three centuries), speeds and noises of either sign and any magnitude, ranges from a landmark sitting on
    the particle to kilometres, range noise from millimetres to tens of metres, innovations from zero to the underflow range."""
    rng = np.random.default_rng(37)
    n = 200_000
    yaw = np.clip(rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 9, n), -(2.0 ** 30) + 1.0, 2.0 ** 30 - 1.0)
    x, y = rng.uniform(-1e3, 1e3, n), rng.uniform(-1e3, 1e3, n)
    v = rng.standard_normal(n)
    nv, nw = rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n), rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 2, n)
    for u0, u1, dt in ((1.0, 0.1, 0.1), (-37.5, 6.0, 0.01), (0.0, 0.0, 1.0), (250.0, -90.0, 0.5)):
        rx, ry, ryaw, rv = x.copy(), y.copy(), yaw.copy(), v.copy()
        dx, dy, dyaw, dv = x.copy(), y.copy(), yaw.copy(), v.copy()
        ref.ref_pf_predict(n, dp(rx), dp(ry), dp(ryaw), dp(rv), u0, u1, dt, dp(nv), dp(nw))
        det.det_pf_predict(n, dp(dx), dp(dy), dp(dyaw), dp(dv), u0, u1, dt, dp(nv), dp(nw), 0, 0, 0, 0.0, 0.0)
        np.testing.assert_allclose(np.column_stack([dx, dy, dyaw, dv]), np.column_stack([rx, ry, ryaw, rv]), **TOL)
        # the step itself (what was added), not only the sum it disappears in: cos / sin of the large headings
        np.testing.assert_allclose(dx - x, rx - x, rtol=1e-6, atol=1e-9 * max(1.0, abs(u0) + 100.0) * dt)
        np.testing.assert_allclose(dy - y, ry - y, rtol=1e-6, atol=1e-9 * max(1.0, abs(u0) + 100.0) * dt)
    # weights: L observations per particle, each with its own distance error
    n = 20_000
    for L, sigma in ((1, 0.2), (4, 0.5), (32, 0.2), (64, 0.001), (16, 30.0)):
        px, py = rng.uniform(-100, 100, n), rng.uniform(-100, 100, n)
        lm = rng.uniform(-100, 100, (L, 2))
        lm[0] = (px[0], py[0])  # a landmark exactly on a particle: distance 0
        true_d = np.hypot(lm[:, 0] - 3.0, lm[:, 1] + 4.0)
        obs = np.ascontiguousarray(np.column_stack([np.maximum(true_d + sigma * rng.standard_normal(L) * 10.0 ** rng.uniform(-2, 1, L), 0.0), lm]))
        # half the cloud near the pose the observations were taken from (weights of ordinary size), half anywhere (underflow)
        px[: n // 2] = 3.0 + sigma * rng.standard_normal(n // 2)
        py[: n // 2] = -4.0 + sigma * rng.standard_normal(n // 2)
        wr, wf, wp = np.empty(n), np.empty(n), np.empty(n)
        ref.ref_pf_update_raw(n, dp(px), dp(py), dp(wr), dp(obs), L, sigma)
        det.det_pf_weights(n, dp(px), dp(py), dp(wf), dp(obs), L, sigma, 0)
        det.det_pf_weights(n, dp(px), dp(py), dp(wp), dp(obs), L, sigma, 1)
        # the literal running product has lost 1e-6 of itself only if a partial product fell below 2^-1054 = 5e-318 and the
        # remaining factors, each at most c = 1 / (sigma sqrt(2 pi)), brought it back: a final weight above 5e-318 max(1, c)^(L-1)
        # is compared at 1e-6 (never higher than the 1e-250 of rounds 1-5, never lower than 1e-290)
        c = 1.0 / (sigma * math.sqrt(2.0 * math.pi))
        thr = min(1e-250, max(1e-290, 10.0 * 5e-318 * max(1.0, c) ** (L - 1)))
        big = wr > thr
        assert np.count_nonzero(big) > 100, (L, sigma)
        np.testing.assert_allclose(wf[big], wr[big], rtol=1e-6)
        np.testing.assert_allclose(wp[big], wr[big], rtol=1e-6)
        assert np.all(wf[~big] <= 10.0 * thr) and np.all(wp[~big] <= 10.0 * thr)


def test_fastslam_predict_agreement_at_the_edges(det, ref):
    """fastslam1.rs:123-137 (+ :70-89: the motion model and normalize_angle) of the D-spec against the literal restatement:
    headings on and next to +-pi, yaw increments that overshoot the wrap by several turns, controls of either sign."""
    rng = np.random.default_rng(41)
    n = 100_000
    mr, md = oracle.ref_fs1_model(), oracle.det_fs1_model()
    yaw = rng.uniform(-np.pi, np.pi, n)
    edge = rng.random(n) < 0.4
    yaw[edge] = np.sign(rng.standard_normal(edge.sum())) * (np.pi - 10.0 ** rng.uniform(-15, -1, edge.sum()))
    yaw[:4] = [np.pi, -np.pi, np.nextafter(np.pi, 0.0), np.nextafter(-np.pi, 0.0)]
    px, py = rng.uniform(-200, 200, n), rng.uniform(-200, 200, n)
    z0, z1 = rng.standard_normal(n), rng.standard_normal(n) * 10.0 ** rng.uniform(-2, 2, n)  # up to ~100 sigma of yaw-rate noise
    for u0, u1 in ((1.0, 0.1), (-3.0, 25.0), (0.0, 0.0), (40.0, -300.0)):
        rx, ry, ryaw = px.copy(), py.copy(), yaw.copy()
        dx, dy, dyaw = px.copy(), py.copy(), yaw.copy()
        ref.ref_fs1_predict(n, dp(rx), dp(ry), dp(ryaw), u0, u1, dp(z0), dp(z1), C.byref(mr))
        det.det_fs1_predict(n, dp(dx), dp(dy), dp(dyaw), u0, u1, dp(z0), dp(z1), 0, 0, 0, C.byref(md))
        np.testing.assert_allclose(np.column_stack([dx, dy]), np.column_stack([rx, ry]), **TOL)
        # the wrapped heading: equal as angles (a result within rounding of +-pi may come out on either side of the cut)
        d = np.abs(dyaw - ryaw)
        d = np.minimum(d, 2 * np.pi - d)
        assert d.max() <= 1e-9, d.max()
        assert np.all(np.abs(dyaw) <= np.pi + 1e-12) and np.all(np.abs(ryaw) <= np.pi + 1e-12)
