"""FastSLAM 2.0 on the CPU: the literal restatement of fastslam2.rs (libm, no FMA, nalgebra's
evaluation order) against the D-spec (include/rr_pf_spec.h "FastSLAM 2.0 proposal") on identical
inputs, numpy as an independent check of the linear algebra, and the reference's own unit tests
(fastslam2.rs:431-545) re-expressed against the restatement."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p

TOL = dict(rtol=1e-6, atol=1e-6)


def random_case(rng, initialised=True):
    pose = np.array([rng.normal(0, 2), rng.normal(0, 2), rng.uniform(-3, 3)])
    lm_xy = pose[:2] + rng.uniform(2, 12) * np.array([math.cos(a := rng.uniform(-3, 3)), math.sin(a)])
    c = rng.uniform(0.05, 3.0) if initialised else 1000.0
    lm = np.array([lm_xy[0], lm_xy[1], c, rng.normal(0, 0.01), rng.normal(0, 0.01), c * rng.uniform(0.5, 1.5)])
    u = (rng.uniform(0.2, 1.5), rng.uniform(-0.3, 0.3))
    d = math.hypot(*(lm_xy - pose[:2])) + rng.normal(0, 0.3)
    ang = math.atan2(lm_xy[1] - pose[1], lm_xy[0] - pose[0]) - pose[2] + rng.normal(0, 0.1)
    return pose, lm, u, d, ang


def test_proposal_literal_det_numpy(det, ref):
    rng = np.random.default_rng(3)
    m = oracle.det_fs2_model()
    for k in range(300):
        pose, lm, u, d, ang = random_case(rng, initialised=k % 7 != 0)
        md, cd, ml, cl = np.empty(3), np.empty(9), np.empty(3), np.empty(9)
        det.det_fs2_proposal(dp(pose), u[0], u[1], d, ang, dp(lm), C.byref(m), dp(md), dp(cd))
        ref.ref_fs2_proposal(dp(pose), u[0], u[1], d, ang, dp(lm), 0.5, 0.0305, dp(ml), dp(cl))
        np.testing.assert_allclose(md, ml, **TOL)
        np.testing.assert_allclose(cd, cl, **TOL)
        # independent evaluation of :173-216
        yaw = pose[2]
        xp = np.array([pose[0] + u[0] * 0.1 * math.cos(yaw), pose[1] + u[0] * 0.1 * math.sin(yaw), yaw + u[1] * 0.1])
        g = np.array([[1, 0, -u[0] * 0.1 * math.sin(yaw)], [0, 1, u[0] * 0.1 * math.cos(yaw)], [0, 0, 1.0]])
        P = g @ np.diag([0.1, 0.1, 0.01]) @ g.T
        if not lm[2] < 100.0:
            np.testing.assert_allclose(cl.reshape(3, 3), P, rtol=1e-9, atol=1e-12)
            continue
        dx, dy = lm[0] - xp[0], lm[1] - xp[1]
        d2 = dx * dx + dy * dy
        dd = math.sqrt(d2)
        H = np.array([[-dx / dd, -dy / dd, 0], [dy / d2, -dx / d2, -1.0]])
        Hl = np.array([[dx / dd, dy / dd], [-dy / d2, dx / d2]])
        Cm = np.array([[lm[2], lm[4]], [lm[3], lm[5]]])
        Qi = np.linalg.inv(Hl @ Cm @ Hl.T + np.diag([0.5, 0.0305]))
        Pp = np.linalg.inv(np.linalg.inv(P) + H.T @ Qi @ H)
        na = lambda a: (a + math.pi) % (2 * math.pi) - math.pi  # noqa: E731
        innov = np.array([d - dd, na(ang - na(math.atan2(dy, dx) - xp[2]))])
        np.testing.assert_allclose(cl.reshape(3, 3), Pp, rtol=1e-7, atol=1e-10)
        np.testing.assert_allclose(ml, xp + Pp @ H.T @ Qi @ innov, rtol=1e-7, atol=1e-9)


def test_sample_cholesky_and_fallback(det, ref):
    rng = np.random.default_rng(4)
    for k in range(200):
        A = rng.normal(size=(3, 3))
        cov = A @ A.T * 0.05 + np.eye(3) * 1e-3
        if k % 5 == 0:
            cov[2, 2] = -0.1  # not positive definite: :227-233 diagonal fallback
        if k % 11 == 0:
            cov[0, 0] = 0.0
        mean, z = rng.normal(size=3), rng.normal(size=3)
        od, ol = np.empty(3), np.empty(3)
        c = np.ascontiguousarray(cov.reshape(-1))
        det.det_fs2_sample(dp(mean), dp(c), dp(z), dp(od))
        ref.ref_fs2_sample(dp(mean), dp(c), dp(z), dp(ol))
        np.testing.assert_allclose(od, ol, **TOL)
        try:
            Lc = np.linalg.cholesky(cov)
        except np.linalg.LinAlgError:
            Lc = np.diag(np.sqrt(np.maximum(np.diag(cov), 0.0)))
        exp = mean + Lc @ z
        exp[2] = (exp[2] + math.pi) % (2 * math.pi) - math.pi
        np.testing.assert_allclose(ol, exp, rtol=1e-9, atol=1e-9)


def run_both(det, ref, n, L, T, seed, n_obs):
    rng = np.random.default_rng(seed)
    lms = rng.uniform(-10, 10, size=(L, 2))
    m = oracle.det_fs2_model()
    # literal: AoS maps; det: planes
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 1.0 / n)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    qx, qy, qyaw, qw = px.copy(), py.copy(), pyaw.copy(), pw.copy()
    planes = oracle.maps_aos_to_planes(lm.copy(), n, L)
    idx_l, idx_d = np.empty(n, np.uint32), np.empty(n, np.uint32)
    xt = np.zeros(3)
    fired_log = []
    for t in range(T):
        xt = np.array([xt[0] + 0.1 * math.cos(xt[2]), xt[1] + 0.1 * math.sin(xt[2]), xt[2] + 0.01])
        z = []
        for l in range(L if t != 3 else 0)[:n_obs]:
            dx, dy = lms[l] - xt[:2]
            z.append((math.hypot(dx, dy) + rng.normal(0, 0.3), math.atan2(dy, dx) - xt[2] + rng.normal(0, 0.05), float(l)))
        z = np.ascontiguousarray(np.array(z, dtype=np.float64).reshape(-1, 3))
        noise = np.ascontiguousarray(rng.normal(size=(n, 3)))
        rho = float(np.floor(rng.random() * 2**53) / 2**53)
        f_l = ref.ref_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(noise), dp(z) if len(z) else None, len(z),
                                 n / 1.5, rho / n, u32p(idx_l))
        f_d = det_update_with_rho(det, n, L, qx, qy, qyaw, qw, planes, z, m, noise, n / 1.5, rho, idx_d)
        assert bool(f_l) == bool(f_d), f"gate differs at step {t}"
        if f_l:
            assert np.array_equal(idx_l, idx_d), f"indices differ at step {t}"
        fired_log.append(int(f_l))
        for a, b in ((px, qx), (py, qy), (pyaw, qyaw), (pw, qw)):
            np.testing.assert_allclose(a, b, **TOL)
        np.testing.assert_allclose(lm, oracle.maps_planes_to_aos(planes, n, L), **TOL)
    return fired_log, (px, py, pyaw, pw, lm), lms, xt


def det_update_with_rho(det, n, L, qx, qy, qyaw, qw, planes, z, m, noise, nth, rho, idx):
    """det_fs2_update, but with a caller-supplied systematic offset (the D-spec entry point takes it
    from its Philox stream): predict + observe + gate + indices + gather out of its parts"""
    det.det_fs2_predict(n, dp(qx), dp(qy), dp(qyaw), dp(planes), 1.0, 0.1, dp(z) if len(z) else None, len(z), dp(noise), 0, 0, 0,
                        C.byref(m))
    det.det_fs1_observe(n, dp(qx), dp(qy), dp(qyaw), dp(qw), dp(planes), dp(z) if len(z) else None, len(z), C.byref(m.base), 1)
    from tests import helpers as H

    fx = H.det_fixed(det, qw.copy())
    total = int(H.det_cdf(det, qw.copy(), fx)[-1])
    s = det.det_fix_total_to_double(total, fx["shift"])
    neff = det.det_fix_neff(total, fx["q2_hi"], fx["q2_lo"])
    if neff < nth:
        cdf = H.det_cdf(det, qw.copy(), fx)
        det.det_indices_systematic(n, oracle.u64p(cdf), total, n, 0, n, rho, u32p(idx))
        P = planes.reshape(L * 6, n)
        P[:] = P[:, idx]
        for a in (qx, qy, qyaw):
            a[:] = a[idx]
        qw[:] = 1.0 / n
        return 1
    qw /= s
    return 0


@pytest.mark.parametrize("n,L,n_obs,seed", [(300, 4, 4, 1), (500, 6, 3, 2)])
def test_update_trajectory_literal_vs_det(det, ref, n, L, n_obs, seed):
    fired, _, _, _ = run_both(det, ref, n, L, 10, seed, n_obs)
    assert any(fired) and not all(fired), fired


# ---- the reference's own tests, fastslam2.rs:431-545, on the literal restatement
def test_reference_proposal_improves_with_observation(ref):
    pose = np.zeros(3)
    lm = np.array([5.0, 0.0, 0.5, 0.0, 0.0, 0.5])
    mean, cov = np.empty(3), np.empty(9)
    ref.ref_fs2_proposal(dp(pose), 1.0, 0.0, 5.0, 0.0, dp(lm), 0.5, 0.0305, dp(mean), dp(cov))
    g = np.array([[1, 0, 0.0], [0, 1, 0.1], [0, 0, 1.0]])
    assert np.linalg.det(cov.reshape(3, 3)) < np.linalg.det(g @ np.diag([0.1, 0.1, 0.01]) @ g.T)
    assert np.linalg.norm(mean - np.array([0.1, 0.0, 0.0])) < 1.0


def test_reference_landmark_convergence(ref):
    n, L = 120, 1
    rng = np.random.default_rng(17)
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 1.0 / 100)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    idx = np.empty(n, np.uint32)
    xt = np.array([0.0, 0.0, math.pi / 4])
    for _ in range(60):
        xt = np.array([xt[0] + 0.5 * 0.1 * math.cos(xt[2]), xt[1] + 0.5 * 0.1 * math.sin(xt[2]), xt[2]])
        dx, dy = 5.0 - xt[0], 5.0 - xt[1]
        z = np.array([[math.hypot(dx, dy) + rng.normal() * math.sqrt(0.5), math.atan2(dy, dx) - xt[2] + rng.normal() * math.sqrt(0.0305), 0.0]])
        ref.ref_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 0.5, 0.0, dp(np.ascontiguousarray(rng.normal(size=(n, 3)))),
                           dp(z), 1, 100 / 1.5, rng.random() / n, u32p(idx))
    e = lm.reshape(n, 6)
    init = e[:, 2] < 100.0
    assert init.any()
    w = pw[init]
    mx, my = (np.average(e[init, 0], weights=w), np.average(e[init, 1], weights=w)) if w.sum() > 0 else (e[init, 0].mean(), e[init, 1].mean())
    assert math.hypot(mx - 5.0, my - 5.0) < 6.0


def test_reference_seeded_tests_replayed_through_the_literal_restatement():
    """fastslam2.rs:443-456 (StdRng seed 7: 20 particles, 3 landmarks, 5 updates, the truth standing still) and :491-545 (StdRng
    seed 17: 120 particles, one landmark at (5, 5), 60 updates along a straight line, `lm_err < 6.0`) with the reference's own
    random stream (oracle/rand_rs.py: rand 0.9 StdRng, rand_distr 0.5.1 StandardNormal / Uniform restated and checked against their
    published vectors in tests/test_rand_port.py), consumed in the reference's order (tests/fs2_replay.py)."""
    from tests import fs2_replay as RP

    e = RP.LiteralEngine(20, 3)
    fired, _ = RP.replay(e, 7, 20, [(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)], np.zeros(3), [1.0, 0.1], 5, truth_moves=False)
    w, maps = e.state()
    assert len(w) == 20 and np.all(np.isfinite(w)) and np.all(np.isfinite(maps)) and len(fired) == 5
    e = RP.LiteralEngine(120, 1)
    fired, _ = RP.replay(e, 17, 120, [(5.0, 5.0)], np.array([0.0, 0.0, math.pi / 4]), [0.5, 0.0], 60, truth_moves=True)
    err = RP.landmark_error(e, (5.0, 5.0))
    assert err < 6.0, f"landmark estimate should converge: err={err}"
    assert any(fired)
