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
