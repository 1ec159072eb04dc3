"""Two restatements of the reference, written independently of each other -- oracle/ref_literal.c (C, everything expanded into scalars)
and oracle/ref_python.py (pure Python, the Rust code's own structure with nalgebra's evaluation order) -- must agree BIT FOR BIT on
random inputs: both are plain IEEE double arithmetic in the reference's order over the same C library.  A slip of transcription in
either (a swapped index of the column-major Matrix2, a transposed Jacobian, `<` for `<=`, the wrong fallback index) shows up here as
a disagreement; no tolerance is there to absorb it.  (Neither is the reference's binary: parity stays unpinned)"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p
from oracle import ref_python as P


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


@pytest.fixture(scope="module")
def ref():
    r = oracle.ref()
    r.ref_set_threads(1)
    r.ref_gauss_likelihood.restype = C.c_double
    r.ref_gauss_likelihood.argtypes = [C.c_double, C.c_double]
    r.ref_normalize_angle.restype = C.c_double
    r.ref_normalize_angle.argtypes = [C.c_double]
    return r


def test_gauss_likelihood_and_normalize_angle(ref):
    rng = np.random.default_rng(1)
    for x, s in zip(rng.normal(0, 3, 2000), rng.uniform(0.01, 5.0, 2000)):
        assert bits(ref.ref_gauss_likelihood(x, s)) == bits(P.gauss_likelihood(float(x), float(s)))
    assert P.gauss_likelihood(0.0, 0.2) == 1.9947114020071635 and P.gauss_likelihood(0.3, 0.2) == 0.6475879783294588
    for a in list(rng.uniform(-30, 30, 2000)) + [math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 0.0]:
        assert bits(ref.ref_normalize_angle(a)) == bits(P.normalize_angle(float(a)))
    assert P.normalize_angle(math.pi) == math.pi and P.normalize_angle(-math.pi) == -math.pi  # strict comparisons: pi stays (Q3)


@pytest.mark.parametrize("n,L,seed", [(1, 1, 2), (37, 4, 3), (200, 7, 4)])
def test_particle_filter_step_by_step(ref, n, L, seed):
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(-2, 2, n), rng.uniform(-2, 2, n)
    yaw, v = rng.uniform(-4, 4, n), rng.uniform(0, 2, n)
    w = np.full(n, 1.0 / n)
    parts = [[float(x[i]), float(y[i]), float(yaw[i]), float(v[i]), float(w[i])] for i in range(n)]
    sigma, dt, u = 0.3, 0.1, (1.0, 0.1)
    for step in range(4):
        nv, nw = rng.normal(0, 2.0, n), rng.normal(0, 0.7, n)
        if step == 2:
            nv[:] = 0.0  # velocity_noise == 0: no sample, the term is exactly 0.0 (particle_filter.rs:259-276)
        ref.ref_pf_predict(n, dp(x), dp(y), dp(yaw), dp(v), u[0], u[1], dt, dp(nv), dp(nw))
        P.pf_predict(parts, u, dt, [float(t) for t in nv], [float(t) for t in nw])
        obs = np.ascontiguousarray(np.column_stack([rng.uniform(0, 6, L), rng.uniform(-3, 3, L), rng.uniform(-3, 3, L)]))
        ref.ref_pf_update_raw(n, dp(x), dp(y), dp(w), dp(obs), L, sigma)
        P.pf_update_raw(parts, [tuple(float(t) for t in row) for row in obs], sigma)
        got = np.array(parts)
        for k, a in enumerate((x, y, yaw, v, w)):
            assert np.array_equal(bits(got[:, k]), bits(a)), (step, k)
        ref.ref_pf_normalize.restype = C.c_double
        s_c = ref.ref_pf_normalize(n, dp(w))
        s_p = P.pf_normalize(parts)
        assert bits(s_c) == bits(s_p)
        assert np.array_equal(bits(np.array(parts)[:, 4]), bits(w))
        ref.ref_pf_neff.restype = C.c_double
        assert bits(ref.ref_pf_neff(n, dp(w))) == bits(P.pf_n_eff(parts))
        est = np.empty(4)
        ref.ref_pf_estimate(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), dp(est))
        est_p = P.pf_estimate(parts)
        assert np.array_equal(bits(est), bits(est_p))
        cov = np.empty(16)
        ref.ref_pf_covariance(n, dp(x), dp(y), dp(yaw), dp(v), dp(w), dp(est), dp(cov))
        assert np.array_equal(bits(cov), bits(np.array(P.pf_covariance(parts, est_p)).reshape(-1)))
        # the two resamplers on these weights (indices only; the copy itself is a memcpy)
        r = rng.uniform(0, 1, n)
        r[0] = 0.0
        if n > 2:
            r[1] = float(np.cumsum(w)[n // 2])  # a draw ON a boundary: `<=` takes this index, `<` would take the next
            r[2] = 1.0 - 1e-17  # rounds to 1.0: beyond a cumulative sum that stops short of 1 -> fallback index
        idx = np.empty(n, np.uint32)
        ref.ref_pf_resample_indices(n, dp(w), dp(r), u32p(idx))
        assert list(idx) == P.pf_resample_indices([float(t) for t in w], [float(t) for t in r])
        ref.ref_mcl_resample_indices(n, dp(w), dp(r), u32p(idx))
        assert list(idx) == P.mcl_resample_indices([float(t) for t in w], [float(t) for t in r])
    # all weights zero: the uniform fallback (particle_filter.rs:433-438)
    for p in parts:
        p[4] = 0.0
    w[:] = 0.0
    assert ref.ref_pf_normalize(n, dp(w)) == 0.0 and P.pf_normalize(parts) == 0.0
    assert np.all(w == 1.0 / n) and all(p[4] == 1.0 / n for p in parts)


def test_resample_fallbacks_differ_between_the_two_localizers(ref):
    """particle_filter.rs:459-465 falls back to index 0, monte_carlo_localization.rs:387-392 forces c[last] = 1 and falls back to last"""
    w = np.array([0.25, 0.25, 0.25, 0.2])  # sums to 0.95: a draw of 0.97 matches nothing in the PF's scan
    r = np.array([0.97, 0.5, 0.0, 0.9499999999999999])
    idx = np.empty(4, np.uint32)
    ref.ref_pf_resample_indices(4, dp(w), dp(r), u32p(idx))
    assert list(idx) == P.pf_resample_indices(list(w), list(r)) == [0, 1, 0, 3]
    ref.ref_mcl_resample_indices(4, dp(w), dp(r), u32p(idx))
    assert list(idx) == P.mcl_resample_indices(list(w), list(r)) == [3, 1, 0, 3]


def _py_particles(px, py, pyaw, pw, lm, n, L):
    out = []
    for p in range(n):
        q = P.Particle(L, float(pw[p]))
        q.x, q.y, q.yaw = float(px[p]), float(py[p]), float(pyaw[p])
        for l in range(L):
            e = lm[(p * L + l) * 6:(p * L + l) * 6 + 6]
            q.landmarks[l] = P.Landmark(float(e[0]), float(e[1]), [[float(e[2]), float(e[4])], [float(e[3]), float(e[5])]])  # column-major c00 c10 c01 c11
        out.append(q)
    return out


def _flat(parts, L):
    n = len(parts)
    px, py, pyaw, pw = (np.array([getattr(q, k) for q in parts]) for k in ("x", "y", "yaw", "weight"))
    lm = np.empty(n * L * 6)
    for p, q in enumerate(parts):
        for l, m in enumerate(q.landmarks):
            lm[(p * L + l) * 6:(p * L + l) * 6 + 6] = (m.x, m.y, m.cov[0][0], m.cov[1][0], m.cov[0][1], m.cov[1][1])
    return px, py, pyaw, pw, lm


def test_update_landmark_both_branches_and_the_singular_fallback(ref):
    rng = np.random.default_rng(7)
    mr = oracle.ref_fs1_model()
    r = [[mr.r00, 0.0], [0.0, mr.r11]]
    for trial in range(400):
        cov = rng.uniform(-0.3, 0.3, (2, 2)) + np.diag(rng.uniform(0.05, 3.0, 2))  # not symmetric on purpose: (I - K H) P is not either (Q12)
        if trial % 5 == 0:
            cov[0, 0] = 1000.0  # first-observation branch (fastslam1.rs:143-149)
        e = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), cov[0, 0], cov[1, 0], cov[0, 1], cov[1, 1]])
        pose = rng.uniform(-3, 3, 3)
        z = (float(rng.uniform(0.5, 15.0)), float(rng.uniform(-3.5, 3.5)))
        wv = C.c_double(float(rng.uniform(0.001, 1.0)))
        q = P.Particle(1, wv.value)
        q.x, q.y, q.yaw = (float(t) for t in pose)
        q.landmarks[0] = P.Landmark(float(e[0]), float(e[1]), [[float(cov[0, 0]), float(cov[0, 1])], [float(cov[1, 0]), float(cov[1, 1])]])
        ref.ref_fs1_update_landmark(pose[0], pose[1], pose[2], C.byref(wv), z[0], z[1], dp(e), C.byref(mr))
        P.update_landmark(q, list(z), 0, r)
        m = q.landmarks[0]
        assert np.array_equal(bits(e), bits([m.x, m.y, m.cov[0][0], m.cov[1][0], m.cov[0][1], m.cov[1][1]])), trial
        assert bits(wv.value) == bits(q.weight), trial
    # singular S: zero covariance and zero R -> try_inverse is None -> identity, det S = 0 -> no likelihood (fastslam1.rs:164,177-182)
    mz = oracle.ref_fs1_model()
    mz.r00 = mz.r11 = 0.0
    e = np.array([4.0, 1.0, 0.0, 0.0, 0.0, 0.0])
    wv = C.c_double(0.37)
    q = P.Particle(1, 0.37)
    q.landmarks[0] = P.Landmark(4.0, 1.0, [[0.0, 0.0], [0.0, 0.0]])
    ref.ref_fs1_update_landmark(0.0, 0.0, 0.0, C.byref(wv), 4.5, 0.2, dp(e), C.byref(mz))
    P.update_landmark(q, [4.5, 0.2], 0, [[0.0, 0.0], [0.0, 0.0]])
    assert wv.value == q.weight == 0.37 and np.array_equal(bits(e[:2]), bits([q.landmarks[0].x, q.landmarks[0].y]))


@pytest.mark.parametrize("n,L,steps,seed", [(30, 5, 8, 11), (80, 3, 6, 12)])
def test_fastslam_update_trajectories(ref, n, L, steps, seed):
    """fastslam1.rs:237-266 over several steps with explicit noise and resampling offsets: first-observation branch in the first step
    (with the covariance it leaves -- the reference's own leaves 1000 I for ever, so a second run sets 0.5 I as the engine's option does),
    EKF branch afterwards, data-dependent resampling, best particle."""
    for init_cov in (float("nan"), 0.5):
        rng = np.random.default_rng(seed)
        mr = oracle.ref_fs1_model()
        mr.init_cov = init_cov
        px, py, pyaw, pw, lm = np.empty(n), np.empty(n), np.empty(n), np.empty(n), np.empty(n * L * 6)
        ref.ref_fs1_create(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm))
        parts = [P.Particle(L) for _ in range(n)]
        assert np.array_equal(bits(np.concatenate(_flat(parts, L))), bits(np.concatenate([px, py, pyaw, pw, lm])))
        lms = rng.uniform(-10, 10, (L, 2))
        # (under the reference's own settings no weight ever changes -- the EKF branch is dead, SURVEY Q11 -- so N_eff stays n: a threshold
        # above n makes that run resample every step, uniform weights through the systematic walk)
        nth = n + 1.0 if math.isnan(init_cov) else 0.999999 * n  # (any spread of the weights at all fires the gate)
        idx = np.empty(n, np.uint32)
        fired_any = False
        for t in range(steps):
            z0, z1 = rng.normal(0, 1, n), rng.normal(0, 1, n)
            truth = np.array([0.1 * t, 0.02 * t, 0.01 * t])
            ids = rng.permutation(L)[: max(1, L - t % 2)]
            z = np.ascontiguousarray([[math.hypot(lms[i, 0] - truth[0], lms[i, 1] - truth[1]) + rng.normal(0, 0.3),
                                       P.normalize_angle(math.atan2(lms[i, 1] - truth[1], lms[i, 0] - truth[0]) - truth[2]) + rng.normal(0, 0.1), float(i)] for i in ids])
            r0 = float(rng.uniform(0, 1.0 / n))
            fired = ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(z0), dp(z1), dp(z), len(z), C.byref(mr), nth, r0, u32p(idx))
            if math.isnan(init_cov):
                parts, fired_p, idx_p = P.fastslam_update(parts, [1.0, 0.1], [tuple(row) for row in z.tolist()], list(zip(z0.tolist(), z1.tolist())), r0, nth=nth)
            else:  # the engine's `first_obs_cov` option: the first observation also resets the covariance (not in the reference)
                for q, n0, n1 in zip(parts, z0.tolist(), z1.tolist()):
                    P.predict_particle(q, [1.0, 0.1], n0, n1)
                for d, a, i in z.tolist():
                    for q in parts:
                        first = q.landmarks[int(i)].cov[0][0] > 100.0
                        P.update_landmark(q, [d, a], int(i), P.R_SIM)
                        if first:
                            q.landmarks[int(i)].cov = [[init_cov, 0.0], [0.0, init_cov]]
                P.normalize_weights(parts)
                fired_p, idx_p = False, None
                if P.compute_neff(parts) < nth:
                    parts, idx_p = P.resample(parts, r0)
                    fired_p = True
            assert bool(fired) == fired_p, t
            fired_any |= fired_p
            if fired_p:
                assert list(idx) == idx_p
            assert np.array_equal(bits(np.concatenate(_flat(parts, L))), bits(np.concatenate([px, py, pyaw, pw, lm]))), (init_cov, t)
            ref.ref_fs1_best_particle.restype = C.c_size_t
            assert ref.ref_fs1_best_particle(n, dp(pw)) == P.get_best_particle_index(parts)
        assert fired_any


def test_best_particle_ties_go_to_the_last(ref):
    parts = [P.Particle(0, w) for w in (0.1, 0.9, 0.3, 0.9, 0.2)]
    assert P.get_best_particle_index(parts) == 3
    pw = np.array([0.1, 0.9, 0.3, 0.9, 0.2])
    ref.ref_fs1_best_particle.restype = C.c_size_t
    assert ref.ref_fs1_best_particle(5, dp(pw)) == 3


@pytest.mark.parametrize("n,lo,hi,spread,seed", [(300, 100, 5000, 0.3, 21), (300, 100, 5000, 6.0, 22), (50, 10, 60, 3.0, 23), (400, 1, 400, 20.0, 24)])
def test_kld_adaptive_resample(ref, n, lo, hi, spread, seed):
    """monte_carlo_localization.rs:322-385: the same kept prefix of the draw sequence, the same count -- a tight cloud (few bins: the
    count stays at min_particles), a wide one (the bound grows with the bins and the loop runs into max_particles), and the formula
    itself over every bin count that can occur."""
    rng = np.random.default_rng(seed)
    x, y, yaw = rng.normal(0, spread, n), rng.normal(0, spread, n), rng.normal(0, 0.5 * spread, n)
    w = rng.uniform(0, 1, n) ** 3
    w /= w.sum()
    r = rng.uniform(0, 1, hi)
    r[0] = 1.0 - 1e-17  # == 1.0: only the forced last cumulative weight answers it
    idx = np.empty(hi, np.uint32)
    ref.ref_mcl_resample_adaptive.restype = C.c_size_t
    ref.ref_mcl_resample_adaptive.argtypes = [C.c_size_t] + [C.POINTER(C.c_double)] * 5 + [C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_uint32)]
    cnt = ref.ref_mcl_resample_adaptive(n, dp(x), dp(y), dp(yaw), dp(w), dp(r), lo, hi, 0.05, 2.326, u32p(idx))
    got = P.resample_adaptive_indices(x.tolist(), y.tolist(), yaw.tolist(), w.tolist(), r.tolist(), lo, hi, 0.05, 2.326)
    assert cnt == len(got) and list(idx[:cnt]) == got
    assert lo <= cnt <= hi
    ref.ref_kld_required.restype = C.c_size_t
    ref.ref_kld_required.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double]
    for k in list(range(0, 200)) + [1000, 10**6]:
        assert ref.ref_kld_required(k, lo, hi, 0.05, 2.326) == P.kld_required_particles(k, lo, hi, 0.05, 2.326), k
    assert P.quantize_particle(-0.1, 0.5, -1e300) == (-1, 1, -2147483648) and P.quantize_particle(float("nan"), 0.49, 0.0) == (0, 0, 0)


def test_reference_seeded_fastslam2_tests_through_both_restatements():
    """fastslam2.rs:443-456 (StdRng seed 7) and :491-545 (seed 17, `lm_err < 6.0`) replayed draw for draw (tests/fs2_replay.py) through
    ref_literal.c AND through ref_python.py: the same gate decisions step for step, the same amount of stream consumed, identical bits in
    every weight, pose and map entry at the end -- and the reference's own assertion holds on both."""
    from tests import fs2_replay as RP

    for seed, n, lms, x0, u, steps, moves in ((7, 20, [(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)], np.zeros(3), [1.0, 0.1], 5, False),
                                             (17, 120, [(5.0, 5.0)], np.array([0.0, 0.0, math.pi / 4]), [0.5, 0.0], 60, True)):
        a, b = RP.LiteralEngine(n, len(lms)), RP.PythonEngine(n, len(lms))
        fa, rng_a = RP.replay(a, seed, n, lms, x0.copy(), u, steps, truth_moves=moves)
        fb, rng_b = RP.replay(b, seed, n, lms, x0.copy(), u, steps, truth_moves=moves)
        assert fa == fb, "gate decisions differ"
        assert rng_a.next_u64() == rng_b.next_u64(), "the two replays consumed different amounts of the stream"
        (wa, ma), (wb, mb) = a.state(), b.state()
        assert np.array_equal(bits(wa), bits(wb)) and np.array_equal(bits(ma), bits(mb))
        assert np.array_equal(bits(np.column_stack([a.px, a.py, a.pyaw])), bits(b.poses()))
        if seed == 17:
            ea, eb = RP.landmark_error(a, (5.0, 5.0)), RP.landmark_error(b, (5.0, 5.0))
            assert ea == eb and ea < 6.0


def test_fastslam2_proposal_sample_and_update_on_random_inputs(ref):
    """compute_proposal (fastslam2.rs:173-216), sample_pose (:219-239) and update_landmark_and_weight (:242-280) one call at a time:
    initialised and uninitialised landmarks, asymmetric covariances, a proposal covariance that is not positive definite (the Cholesky
    gives up: the diagonal fallback) and one with a zero pivot."""
    rng = np.random.default_rng(31)
    r = P.R_SIM
    for trial in range(300):
        pose = rng.uniform(-3, 3, 3)
        cov2 = rng.uniform(-0.2, 0.2, (2, 2)) + np.diag(rng.uniform(0.05, 3.0, 2))
        if trial % 6 == 0:
            cov2[0, 0] = 1000.0  # uninitialised (Landmark::is_initialized, :49-51)
        e = np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), cov2[0, 0], cov2[1, 0], cov2[0, 1], cov2[1, 1]])
        z = (float(rng.uniform(0.5, 15.0)), float(rng.uniform(-3.5, 3.5)))
        u = (float(rng.uniform(-1, 2)), float(rng.uniform(-1, 1)))
        q = P.Particle(1, 0.01)
        q.x, q.y, q.yaw = (float(t) for t in pose)
        q.landmarks[0] = P.Landmark(float(e[0]), float(e[1]), [[float(cov2[0, 0]), float(cov2[0, 1])], [float(cov2[1, 0]), float(cov2[1, 1])]])
        mean, cov = np.empty(3), np.empty(9)
        ref.ref_fs2_proposal(dp(np.ascontiguousarray(pose)), u[0], u[1], z[0], z[1], dp(e), r[0][0], r[1][1], dp(mean), dp(cov))
        m_p, c_p = P.compute_proposal(q, list(u), list(z), 0, r)
        assert np.array_equal(bits(mean), bits(m_p)) and np.array_equal(bits(cov), bits(np.array(c_p).reshape(-1))), trial
        nz = rng.normal(0, 1, 3)
        c_use = cov.copy()
        if trial % 7 == 0:
            c_use[0] = -abs(c_use[0])  # not positive definite: the fallback of :226-233
        if trial % 11 == 0:
            c_use[:] = 0.0
            c_use[4] = c_use[8] = 0.3  # a zero first pivot
        out = np.empty(3)
        ref.ref_fs2_sample(dp(mean), dp(c_use), dp(nz), dp(out))
        sp = P.sample_pose(list(mean), c_use.reshape(3, 3).tolist(), nz.tolist())
        sp[2] = P.normalize_angle(sp[2])  # (ref_fs2_sample includes set_pose's wrap, fastslam2.rs:78-82)
        assert np.array_equal(bits(out), bits(sp)), trial
        e2 = e.copy()
        w_c = ref.ref_fs2_update_landmark(pose[0], pose[1], pose[2], z[0], z[1], dp(e2), r[0][0], r[1][1])
        w_p = P.update_landmark_and_weight(q, list(z), 0, r)
        m = q.landmarks[0]
        assert bits(w_c) == bits(w_p) and np.array_equal(bits(e2), bits([m.x, m.y, m.cov[0][0], m.cov[1][0], m.cov[0][1], m.cov[1][1]])), trial
