"""GPU parity of the FastSLAM 2.0 engine (through the C ABI) against
  * oracle/det_spec.c   -- BIT-EXACT: sampled poses, maps, weights, gate decision, resample indices
  * oracle/ref_literal.c -- the reference arithmetic (fastslam2.rs), rtol = atol = 1e-6
and the reference's own unit tests (fastslam2.rs:431-545) re-expressed against the engine."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p
from tests import helpers as H
from tests.test_gpu_fs1_parity import bits_equal, make_state, scene

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-6, atol=1e-6)


@pytest.fixture(scope="module")
def fs2():
    from rust_robotics_amd.slam import fastslam2

    return fastslam2


def obs_for(fs2, pose, lms, seed, step):
    return np.array(fs2.get_observations(pose, [tuple(p) for p in lms], seed=seed, step=step)).reshape(-1, 3)


@pytest.mark.parametrize("n,with_obs", [(1, True), (777, True), (5000, True), (1000, False)])
def test_proposal_sampling_matches_oracles(fs2, det, ref, n, with_obs):
    """the sampling step alone (:339-358) with explicit normals: bit-exact vs the D-spec, 1e-6 vs the
    literal restatement; a third of the particles have the observed landmark uninitialised"""
    L = 5
    lms = scene(L, 3)
    poses, maps = make_state(n, L, lms, 4)
    maps[::3, 2, 2] = maps[::3, 2, 5] = 1000.0  # landmark 2 not initialised there
    f = fs2.FastSlam2(n, L, seed=5)
    f.set_state(poses, maps)
    z = obs_for(fs2, np.array([0.1, 0.0, 0.05]), lms, 9, 0)
    z = np.ascontiguousarray(z[[2, 0, 4]]) if with_obs else z[:0]
    noise = np.ascontiguousarray(np.random.default_rng(6).normal(size=(n, 3)))
    f.propose_with_noise([1.0, 0.1], z, noise)
    gp, _ = f.get_state()
    m = oracle.det_fs2_model()
    px, py, pyaw = (poses[:, k].copy() for k in (1, 2, 3))  # copies: a column of a 1-row array is a view
    planes = oracle.maps_aos_to_planes(maps.copy(), n, L)
    det.det_fs2_predict(n, dp(px), dp(py), dp(pyaw), dp(planes), 1.0, 0.1, dp(z) if len(z) else None, len(z), dp(noise), 0, 0, 0, C.byref(m))
    assert bits_equal(gp[:, 1], px) and bits_equal(gp[:, 2], py) and bits_equal(gp[:, 3], pyaw)
    for p in range(min(n, 60)):
        pose = np.ascontiguousarray(poses[p, 1:4])
        if with_obs:
            mean, cov, out = np.empty(3), np.empty(9), np.empty(3)
            ref.ref_fs2_proposal(dp(pose), 1.0, 0.1, z[0, 0], z[0, 1], dp(np.ascontiguousarray(maps[p, int(z[0, 2])])), 0.5, 0.0305, dp(mean), dp(cov))
            ref.ref_fs2_sample(dp(mean), dp(cov), dp(np.ascontiguousarray(noise[p])), dp(out))
            np.testing.assert_allclose(gp[p, 1:4], out, **TOL)


@pytest.mark.parametrize("chunks,read_every,n_observed", [(1, 1, 12), (0, 4, 12), (3, 5, 7)])
def test_trajectory_bit_exact_vs_det(fs2, det, chunks, read_every, n_observed):
    """whole updates with the engine's Philox streams; read_every > 1 leaves the resample gathers to
    be consumed lazily by the next update's proposal / EKF kernels"""
    n, L, T = 1500, 12, 12
    lms = scene(L, 41)
    prm = fs2.default_params()
    prm.base.nth = n / 1.5
    prm.base.initial_weight = 1.0 / n
    f = fs2.FastSlam2(n, L, params=prm, seed=77, obs_chunks=chunks)
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 1.0 / n)
    planes = oracle.maps_aos_to_planes(np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)), n, L)
    m = oracle.det_fs2_model()
    idx = np.empty(n, np.uint32)
    fired_log = []
    for t in range(T):
        z = obs_for(fs2, H.true_pose(t + 1), lms, seed=77, step=t)
        z = np.ascontiguousarray(z[(t % 3):][:n_observed]) if t != 4 else z[:0]
        f.update([1.0, 0.1], z)
        used = f.counters()[2]
        fired = det.det_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), 1.0, 0.1, dp(z) if len(z) else None, len(z),
                                   C.byref(m), None, n / 1.5, 77, t, t, used, u32p(idx))
        assert f.last_resample_fired() == bool(fired), f"gate differs at step {t}"
        fired_log.append(int(fired))
        if fired:
            assert np.array_equal(f.last_resample_indices(), idx), f"indices differ at step {t}"
        if (t + 1) % read_every and t != T - 1:
            continue
        gp, gm = f.get_state()
        assert bits_equal(gp[:, 0], pw), f"weights step {t}"
        assert bits_equal(gp[:, 1], px) and bits_equal(gp[:, 2], py) and bits_equal(gp[:, 3], pyaw), f"poses step {t}"
        assert bits_equal(gm.reshape(-1), oracle.maps_planes_to_aos(planes, n, L)), f"maps step {t}"
    assert any(fired_log) and not all(fired_log), fired_log


def test_reference_tests_reexpressed(fs2):
    """fastslam2.rs:436-456 (create / update does not panic) and :491-545 (landmark convergence)"""
    ps = fs2.create_particles(20, 3)
    assert len(ps) == 20 and all(len(p.landmarks) == 3 and p.weight == 1.0 / 100 for p in ps)
    lms = [(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)]
    for t in range(5):
        fs2.fastslam2_update(ps, [1.0, 0.1], fs2.get_observations([0.0, 0.0, 0.0], lms, seed=7, step=t), seed=7)
    assert len(ps) == 20
    # :491-545.  The bound (6 m, landmark 7 m away) is loose and still seed dependent: the literal
    # restatement meets it for ~70 % of numpy seeds (the reference pins StdRng seed 17), so the engine
    # is asked for the same thing statistically -- the median over nine seeds
    n, errs = 120, []
    for seed in range(9):
        f = fs2.FastSlam2(n, 1, seed=17 + seed)
        xt = np.array([0.0, 0.0, math.pi / 4])
        for t in range(60):
            xt = np.array([xt[0] + 0.05 * math.cos(xt[2]), xt[1] + 0.05 * math.sin(xt[2]), xt[2]])
            f.update([0.5, 0.0], fs2.get_observations(xt, [(5.0, 5.0)], seed=17 + seed, step=t))
        poses, maps = f.get_state()
        init = maps[:, 0, 2] < 100.0
        assert init.any()
        w = poses[init, 0]
        mx, my = (np.average(maps[init, 0, 0], weights=w), np.average(maps[init, 0, 1], weights=w)) if w.sum() > 0 else (
            maps[init, 0, 0].mean(), maps[init, 0, 1].mean())
        errs.append(math.hypot(mx - 5.0, my - 5.0))
    assert np.median(errs) < 6.0, errs


def test_sharded_equals_unsharded_world1(fs2):
    """the sharded update (peer-to-peer transport, here one shard) of a FastSLAM 2.0 filter"""
    n, L = 3000, 6
    lms = scene(L, 5)
    prm = fs2.default_params()
    prm.base.nth = n / 1.5
    prm.base.initial_weight = 1.0 / n
    a = fs2.FastSlam2(n, L, params=prm, seed=3, obs_chunks=2)
    b = fs2.ShardedFastSlam2(0, 1, n, L, params=prm, seed=3, obs_chunks=2)
    fs2._f1.ShardedFastSlam1.link_local([b])
    for t in range(8):
        z = obs_for(fs2, H.true_pose(t + 1), lms, seed=3, step=t)
        a.update([1.0, 0.1], z)
        b.update([1.0, 0.1], z)
    assert not b.timed_out()
    (pa, ma), (pb, mb) = a.get_state(), b.get_state()
    assert bits_equal(pa, pb) and bits_equal(ma, mb)


def test_duplicate_observations_and_parameter_validation(fs2, det):
    """the same landmark twice in one step (the second update must see the first one's result: the engine
    settles a pending resample first and updates in place) and rr_fs2_create's parameter checks"""
    n, L = 900, 4
    lms = scene(L, 8)
    prm = fs2.default_params()
    prm.base.nth = n / 1.5
    prm.base.initial_weight = 1.0 / n
    f = fs2.FastSlam2(n, L, params=prm, seed=31, obs_chunks=0)
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 1.0 / n)
    planes = oracle.maps_aos_to_planes(np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)), n, L)
    m = oracle.det_fs2_model()
    idx = np.empty(n, np.uint32)
    for t in range(8):
        z = obs_for(fs2, H.true_pose(t + 1), lms, seed=31, step=t)
        if t % 4 == 1:
            z = np.ascontiguousarray(np.vstack([z, z[1:2], z[0:1]]))  # landmarks 1 and 0 observed twice
        elif t % 4 == 3:  # ADJACENT repeats (incl. init-then-EKF of a landmark on its first step when t == 3 is its first sighting)
            z = np.ascontiguousarray(np.vstack([z[0:1], z[0:1], z[1:], z[-1:]]))
        f.update([1.0, 0.1], z)
        fired = det.det_fs2_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), 1.0, 0.1, dp(z), len(z), C.byref(m), None,
                                   n / 1.5, 31, t, t, f.counters()[2], u32p(idx))
        assert f.last_resample_fired() == bool(fired)
    gp, gm = f.get_state()
    assert bits_equal(gp[:, 0], pw) and bits_equal(gp[:, 1], px) and bits_equal(gp[:, 3], pyaw)
    assert bits_equal(gm.reshape(-1), oracle.maps_planes_to_aos(planes, n, L))
    from rust_robotics_amd.core import RoboticsError

    bad = fs2.default_params()
    bad.motion_cov[1] = -1.0
    with pytest.raises(RoboticsError):
        fs2.FastSlam2(10, 2, params=bad)
    bad = fs2.default_params()
    bad.base.first_obs_cov = float("nan")
    with pytest.raises(RoboticsError):
        fs2.FastSlam2(10, 2, params=bad)
    with pytest.raises(RoboticsError):
        f.update([1.0, 0.1], [(1.0, 0.0, L)])  # landmark id out of range


def test_reference_seeded_tests_replayed_on_the_gpu(fs2):
    """The same two seeded tests (fastslam2.rs:443-456 seed 7, :491-545 seed 17 with `lm_err < 6.0`) on the GPU, through the
    explicit-noise seams, next to the literal restatement fed the identical stream: the reference's own assertions hold on both,
    the gate decisions agree step for step, and the two engines end within 1e-6 of each other (weights, landmark means and
    covariances of every particle) -- the only numeric assertions the reference holds for this path, run with its own seeds."""
    from tests import fs2_replay as RP

    cases = [(7, 20, [(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)], np.zeros(3), [1.0, 0.1], 5, False),
             (17, 120, [(5.0, 5.0)], np.array([0.0, 0.0, math.pi / 4]), [0.5, 0.0], 60, True)]
    for seed, n, lms, x0, u, steps, moves in cases:
        g, lit = RP.GpuEngine(fs2, n, len(lms)), RP.LiteralEngine(n, len(lms))
        fg, rg = RP.replay(g, seed, n, lms, x0.copy(), u, steps, moves)
        fl, rl = RP.replay(lit, seed, n, lms, x0.copy(), u, steps, moves)
        assert fg == fl, "the N_eff gate decided differently somewhere along the trajectory"
        assert rg.next_u64() == rl.next_u64(), "the two replays consumed different amounts of the random stream"
        wg, mg = g.state()
        wl, ml = lit.state()
        assert len(wg) == n and np.all(np.isfinite(wg)) and np.all(np.isfinite(mg))
        np.testing.assert_allclose(wg, wl, **TOL)
        np.testing.assert_allclose(mg, ml, **TOL)
        if seed == 17:
            assert any(fg)
            for e in (g, lit):
                err = RP.landmark_error(e, (5.0, 5.0))
                assert err < 6.0, f"landmark estimate should converge: err={err}"
