"""GPU parity of the FastSLAM 1.0 engine (through the C ABI) against
  * oracle/det_spec.c   -- BIT-EXACT: poses, maps, weights, gate decision, resample indices
  * oracle/ref_literal.c -- the reference arithmetic (fastslam1.rs), rtol = atol = 1e-6
and the reference's own unit tests (fastslam1.rs:308-401) re-expressed against the engine.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p, u64p
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-6, atol=1e-6)


@pytest.fixture(scope="module")
def fs():
    from rust_robotics_amd.slam import fastslam1

    assert fastslam1._ffi.lib().rr_device_count() >= 1
    return fastslam1


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, dtype=np.float64).view(np.uint64), np.ascontiguousarray(b, dtype=np.float64).view(np.uint64))


def scene(L, seed, half=13.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, size=(L, 2))


def make_state(n, L, lms, seed, cov=0.5):
    """maps pre-initialised so that every pair takes the EKF branch"""
    rng = np.random.default_rng(seed)
    poses = np.column_stack([np.full(n, 1.0 / n), rng.normal(0, 0.2, n), rng.normal(0, 0.2, n), rng.normal(0, 0.05, n)])
    maps = np.zeros((n, L, 6))
    maps[:, :, 0] = lms[None, :, 0] + rng.normal(0, 0.5, (n, L))
    maps[:, :, 1] = lms[None, :, 1] + rng.normal(0, 0.5, (n, L))
    maps[:, :, 2] = cov
    maps[:, :, 5] = cov
    maps[:, :, 3] = rng.normal(0, 0.01, (n, L))  # slightly asymmetric on purpose (Q12)
    maps[:, :, 4] = rng.normal(0, 0.01, (n, L))
    return poses, maps


def observations_for(fs, pose, lms, seed, step):
    return np.array(fs.get_observations(pose, [tuple(p) for p in lms], seed=seed, step=step)).reshape(-1, 3)


def test_simulator_matches_det_and_reference_gate(fs, det, ref):
    lms = np.array([[5.0, 0.0], [100.0, 100.0], [0.0, 19.9], [3.0, -4.0]])
    xt = np.array([0.0, 0.0, 0.3])
    z = observations_for(fs, xt, lms, seed=9, step=4)
    out = np.empty((4, 3))
    cnt = det.det_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), 4, 20.0, 0.5, 0.0305, 9, 4, dp(out))
    assert cnt == len(z) == 3 and bits_equal(z, out[:cnt])
    # fastslam1.rs:325-341: only the landmark inside MAX_RANGE is observed
    z2 = fs.get_observations([0.0, 0.0, 0.0], [(5.0, 0.0), (100.0, 100.0)], seed=1)
    assert len(z2) == 1 and z2[0][2] == 0 and abs(z2[0][0] - 5.0) < 5.0
    # literal reference with the same unit normals
    zn = np.empty(8)
    z0, z1 = np.empty(4), np.empty(4)
    det.det_normal2_v(9, 5, 4, 0, 4, dp(z0), dp(z1))
    zn[0::2], zn[1::2] = z0, z1
    m = oracle.ref_fs1_model()
    outr = np.empty((4, 3))
    cr = ref.ref_fs1_get_observations(dp(xt), dp(np.ascontiguousarray(lms)), 4, 20.0, dp(zn), C.byref(m), dp(outr))
    assert cr == cnt
    np.testing.assert_allclose(z, outr[:cr], rtol=1e-13, atol=1e-13)


def test_initial_state_matches_reference(fs):
    """fastslam1.rs:382-400"""
    f = fs.FastSlam1(10, 4)
    poses, maps = f.get_state()
    assert np.all(poses[:, 1:] == 0.0)
    assert np.all(np.abs(poses[:, 0] - 1.0 / 100) < np.finfo(float).eps)
    assert np.all(maps[:, :, 0] == 0) and np.all(maps[:, :, 1] == 0)
    assert np.all(maps[:, :, 2] == 1000.0) and np.all(maps[:, :, 5] == 1000.0)
    assert np.all(maps[:, :, 3] == 0) and np.all(maps[:, :, 4] == 0)
    ps = fs.create_particles(50, 5)
    assert len(ps) == 50 and all(len(p.landmarks) == 5 for p in ps)


@pytest.mark.parametrize("n", [1, 64, 1000, 4097])
def test_predict_with_noise(fs, det, ref, n):
    rng = np.random.default_rng(3)
    poses = np.column_stack([np.full(n, 1.0 / n), rng.normal(0, 1, n), rng.normal(0, 1, n), rng.uniform(-3.1, 3.1, n)])
    z0, z1 = rng.normal(size=n), rng.normal(size=n)
    f = fs.FastSlam1(n, 2)
    f.set_state(poses, None)
    f.predict_with_noise([1.0, 0.1], z0, z1)
    got = f.poses()
    px, py, pyaw = (np.array(poses[:, k], dtype=np.float64, order='C', copy=True) for k in (1, 2, 3))
    md = oracle.det_fs1_model()
    det.det_fs1_predict(n, dp(px), dp(py), dp(pyaw), 1.0, 0.1, dp(z0), dp(z1), 0, 0, 0, C.byref(md))
    assert bits_equal(got[:, 1], px) and bits_equal(got[:, 2], py) and bits_equal(got[:, 3], pyaw)
    rx, ry, ryaw = (np.array(poses[:, k], dtype=np.float64, order='C', copy=True) for k in (1, 2, 3))
    mr = oracle.ref_fs1_model()
    ref.ref_fs1_predict(n, dp(rx), dp(ry), dp(ryaw), 1.0, 0.1, dp(z0), dp(z1), C.byref(mr))
    np.testing.assert_allclose(got[:, 1:], np.column_stack([rx, ry, ryaw]), rtol=1e-13, atol=1e-14)
    assert np.all(np.abs(got[:, 3]) <= math.pi)


@pytest.mark.parametrize("n,L,chunks", [(300, 6, 1), (1000, 50, 1), (1000, 50, 4), (2049, 33, 7), (500, 200, 0), (200, 3000, 0), (100, 5000, 64)])
def test_observe_ekf_matches_oracles(fs, det, ref, n, L, chunks):
    lms = scene(L, 11)
    poses, maps = make_state(n, L, lms, 12)
    truth = np.array([0.0, 0.0, 0.0])
    z = observations_for(fs, truth, lms, seed=13, step=0)
    assert len(z) == L
    f = fs.FastSlam1(n, L, obs_chunks=chunks)
    f.set_state(poses, maps)
    f.observe(z)
    used = f.counters()[2]
    if chunks:
        assert used == math.ceil(L / math.ceil(L / chunks))
    gp, gm = f.get_state()
    # D-spec (landmark-major planes), same chunking
    px, py, pyaw, pw = (np.array(poses[:, k], dtype=np.float64, order='C', copy=True) for k in (1, 2, 3, 0))
    planes = oracle.maps_aos_to_planes(maps, n, L)
    md = oracle.det_fs1_model()
    det.det_fs1_observe(n, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), dp(z), len(z), C.byref(md), used)
    assert bits_equal(gp[:, 0], pw), "accumulated weights"
    assert bits_equal(gm.reshape(-1), oracle.maps_planes_to_aos(planes, n, L)), "maps"
    # literal reference: observation outer, particle inner (fastslam1.rs:250-256)
    rw = np.array(poses[:, 0], copy=True)
    rm = maps.copy().reshape(-1)
    mr = oracle.ref_fs1_model()
    for k in range(len(z)):
        for p in range(n):
            wv = C.c_double(rw[p])
            e = rm[(p * L + int(z[k, 2])) * 6:(p * L + int(z[k, 2])) * 6 + 6]
            ref.ref_fs1_update_landmark(poses[p, 1], poses[p, 2], poses[p, 3], C.byref(wv), z[k, 0], z[k, 1], dp(e), C.byref(mr))
            rw[p] = wv.value
        if n * L > 60000 and k >= 20:
            break
    if n * L <= 60000:
        np.testing.assert_allclose(gm.reshape(-1), rm, **TOL)
        big = rw > 1e-290  # (a handful of factors of at most 1.29: tests/test_gpu_baseline_literal.py has the bound)
        np.testing.assert_allclose(gp[big, 0], rw[big], rtol=1e-6)


@pytest.mark.parametrize("first_obs_cov", [None, 10.0])
def test_adjacent_duplicate_ids_update_sequentially(fs, det, ref, first_obs_cov):
    """fastslam1.rs:250-256 runs a step's observations one after the other, so a landmark that is
    observed twice in a row gets init-then-EKF (or EKF-then-EKF): the second update must see the
    first one's result.  Adjacent repeats are the case a software-pipelined plane load gets wrong."""
    n, L = 700, 3
    lms = scene(L, 61)
    if first_obs_cov is None:  # maps initialised: EKF branch twice
        poses, maps = make_state(n, L, lms, 62)
        f = fs.FastSlam1(n, L, obs_chunks=0)
    else:  # fresh maps (cov = 1000 I): initialise on the first observation, EKF on the repeat
        prm = fs.default_params()
        prm.first_obs_cov = first_obs_cov
        poses, maps = make_state(n, L, lms, 62)
        maps[:] = np.array([0, 0, 1000.0, 0, 0, 1000.0])
        f = fs.FastSlam1(n, L, params=prm, obs_chunks=0)
    z1 = observations_for(fs, np.array([0.0, 0.0, 0.0]), lms, seed=63, step=0)
    z2 = observations_for(fs, np.array([0.0, 0.0, 0.0]), lms, seed=63, step=1)
    # ids 1, 1, 0, 0, 2, 1: two adjacent repeats and a distant one
    z = np.ascontiguousarray(np.vstack([z1[1], z2[1], z1[0], z2[0], z1[2], z1[1]]))
    f.set_state(poses, maps)
    f.observe(z)
    assert f.counters()[2] == 1  # repeated ids force a single chunk
    gp, gm = f.get_state()
    px, py, pyaw, pw = (np.array(poses[:, k], dtype=np.float64, order='C', copy=True) for k in (1, 2, 3, 0))
    planes = oracle.maps_aos_to_planes(maps, n, L)
    md = oracle.det_fs1_model()
    if first_obs_cov is not None:
        md.init_cov = first_obs_cov
    det.det_fs1_observe(n, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), dp(z), len(z), C.byref(md), 1)
    assert bits_equal(gp[:, 0], pw), "accumulated weights"
    assert bits_equal(gm.reshape(-1), oracle.maps_planes_to_aos(planes, n, L)), "maps"
    if first_obs_cov is not None:
        assert np.ptp(gp[:, 0]) > 0.0, "the repeat of a freshly initialised landmark must take the EKF branch and reweight"
    # literal reference: observation outer, particle inner
    rw = np.array(poses[:, 0], copy=True)
    rm = maps.copy().reshape(-1)
    mr = oracle.ref_fs1_model()
    if first_obs_cov is not None:
        mr.init_cov = first_obs_cov
    for k in range(len(z)):
        for p in range(n):
            wv = C.c_double(rw[p])
            e = rm[(p * L + int(z[k, 2])) * 6:(p * L + int(z[k, 2])) * 6 + 6]
            ref.ref_fs1_update_landmark(poses[p, 1], poses[p, 2], poses[p, 3], C.byref(wv), z[k, 0], z[k, 1], dp(e), C.byref(mr))
            rw[p] = wv.value
    np.testing.assert_allclose(gm.reshape(-1), rm, **TOL)
    big = rw > 1e-290  # (a handful of factors of at most 1.29: tests/test_gpu_baseline_literal.py has the bound)
    np.testing.assert_allclose(gp[big, 0], rw[big], rtol=1e-6)


def test_first_observation_branch_reference_quirk(fs, det):
    """Q11: with the reference's parameters the first observation sets x,y and leaves cov at 1000,
    so the weights never change through fastslam_update alone; first_obs_cov switches that."""
    n, L = 256, 5
    lms = scene(L, 21)
    f = fs.FastSlam1(n, L, seed=22)
    for t in range(3):
        f.update([1.0, 0.1], observations_for(fs, H.true_pose(t + 1), lms, seed=22, step=t))
    poses, maps = f.get_state()
    assert np.all(maps[:, :, 2] == 1000.0) and np.all(maps[:, :, 5] == 1000.0)
    assert np.any(maps[:, :, 0] != 0.0)
    assert np.allclose(poses[:, 0], 1.0 / n)  # normalised from 1/100 each, never reweighted
    prm = fs.default_params()
    prm.first_obs_cov = 10.0
    g = fs.FastSlam1(n, L, params=prm, seed=22)
    for t in range(3):
        g.update([1.0, 0.1], observations_for(fs, H.true_pose(t + 1), lms, seed=22, step=t))
    p2, m2 = g.get_state()
    assert np.all(m2[:, :, 2] < 100.0) and np.ptp(p2[:, 0]) > 0.0


def test_systematic_resample_and_gather(fs, det, ref):
    n, L = 5000, 7
    lms = scene(L, 31)
    poses, maps = make_state(n, L, lms, 32)
    rng = np.random.default_rng(33)
    poses[:, 0] = rng.random(n) ** 6 * np.exp(-rng.random(n) * 15)
    rho = float(np.floor(rng.random() * 2**53) / 2**53)
    f = fs.FastSlam1(n, L)
    f.set_state(poses, maps)
    f.resample_systematic(rho)
    idx = f.last_resample_indices()
    w = np.array(poses[:, 0], copy=True)
    fx = H.det_fixed(det, w)
    cdf = H.det_cdf(det, w, fx)
    e = np.empty(n, np.uint32)
    det.det_indices_systematic(n, u64p(cdf), fx["total"], n, 0, n, rho, u32p(e))
    assert np.array_equal(idx, e)
    er = np.empty(n, np.uint32)
    ref.ref_fs1_resample_indices(n, dp(w.copy()), rho / n, u32p(er))
    assert np.array_equal(idx, er)
    gp, gm = f.get_state()
    assert bits_equal(gp[:, 1:], poses[idx, 1:]) and bits_equal(gm, maps[idx])
    assert np.all(gp[:, 0] == 1.0 / n)
    assert f.last_resample_fired()


def test_all_zero_weights_walk_to_last_particle(fs):
    """fastslam1.rs:196-203,224-226: sum w == 0 => no normalisation, N_eff = 0 < NTH, and the walk
    ends on the last particle for every slot"""
    n, L = 300, 3
    poses = np.column_stack([np.zeros(n), np.arange(n, dtype=float), np.zeros(n), np.zeros(n)])
    f = fs.FastSlam1(n, L)
    f.set_state(poses, None)
    assert f.n_eff() == 0.0
    f.normalize_resample()
    assert f.last_resample_fired()
    assert np.all(f.last_resample_indices() == n - 1)
    assert np.all(f.poses()[:, 1] == n - 1)


def test_best_particle_ties_go_last(fs):
    """fastslam1.rs:344-360 + Q14"""
    f = fs.FastSlam1(5, 3)
    poses = np.zeros((5, 4))
    poses[:, 0] = [0.1, 0.5, 0.9, 0.3, 0.2]
    poses[:, 1] = np.arange(5)
    f.set_state(poses, None)
    pose, w, i = f.best_particle()
    assert i == 2 and abs(w - 0.9) < np.finfo(float).eps and pose[0] == 2.0
    poses[:, 0] = [0.9, 0.5, 0.9, 0.3, 0.9]
    f.set_state(poses, None)
    assert f.best_particle()[2] == 4
    n = 100_000
    g = fs.FastSlam1(n, 1)
    p = np.zeros((n, 4))
    p[:, 0] = np.random.default_rng(1).random(n)
    p[[17, 99_998], 0] = 2.0
    g.set_state(p, None)
    assert g.best_particle()[2] == 99_998


@pytest.mark.parametrize("chunks,read_every,n_observed", [(1, 1, 24), (0, 1, 24), (0, 4, 24), (3, 5, 17), (0, 100, 9)])
def test_trajectory_bit_exact_vs_det(fs, det, chunks, read_every, n_observed):
    """read_every > 1 leaves several updates between accessor calls, so the resample gathers are
    consumed lazily by the next update's own loads; n_observed < L exercises the separate gather of
    the unobserved landmarks' planes."""
    trajectory_vs_det(fs, det, chunks, read_every, n_observed)


def trajectory_vs_det(fs, det, chunks, read_every, n_observed, n=2000):
    L, T = 24, 12
    lms = scene(L, 41)
    prm = fs.default_params()
    prm.first_obs_cov = 2.0   # reach the EKF branch through updates alone
    prm.nth = n / 1.5         # the evident intent of fastslam1.rs:18
    f = fs.FastSlam1(n, L, params=prm, seed=77, obs_chunks=chunks)
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    planes = oracle.maps_aos_to_planes(np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)), n, L)
    md = oracle.det_fs1_model()
    md.init_cov = 2.0
    idx = np.empty(n, np.uint32)
    fired_any = False
    for t in range(T):
        z = observations_for(fs, H.true_pose(t + 1), lms, seed=77, step=t)
        z = np.ascontiguousarray(z[(t % 3):][:n_observed])  # a varying subset of the landmarks
        f.update([1.0, 0.1], z)
        used = f.counters()[2]
        fired = det.det_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), 1.0, 0.1, dp(z), len(z), C.byref(md),
                                   n / 1.5, 77, t, t, used, u32p(idx))
        assert f.last_resample_fired() == bool(fired), f"gate differs at step {t}"
        fired_any |= bool(fired)
        if fired:
            assert np.array_equal(f.last_resample_indices(), idx), f"indices differ at step {t}"
        if (t + 1) % read_every and t != T - 1:
            continue
        gp, gm = f.get_state()
        assert bits_equal(gp[:, 0], pw), f"weights step {t}"
        assert bits_equal(gp[:, 1], px) and bits_equal(gp[:, 2], py) and bits_equal(gp[:, 3], pyaw), f"poses step {t}"
        assert bits_equal(gm.reshape(-1), oracle.maps_planes_to_aos(planes, n, L)), f"maps step {t}"
    assert fired_any
    pose, w, i = f.best_particle()
    assert i == det.det_fs1_best_particle(n, dp(pw))
    assert np.hypot(*(pose[:2] - H.true_pose(T)[:2])) < 2.0
    return f


def degrade_in_process():
    """(fresh interpreter, RR_FS1_FACTOR_WAIT_US=0) the closing chunk of k_fs1_observe never waits: a factor that is not there at
    its first look makes it leave the weight to k_fs1_combine"""
    from rust_robotics_amd.slam import fastslam1

    gave_up = 0
    for chunks, read_every, n_observed, n in ((3, 5, 17, 2000), (6, 1, 24, 40_000), (0, 4, 24, 2000)):
        f = trajectory_vs_det(fastslam1, oracle.det(), chunks, read_every, n_observed, n=n)
        g, waits = f.observe_stats()
        assert g == 0 or not waits, "a handle that has given up once forms its weights in the follow-up kernel from then on"
        gave_up += g
    assert gave_up >= 1, "no closing workgroup ever found a factor missing: the degrade path did not run"
    print("OBSERVE_DEGRADE_OK", gave_up)


def test_observe_degrades_when_a_chunk_factor_does_not_arrive(fs):
    """k_fs1_observe's closing chunk waits (bounded) for the other chunks' weight factors, which rests on workgroups being
    dispatched in ascending order (fs1_kernels.inc).  With the wait set to zero every factor that is late is a miss: the closing
    workgroup must leave the weight to k_fs1_combine, the results -- poses, maps, weights, gate decisions, resample indices over
    whole trajectories -- must stay bit-identical to the deterministic specification, no call may fail, and the handle must count
    the give-up and stop waiting inside the kernel (the reference's left-to-right `weight *=`,
    fastslam1.rs:250-256, is what the chunk order preserves)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"import sys; sys.path.insert(0, {root!r}); from tests.test_gpu_fs1_parity import degrade_in_process; degrade_in_process()"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=root, RR_FS1_FACTOR_WAIT_US="0"))
    assert r.returncode == 0 and "OBSERVE_DEGRADE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("algorithm", [1, 2])
def test_one_launch_plan_equals_the_two_kernel_plan(fs, algorithm, monkeypatch):
    """rr::k_quantize_plan_mark<FS_WEIGHTS> (integer image, gate, normalise-or-mark in one launch with the tile sums handed
    over in-kernel) against k_quantize_reduce + k_fs1_plan (RR_PF_FUSED_PLAN=0, read at the first update): the same
    weights, poses, maps, gate decisions and resample indices, bit for bit, with several tiles and a gate that opens
    on some updates only."""
    n, L, T = 30_000, 12, 10  # 15 tiles
    lms = scene(L, 43)

    def run(fused):
        monkeypatch.setenv("RR_PF_FUSED_PLAN", "1" if fused else "0")
        prm = fs.default_params()
        prm.first_obs_cov = 2.0
        prm.nth = n / 1.5
        if algorithm == 2:
            from rust_robotics_amd.slam import fastslam2

            f = fastslam2.FastSlam2(n, L, seed=78)
        else:
            f = fs.FastSlam1(n, L, params=prm, seed=78)
        out = []
        for t in range(T):
            z = np.ascontiguousarray(observations_for(fs, H.true_pose(t + 1), lms, seed=78, step=t))
            f.update([1.0, 0.1], z)
            fired = f.last_resample_fired()
            gp, gm = f.get_state()
            out.append((fired, f.last_resample_indices().copy() if fired else None, gp.copy(), gm.copy()))
        return out

    a, b = run(True), run(False)
    if algorithm == 1:
        fired = [x[0] for x in a]
        assert any(fired), fired
    for t, ((fa, ia, pa, ma), (fb, ib, pb, mb)) in enumerate(zip(a, b)):
        assert fa == fb, f"gate differs at update {t}"
        if fa:
            assert np.array_equal(ia, ib), f"indices differ at update {t}"
        assert bits_equal(pa, pb), f"poses / weights differ at update {t}"
        assert bits_equal(ma, mb), f"maps differ at update {t}"


def test_best_particle_reads_through_a_pending_resample(fs):
    """rr_fs1_best_particle while the last update's resample is still only markers (one launch, host mailbox, no settling):
    index, weight and pose must be those of the settled set that get_state produces afterwards (fastslam1.rs:269-274 after
    :205-234: equal weights 1/n, ties -> the last index, whose particle is a clone of some source)."""
    n, L = 3000, 10
    lms = scene(L, 53)
    prm = fs.default_params()
    prm.first_obs_cov = 2.0
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=81)
    fired = 0
    for t in range(12):
        f.update_async([1.0, 0.1], np.ascontiguousarray(observations_for(fs, H.true_pose(t + 1), lms, seed=81, step=t)))
        pose, w, i = f.best_particle()          # the resample (if the gate opened) is still pending here
        f.update_async([1.0, 0.1], np.ascontiguousarray(observations_for(fs, H.true_pose(t + 1), lms, seed=81, step=100 + t)))
        pose, w, i = f.best_particle()
        gp, _ = f.get_state()                   # settles
        fired += int(f.last_resample_fired())
        assert bits_equal(np.array([w, *pose]), gp[i]), f"round {t}"
        assert i == max(k for k in range(n) if gp[k, 0] == gp[:, 0].max())
        pose2, w2, i2 = f.best_particle()       # and once more on the settled set
        assert i2 == i and bits_equal(pose2, pose) and w2 == w
    assert fired > 0


def test_async_updates_equal_synchronised_updates(fs):
    """A run of rr_fs1_update_async calls with nothing in between takes the three-launch update: the first kernel resolves the
    previous plan's markers, moves the poses and carries the update's observations from a pinned ring slot to the device
    (k_fs1_resolve_predict), k_fs1_observe's last chunk forms the weights, the plan kernel follows.  With an accessor after
    every update the same filter takes k_fs1_resolve + k_fs1_predict and an H2D copy of the observations instead.  Same
    poses, weights and maps, bit for bit -- over more updates than the ring has slots, the observations different every time,
    with the host far ahead of the device (a slot overwritten too early would change the result)."""
    n, L, T = 150_000, 30, 100
    lms = scene(L, 47)
    zs = [np.ascontiguousarray(observations_for(fs, H.true_pose(t + 1), lms, seed=79, step=t)[(t % 4):]) for t in range(T)]

    def run(asynchronous):
        prm = fs.default_params()
        prm.first_obs_cov = 2.0
        prm.nth = n / 1.5
        f = fs.FastSlam1(n, L, params=prm, seed=79)
        fired = 0
        for t in range(T):
            if asynchronous:
                f.update_async([1.0, 0.1], zs[t])
            else:
                f.update([1.0, 0.1], zs[t])
                if f.last_resample_fired():
                    fired += 1
                    f.last_resample_indices()
        return f.get_state(), fired

    (pa, ma), _ = run(True)
    (pb, mb), fired = run(False)
    assert fired > 5
    assert bits_equal(pa, pb)
    assert bits_equal(ma, mb)


def test_trajectory_vs_literal_reference(fs, det, ref):
    n, L, T = 400, 6, 10
    lms = scene(L, 51, half=8.0)
    prm = fs.default_params()
    prm.first_obs_cov = 2.0
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=5, obs_chunks=1)
    px, py, pyaw = (np.zeros(n) for _ in range(3))
    pw = np.full(n, 0.01)
    lm = np.tile(np.array([0, 0, 1000.0, 0, 0, 1000.0]), (n, L, 1)).reshape(-1).copy()
    mr = oracle.ref_fs1_model()
    mr.init_cov = 2.0
    idx = np.empty(n, np.uint32)
    mism = 0
    for t in range(T):
        z = observations_for(fs, H.true_pose(t + 1), lms, seed=5, step=t)
        z0, z1 = np.empty(n), np.empty(n)
        det.det_normal2_v(5, 3, t, 0, n, dp(z0), dp(z1))
        rho = det.det_resample_rho(5, t)
        fired = ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 1.0, 0.1, dp(z0), dp(z1), dp(z), len(z),
                                   C.byref(mr), n / 1.5, rho / n, u32p(idx))
        f.update([1.0, 0.1], z)
        assert f.last_resample_fired() == bool(fired)
        if fired:
            mism += int(np.count_nonzero(f.last_resample_indices() != idx))
        if mism == 0:
            gp, gm = f.get_state()
            np.testing.assert_allclose(gp, np.column_stack([pw, px, py, pyaw]), **TOL)
            np.testing.assert_allclose(gm.reshape(-1), lm, **TOL)
    assert mism == 0


def test_reference_shim_and_no_panic(fs):
    """fastslam1.rs:363-379 through the caller-owned-vector shim"""
    particles = fs.create_particles(20, 3)
    landmarks = [(10.0, 0.0), (0.0, 10.0), (10.0, 10.0)]
    for t in range(5):
        z = fs.get_observations([0.0, 0.0, 0.0], landmarks, seed=3, step=t)
        fs.fastslam_update(particles, [1.0, 0.1], z, seed=3)
    assert len(particles) == 20
    assert all(math.isfinite(p.x) and math.isfinite(p.weight) for p in particles)
    best = fs.get_best_particle(particles)
    assert best.weight == max(p.weight for p in particles)
    ps = fs.create_particles(5, 3)
    for p, w in zip(ps, [0.1, 0.5, 0.9, 0.3, 0.2]):
        p.weight = w
    assert abs(fs.get_best_particle(ps).weight - 0.9) < np.finfo(float).eps


def test_invalid_observation_ids_are_rejected(fs):
    from rust_robotics_amd.core import RoboticsError

    f = fs.FastSlam1(16, 3)
    for bad in ([(1.0, 0.0, 3)], [(1.0, 0.0, -1)], [(float("nan"), 0.0, 0)], [(1.0, 0.0, 0.5)]):
        with pytest.raises(RoboticsError):
            f.update([1.0, 0.1], bad)
    f.update([1.0, 0.1], [(1.0, 0.0, 1), (2.0, 0.1, 1)])  # duplicate ids: processed sequentially
    assert f.counters()[2] == 1


def test_config3_size_properties(fs, det):
    """BASELINE configs[2] shape (1e5 x 200) for a few steps: size-independent properties"""
    n, L = 100_000, 200
    lms = scene(L, 61)
    prm = fs.default_params()
    prm.first_obs_cov = 0.5
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=8)
    for t in range(4):
        z = observations_for(fs, H.true_pose(t + 1, v=0.5), lms, seed=8, step=t)
        assert len(z) == L
        f.update([0.5, 0.1], z)
    poses = f.poses()
    assert np.all(np.isfinite(poses))
    assert abs(poses[:, 0].sum() - 1.0) < 1e-9
    pose, w, i = f.best_particle()
    assert w == poses[:, 0].max() and i == int(np.nonzero(poses[:, 0] == w)[0][-1])
    lm_best = f.landmarks_of(i)
    assert np.all(lm_best[:, 2] < 100.0) and np.all(np.isfinite(lm_best))
    assert np.median(np.hypot(lm_best[:, 0] - lms[:, 0], lm_best[:, 1] - lms[:, 1])) < 1.5
    # resample: survivors are exact copies (checked on the pose planes and one landmark column)
    before = f.poses()
    lm_before = f.landmarks_of(12345)
    f.resample_systematic(0.37)
    idx = f.last_resample_indices()
    after = f.poses()
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)
    assert bits_equal(after[:, 1:], before[idx, 1:]) and np.all(after[:, 0] == 1.0 / n)
    k = int(np.searchsorted(idx, 12345))
    if k < n and idx[k] == 12345:
        assert bits_equal(f.landmarks_of(k), lm_before)


def test_config4_full_size_on_one_gpu(fs):
    """BASELINE configs[3] at FULL size -- 1 000 000 particles x 200 landmarks (19.3 GB of maps, two buffer sets) -- unsharded
    on one GPU (the 8-GPU run splits exactly this filter into 125 000-particle blocks with the same bits): size-independent
    properties over a few updates, a forced resample whose survivors are exact copies, and the per-(particle, landmark)
    update count."""
    n, L = 1_000_000, 200
    lms = scene(L, 61)
    prm = fs.default_params()
    prm.first_obs_cov = 0.5
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=8)
    fired = []
    for t in range(3):
        z = observations_for(fs, H.true_pose(t + 1, v=0.5), lms, seed=8, step=t)
        assert len(z) == L
        f.update([0.5, 0.1], z)
        fired.append(f.last_resample_fired())
    poses = f.poses()
    assert poses.shape == (n, 4) and np.all(np.isfinite(poses))
    assert abs(poses[:, 0].sum() - 1.0) < 1e-9
    pose, w, i = f.best_particle()
    assert w == poses[:, 0].max() and i == int(np.nonzero(poses[:, 0] == w)[0][-1])  # ties -> last (fastslam1.rs:269-274)
    lm_best = f.landmarks_of(i)
    assert np.all(lm_best[:, 2] < 100.0) and np.all(np.isfinite(lm_best))
    assert np.median(np.hypot(lm_best[:, 0] - lms[:, 0], lm_best[:, 1] - lms[:, 1])) < 1.5
    before = f.poses()
    probe = 777_777
    lm_before = f.landmarks_of(probe)
    f.resample_systematic(0.37)
    idx = f.last_resample_indices()
    after = f.poses()
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)
    assert bits_equal(after[:, 1:], before[idx, 1:]) and np.all(after[:, 0] == 1.0 / n)
    cnt = np.bincount(idx, minlength=n)
    assert np.max(np.abs(cnt - n * before[:, 0])) <= 1.0 + 1e-6  # systematic: offspring within 1 of n w
    k = int(np.searchsorted(idx, probe))
    if k < n and idx[k] == probe:
        assert bits_equal(f.landmarks_of(k), lm_before)
    f.close()


def test_four_million_particles_times_200_landmarks(fs):
    """Beyond BASELINE configs[3]: 4 000 000 particles x 200 landmarks -- 77 GB of maps in two buffer sets, (3 + 6 L) N = 4.8e9 plane
    elements, i.e. past the 2^32 that one dimension of a launch can count (rr_fs1_create's fill was such a launch until round 6 and left
    most planes of a filter this size unwritten: tools/max_size_probe_fastslam.py found it; the probe takes the same run to
    14 000 000 particles = 270 GB).  Size-independent properties: a fresh filter's last plane holds the initial covariance, every
    landmark of the best particle is initialised by the first update and lies near the truth after three, weights are normalised,
    a forced resample's survivors are exact copies."""
    n, L = 4_000_000, 200
    if H.gpu_free_bytes() < 100e9:
        pytest.skip(f"needs 78 GB of device memory, {H.gpu_free_bytes() / 1e9:.0f} GB are free")
    lms = scene(L, 61)
    prm = fs.default_params()
    prm.first_obs_cov = 0.5
    prm.nth = n / 1.5
    f = fs.FastSlam1(n, L, params=prm, seed=8)
    for probe in (0, n // 2, n - 1):  # before any update: (0, 0, 1000 I) in every landmark of every particle (fastslam1.rs:34-41)
        m = f.landmarks_of(probe)
        assert np.all(m[:, :2] == 0.0) and np.all(m[:, 2] == 1000.0) and np.all(m[:, 5] == 1000.0) and np.all(m[:, 3:5] == 0.0), probe
    for t in range(3):
        z = observations_for(fs, H.true_pose(t + 1, v=0.5), lms, seed=8, step=t)
        assert len(z) == L
        f.update([0.5, 0.1], z)
    poses = f.poses()
    assert poses.shape == (n, 4) and np.all(np.isfinite(poses))
    assert abs(poses[:, 0].sum() - 1.0) < 1e-9
    pose, w, i = f.best_particle()
    assert w == poses[:, 0].max() and i == int(np.nonzero(poses[:, 0] == w)[0][-1])
    for probe in (i, n - 1):
        m = f.landmarks_of(probe)
        assert np.all(m[:, 2] < 100.0) and np.all(np.isfinite(m))
        assert np.median(np.hypot(m[:, 0] - lms[:, 0], m[:, 1] - lms[:, 1])) < 1.5
    before = f.poses()
    probe = n - 12_345
    lm_before = f.landmarks_of(probe)
    f.resample_systematic(0.37)
    idx = f.last_resample_indices()
    after = f.poses()
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)
    assert bits_equal(after[:, 1:], before[idx, 1:]) and np.all(after[:, 0] == 1.0 / n)
    k = int(np.searchsorted(idx, probe))
    if k < n and idx[k] == probe:
        assert bits_equal(f.landmarks_of(k), lm_before)
    f.close()


def test_an_observation_list_that_does_not_fit_the_lds_staging_is_refused_before_anything_runs(fs):
    """A chunk's observations are staged in LDS (24 bytes each, 150 KB): 8 000 distinct landmarks in one update are 64 chunks of 125 and
    fine; the same list with ONE landmark id repeated must run as one sequential chunk (fastslam1.rs:250-256's order for a landmark
    seen twice) and does not fit -- refused as an invalid parameter with nothing of the update done (until round 6 the observe launch
    failed after the predict had run, and HIP's sticky error failed the next call too)."""
    from rust_robotics_amd.core import RoboticsError

    L, n = 8000, 300
    lms = scene(L, 71)
    f = fs.FastSlam1(n, L, seed=3)
    z = observations_for(fs, H.true_pose(1, v=0.5), lms, seed=3, step=0)
    assert len(z) == L
    f.update([0.5, 0.1], z)
    assert f.counters()[2] == 64
    before, counters = f.poses(), f.counters()
    with pytest.raises(RoboticsError, match="too many fastslam observations") as ei:
        f.update([0.5, 0.1], np.vstack([z, z[:1]]))
    assert ei.value.kind == "InvalidParameter"
    assert f.counters() == counters and bits_equal(f.poses(), before), "a refused update must not have moved anything"
    f.update([0.5, 0.1], z)  # the handle is as usable as before
    assert np.all(np.isfinite(f.best_particle()[0]))
    f.update([0.5, 0.1], np.vstack([z[:6000], z[:1]]))  # 6 001 in one sequential chunk: 144 KB, fits
    assert f.counters()[2] == 1
    f.close()
