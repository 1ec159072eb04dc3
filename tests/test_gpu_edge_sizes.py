"""Edge sizes and long runs of the GPU engine: particle counts around every tile boundary the
kernels use (64, 256, 512, 2048), more observations than fit in the launch packet, no
observations, BASELINE configs[4]'s per-GPU shape, and a 400-step run."""
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p, u64p
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def loc():
    import rust_robotics_amd.localization as l

    return l


def bits_eq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


@pytest.mark.parametrize("n", [1, 2, 63, 65, 255, 257, 511, 513, 2047, 2049, 4097, 10_001])
@pytest.mark.parametrize("scheme", [0, 1])
def test_odd_particle_counts_bit_exact(loc, det, n, scheme):
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3,
                                           yaw_rate_noise=math.radians(5.0))
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=9, resample_scheme=scheme, record_indices=True)
    x, y, yaw, v = (np.empty(n) for _ in range(4))
    st = np.array([0.0, 0.0, 0.0, 1.0])
    det.det_pf_init(n, 9, 0, dp(st), dp(x), dp(y), dp(yaw), dp(v))
    d = H.DetPF(det, x, y, yaw, v, dt=0.1, sigma=0.5, sigma_v=0.3, sigma_w=math.radians(5.0), threshold=1.0, gate=1, scheme=scheme,
                lik=0, seed=9)
    rng = np.random.default_rng(10)
    for t in range(6):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        if t % 2:  # alternate the asynchronous (lazy) and the synchronous entry points
            pf.step_async([1.0, 0.1], obs)
        else:
            pf.step([1.0, 0.1], obs)
        d.step([1.0, 0.1], obs)
    got = pf.get_particles_array()
    assert np.array_equal(pf.last_resample_indices(), d.idx)
    for k, e in enumerate((d.x, d.y, d.yaw, d.v)):
        assert bits_eq(got[:, k], e), f"n={n} col {k}"
    assert np.all(got[:, 4] == 1.0 / n)


@pytest.mark.parametrize("L", [0, 1, 96, 97, 500])
def test_observation_counts_around_the_kernarg_limit(loc, det, L):
    n = 3000
    lms = H.landmarks_grid(max(L, 1), 3)[:L]
    pose = H.true_pose(5)
    obs = H.observations(lms, pose, 0.5, np.random.default_rng(4)) if L else np.zeros((0, 3))
    x, y, yaw, v = H.cloud(n, 5, center=(pose[0], pose[1], pose[2], 1.0))
    cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5)
    pf = loc.ParticleFilterLocalizer(cfg)
    pf.set_particles_array(H.aos(x, y, yaw, v, np.full(n, 1.0 / n)))
    pf.update_with_observations(obs)
    w = np.empty(n)
    det.det_pf_weights(n, dp(x), dp(y), dp(w), dp(np.ascontiguousarray(obs)), L, 0.5, 0)
    assert bits_eq(pf.raw_weights(), w)
    pf2 = loc.ParticleFilterLocalizer(cfg, seed=3, resample_scheme=1)
    e = pf2.step([1.0, 0.1], obs)
    assert np.all(np.isfinite(e))


@pytest.mark.parametrize("L", [2730, 2731, 6400])
def test_observation_counts_up_to_the_lds_limit(loc, det, L):
    """The observation block of a step is staged in LDS, 24 bytes per observation: 2 730 is the last count below 64 KB, 6 400 the
    engine's limit (150 KB of the CU's 160); one more is refused.  sigma is chosen so that 1 / (sigma sqrt(2 pi)) = e^0.5, i.e. the
    fused likelihood's exponent L ln c - sum diff^2 / 2 sigma^2 stays near 0 for particles next to the truth instead of running off
    to -L/2 (the linear-space product underflows at such counts whatever one does)."""
    n, sigma = 3000, 1.0 / (math.exp(0.5) * math.sqrt(2.0 * math.pi))
    lms = H.landmarks_grid(L, 3)
    pose = H.true_pose(5)
    rng = np.random.default_rng(4)
    obs = np.ascontiguousarray(H.observations(lms, pose, sigma, rng))
    x, y = pose[0] + rng.normal(0, 0.002, n), pose[1] + rng.normal(0, 0.002, n)
    yaw, v = np.full(n, pose[2]), np.ones(n)
    cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=sigma)
    pf = loc.ParticleFilterLocalizer(cfg)
    pf.set_particles_array(H.aos(x, y, yaw, v, np.full(n, 1.0 / n)))
    pf.update_with_observations(obs)
    w = np.empty(n)
    det.det_pf_weights(n, dp(x), dp(y), dp(w), dp(obs), L, sigma, 0)
    got = pf.raw_weights()
    assert bits_eq(got, w)
    assert np.count_nonzero(np.isfinite(got) & (got > 0.0)) > n // 2, "the comparison must not be one of zeros"
    pf2 = loc.ParticleFilterLocalizer(cfg, seed=3, resample_scheme=1)
    for _ in range(3):  # the lazy step kernels stage the same block
        e = pf2.step([1.0, 0.1], obs)
    assert np.all(np.isfinite(e))
    if L == 6400:
        more = np.vstack([obs, obs[:1]])
        with pytest.raises(loc.RoboticsError, match="too many observations"):
            pf.update_with_observations(more)
        with pytest.raises(loc.RoboticsError, match="too many observations"):
            pf2.step([1.0, 0.1], more)


def test_long_run_stays_locked_and_finite(loc):
    n = 20_000
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.3, velocity_noise=0.5,
                                           yaw_rate_noise=math.radians(10.0))
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=11, resample_scheme=1)
    rng = np.random.default_rng(12)
    for t in range(400):
        pf.step_async([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.3, rng))
    est = pf.estimate()
    assert np.all(np.isfinite(est))
    assert np.hypot(*(est[:2] - H.true_pose(400)[:2])) < 1.0
    p = pf.get_particles_array()
    assert np.all(np.isfinite(p)) and np.all(p[:, 4] == 1.0 / n)
    cov = pf.calc_covariance()
    assert np.all(np.diag(cov) >= 0)


def test_config5_per_gpu_shape_properties(loc, det):
    """BASELINE configs[4]: 1.6e7 particles x 64 landmarks over 8 GPUs = 2e6 x 64 per GPU"""
    n, L = 2_000_000, 64
    lms = H.landmarks_grid(L, 2)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=2, resample_scheme=1, record_indices=True)
    rng = np.random.default_rng(3)
    for t in range(4):
        pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
    pf.predict_with_control([1.0, 0.1])
    pf.update_with_observations(H.observations(lms, H.true_pose(5), 0.2, rng))
    before = pf.get_particles_array()
    raw = pf.raw_weights()
    assert abs(before[:, 4].sum() - 1.0) < 1e-9
    pf.resample()
    idx = pf.last_resample_indices()
    after = pf.get_particles_array()
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)
    assert np.array_equal(after[:, :4].view(np.uint64), before[idx, :4].view(np.uint64))
    cnt = np.bincount(idx, minlength=n)
    assert np.max(np.abs(cnt - n * before[:, 4])) <= 1.0 + 1e-6  # systematic: offspring within 1 of n*w
    fx = H.det_fixed(det, raw)
    cdf = H.det_cdf(det, raw, fx)
    e = np.empty(n, np.uint32)
    rstep = pf.counters()[1] - 1
    det.det_indices_systematic(n, u64p(cdf), fx["total"], n, 0, n, det.det_resample_rho(2, rstep), u32p(e))
    assert np.array_equal(idx, e)


def test_fastslam_without_landmarks():
    """create_particles(n, 0) is legal in the reference (fastslam1.rs:302-306): poses only.  The AoS pose
    image (4 doubles per particle) must not be staged in a buffer set that holds 3 planes."""
    from rust_robotics_amd.slam import fastslam1 as fs

    n = 3000
    f = fs.FastSlam1(n, 0, seed=5)
    rng = np.random.default_rng(6)
    poses = np.column_stack([np.full(n, 1.0 / n), rng.normal(0, 1, n), rng.normal(0, 1, n), rng.uniform(-3, 3, n)])
    for _ in range(2):  # both buffer sets take a turn as "the inactive one"
        f.set_state(poses, None)
        got = f.poses()
        assert bits_eq(got, poses)
        f.update([1.0, 0.1], [])
        f.resample_systematic(0.25)
        moved = f.poses()
        assert np.all(np.isfinite(moved)) and moved.shape == (n, 4)
        assert np.allclose(moved[:, 0], 1.0 / n)
    pose, w, i = f.best_particle()
    assert 0 <= i < n and np.all(np.isfinite(pose))


def test_index_drift_against_the_literal_walk_at_1e6(loc):
    """the integer CDF picks the same particle as the reference's serial float cumsum except
    for draws within the cumsum's own accumulated rounding error of a step.  Bounded here on the DEVICE's own
    weights at BASELINE size: identical draws into the engine (resample seams) and into the literal restatement
    (ref_mcl_resample_indices = monte_carlo_localization.rs:328-392; ref_fs1_resample_indices = fastslam1.rs:205-234);
    at most a handful of the 1e6 slots differ and every differing slot picks the neighbouring particle."""
    n, L = 1_000_000, 32
    ref = oracle.ref()
    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(17)
    cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.2)
    for scheme in (0, 1):
        pf = loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=3, resample_scheme=scheme, record_indices=True)
        for t in range(3):  # a few real steps so that the cloud and the weights are those of a tracking filter
            pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
        pf.predict_with_control([1.0, 0.1])
        pf.update_with_observations(H.observations(lms, H.true_pose(4), 0.2, rng))
        w = pf.get_particles_array()[:, 4].copy()  # normalised weights as the reference would hold them
        nz = np.count_nonzero(w)
        assert nz > n // 100
        if scheme == 0:
            r = np.floor(rng.random(n) * 2**53) / 2**53
            pf.resample_with_uniforms(r)
            lit = np.empty(n, np.uint32)
            ref.ref_mcl_resample_indices(n, dp(w), dp(r), u32p(lit))
        else:
            rho = float(np.floor(rng.random() * 2**53) / 2**53)
            pf.resample_systematic(rho)
            lit = np.empty(n, np.uint32)
            ref.ref_fs1_resample_indices(n, dp(w.copy()), rho / n, u32p(lit))
        got = pf.last_resample_indices().astype(np.int64)
        lit = lit.astype(np.int64)
        diff = np.nonzero(got != lit)[0]
        assert diff.size <= 8, f"scheme {scheme}: {diff.size} of {n} slots differ from the literal walk"
        zeros_before = np.concatenate([[0], np.cumsum(w == 0.0)])
        for k in diff:  # neighbours, skipping zero-weight particles in between (they feed no slot in either walk)
            a, b = sorted((int(got[k]), int(lit[k])))
            assert (b - a) - (zeros_before[b] - zeros_before[a + 1]) <= 1, (k, a, b)


def test_config5_full_size_on_one_gpu(loc):
    """BASELINE configs[4] at FULL size -- 16 000 000 particles x 64 landmarks -- unsharded on one GPU: 7813 scan tiles, i.e.
    beyond the fused plan kernel's limit (k_scan_tiles + k_mark instead of k_plan_mark), eager gather.  Properties that do
    not depend on the size: normalised weights, systematic offspring within 1 of n w, survivors are exact copies, the
    estimate tracks the truth."""
    n, L = 16_000_000, 64
    lms = H.landmarks_grid(L, 2)
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=2, resample_scheme=1, record_indices=True)
    rng = np.random.default_rng(3)
    for t in range(3):
        pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
    est = pf.estimate()
    assert np.all(np.isfinite(est)) and np.hypot(*(est[:2] - H.true_pose(3)[:2])) < 0.5
    pf.predict_with_control([1.0, 0.1])
    pf.update_with_observations(H.observations(lms, H.true_pose(4), 0.2, rng))
    before = pf.get_particles_array()
    assert abs(before[:, 4].sum() - 1.0) < 1e-9
    pf.resample()
    idx = pf.last_resample_indices()
    after = pf.get_particles_array()
    assert np.all(np.diff(idx.astype(np.int64)) >= 0)
    assert np.array_equal(after[:, :4].view(np.uint64), before[idx, :4].view(np.uint64))
    cnt = np.bincount(idx, minlength=n)
    assert np.max(np.abs(cnt - n * before[:, 4])) <= 1.0 + 1e-6
    assert np.all(after[:, 4] == 1.0 / n)


def test_a_billion_particles_on_one_gpu(loc):
    """The largest sizes: 10^9 particles (84 GB of the 288) unsharded on one GPU -- 62.5 x BASELINE configs[4], 488 282 scan tiles, every
    index beyond 2^24 and the byte offsets beyond 2^32 -- through the asynchronous step with the systematic resampler every step, checked
    by what needs no N-sized copy to the host: the mean tracks the truth, N_eff lies in (0, N], the covariance is finite with a
    non-negative diagonal, the counters count.  (tools/max_size_probe.py takes the same run to the ABI's limit, 2^31 - 1 particles =
    181 GB: profiles/r06z5_max_size_probe.jsonl.)  One particle more than the limit is refused with the reference-style message."""
    n, steps = 1_000_000_000, 4
    if H.gpu_free_bytes() < 100e9:
        pytest.skip(f"needs 84 GB of device memory, {H.gpu_free_bytes() / 1e9:.0f} GB are free")
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=5, resample_scheme=1)
    assert pf.particle_count() == n
    rng = np.random.default_rng(6)
    for t in range(steps):
        pf.step_async([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.2, rng))
    est = pf.estimate()
    truth = H.true_pose(steps)
    assert np.all(np.isfinite(est)) and np.hypot(est[0] - truth[0], est[1] - truth[1]) < 0.5
    cov = pf.calc_covariance()
    assert np.all(np.isfinite(cov)) and np.all(np.diag(cov) >= 0.0)
    pf.predict_with_control([1.0, 0.1])
    pf.update_with_observations(H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(steps + 1), 0.2, rng))
    assert 0.0 < pf.n_eff() <= n
    assert pf.counters()[0] == steps + 1
    del pf
    with pytest.raises(loc.RoboticsError, match="below 2\\^31"):
        loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(min_particles=2**31, max_particles=2**31), seed=5, resample_scheme=1)
