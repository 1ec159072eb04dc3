"""GPU parity of the KLD-adaptive MonteCarloLocalizer (rr_pf_create_adaptive) through the C ABI:
  * the adaptive resample against the literal restatement of monte_carlo_localization.rs:322-385
    and against the D-spec on identical uniforms: same particle count, same source indices;
  * whole steps against the D-spec with the engine's Philox streams: bit-exact particle sets;
  * the reference's own unit tests (:489-577) re-expressed against the engine."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from oracle import dp, u32p, u64p
from tests import helpers as H
from tests.test_kld_oracles import clouds, det_adaptive

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def loc():
    import rust_robotics_amd.localization as m

    return m


def make(loc, lo, hi, **kw):
    cfg = loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi, **kw)
    return loc.MonteCarloLocalizer(cfg, seed=13), cfg


@pytest.mark.parametrize("lo,hi", [(100, 1500), (1, 40), (50, 5000), (256, 3000)])
def test_adaptive_resample_matches_oracles(loc, det, ref, lo, hi):
    rng = np.random.default_rng(8)
    for name, x, y, yaw, w in clouds():
        n = x.size
        if n > hi:
            x, y, yaw, w = x[:hi], y[:hi], yaw[:hi], w[:hi] + (0.01 if w[:hi].sum() == 0 else 0.0)
            n = hi
        x, y, yaw = (np.ascontiguousarray(a) for a in (x, y, yaw))
        mcl, _ = make(loc, lo, hi)
        assert mcl.particle_count() == lo and mcl.particle_capacity() == hi
        mcl.set_particles_array(np.column_stack([x, y, yaw, np.zeros(n), w]))
        assert mcl.particle_count() == n
        r = np.floor(rng.random(hi) * 2**53) / 2**53
        n_new = mcl.resample_adaptive_with_uniforms(r)
        wn = np.ascontiguousarray(w / w.sum())
        idx_l = np.empty(hi, np.uint32)
        cnt_l = ref.ref_mcl_resample_adaptive(n, dp(x), dp(y), dp(yaw), dp(wn), dp(r), lo, hi, 0.05, 2.326, u32p(idx_l))
        cnt_d, idx_d = det_adaptive(det, x, y, yaw, np.ascontiguousarray(w), r, lo, hi)
        assert n_new == cnt_d == cnt_l, name
        assert mcl.particle_count() == n_new
        got = mcl.last_resample_indices()
        assert np.array_equal(got, idx_d) and np.array_equal(got, idx_l[:cnt_l]), name
        p = mcl.get_particles_array()
        assert p.shape == (n_new, 5)
        assert np.array_equal(p[:, 0].view(np.uint64), x[got].view(np.uint64))
        assert np.array_equal(p[:, 2].view(np.uint64), yaw[got].view(np.uint64))
        assert np.all(p[:, 4] == 1.0 / n_new)  # :359-362


def test_steps_bit_exact_vs_det(loc, det):
    """try_step x 25 (:291-300) with the engine's Philox streams against a D-spec loop on the CPU"""
    lo, hi, sig = 80, 900, 0.4
    cfg = loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi, range_noise=sig, velocity_noise=0.4,
                                           yaw_rate_noise=math.radians(8.0))
    mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=21)
    n = lo
    st = np.array([0.0, 0.0, 0.0, 1.0])
    x, y, yaw, v = (np.zeros(hi) for _ in range(4))
    det.det_pf_init(n, 21, 0, dp(st), dp(x), dp(y), dp(yaw), dp(v))
    w = np.zeros(hi)
    rng = np.random.default_rng(3)
    counts = []
    for t in range(25):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), sig, rng)
        mcl.try_step([1.0, 0.1], obs)
        det.det_pf_predict(n, dp(x), dp(y), dp(yaw), dp(v), 1.0, 0.1, 0.1, None, None, 21, t, 0, 0.4, math.radians(8.0))
        det.det_pf_weights(n, dp(x), dp(y), dp(w), dp(np.ascontiguousarray(obs)), len(obs), sig, 0)
        fx = H.det_fixed(det, w[:n].copy())
        cdf = H.det_cdf(det, w[:n].copy(), fx)
        idx = np.empty(hi, np.uint32)
        cnt = det.det_mcl_resample_adaptive(n, dp(x), dp(y), dp(yaw), u64p(cdf), int(cdf[-1]), None, 21, t, lo, hi, 0.05, 2.326, u32p(idx))
        for a in (x, y, yaw, v):
            a[:cnt] = a[idx[:cnt]]
        n = cnt
        counts.append(n)
        assert mcl.particle_count() == n, f"step {t}"
        p = mcl.get_particles_array()
        for k, a in enumerate((x, y, yaw, v)):
            assert np.array_equal(p[:, k].view(np.uint64), a[:n].view(np.uint64)), f"step {t} field {k}"
    assert len(set(counts)) > 1, counts  # the count really moved
    assert all(lo <= c <= hi for c in counts)


def test_reference_particle_count_adapts(loc):
    """monte_carlo_localization.rs:519-552"""
    mcl, cfg = make(loc, 100, 1500)
    n = 800
    i = np.arange(n)
    mcl.set_particles_array(np.column_stack([(i % 4) * 3.0 + i * 0.002, (i % 4) * 2.0, np.zeros(n), np.zeros(n), np.full(n, 1.0 / n)]))
    mcl.resample()
    expanded = mcl.particle_count()
    assert expanded > cfg.min_particles
    n = 600
    mcl.set_particles_array(np.column_stack([np.full(n, 1.0), np.full(n, 1.0), np.full(n, 0.1), np.zeros(n), np.full(n, 1.0 / n)]))
    mcl.resample()
    assert mcl.particle_count() <= expanded


def test_reference_count_stays_within_bounds_and_converges(loc):
    """monte_carlo_localization.rs:554-577 and :489-516"""
    cfg = loc.MonteCarloLocalizationConfig(min_particles=120, max_particles=600, velocity_noise=0.5, yaw_rate_noise=0.2)
    mcl = loc.MonteCarloLocalizer(cfg, seed=4)
    lms = [(0.0, 0.0), (15.0, 0.0), (8.0, 12.0)]
    truth = np.zeros(3)
    for _ in range(40):
        truth += [0.8 * math.cos(truth[2]) * 0.1, 0.8 * math.sin(truth[2]) * 0.1, 0.05 * 0.1]
        obs = [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms]
        mcl.try_step([0.8, 0.05], obs)
        assert 120 <= mcl.particle_count() <= 600
    cfg = loc.MonteCarloLocalizationConfig(min_particles=250, max_particles=1200, range_noise=0.25, velocity_noise=0.05,
                                           yaw_rate_noise=0.02)
    mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 0.0], cfg, seed=5)
    lms = [(0.0, 0.0), (10.0, 0.0), (5.0, 8.0)]
    truth = np.zeros(4)
    for _ in range(60):
        truth[:3] += [1.0 * math.cos(truth[2]) * 0.1, 1.0 * math.sin(truth[2]) * 0.1, 0.03 * 0.1]
        est = mcl.try_step([1.0, 0.03], [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms])
    assert math.hypot(est[0] - truth[0], est[1] - truth[1]) < 1.0


def test_validation_and_misuse(loc):
    inv = loc.RoboticsError
    for bad, msg in ((dict(min_particles=0), "min_particles must be greater than zero"),
                     (dict(min_particles=10, max_particles=5), "max_particles must be greater than or equal"),
                     (dict(kld_epsilon=0.0), "kld_epsilon must be positive"), (dict(kld_z=float("nan")), "kld_z must be positive")):
        with pytest.raises(inv) as ei:
            loc.MonteCarloLocalizer(loc.MonteCarloLocalizationConfig(**bad))
        assert msg in str(ei.value)
    mcl, _ = make(loc, 10, 50)
    with pytest.raises(inv):
        mcl.resample_with_uniforms(np.zeros(10))  # the fixed-N seam is refused
    with pytest.raises(inv):
        mcl.resample_adaptive_with_uniforms(np.zeros(7))  # needs max_particles uniforms
    with pytest.raises(inv):
        mcl.set_particles_array(np.zeros((51, 5)))  # beyond the capacity
    with pytest.raises(inv):
        loc.MonteCarloLocalizer(loc.MonteCarloLocalizationConfig(min_particles=10, max_particles=50), resample_scheme=1)


def test_large_adaptive_resample_matches_det(loc, det):
    """several tiles of particles, a capacity that needs many chunks of the counting kernel, thousands of
    occupied bins: count and indices against the sequential D-spec loop"""
    rng = np.random.default_rng(12)
    n, lo, hi = 30_000, 2_000, 120_000
    x, y, yaw = rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(-3, 3, n)
    w = rng.random(n) ** 4 + 1e-6
    mcl, _ = make(loc, lo, hi)
    mcl.set_particles_array(np.column_stack([x, y, yaw, np.zeros(n), w]))
    r = np.floor(rng.random(hi) * 2**53) / 2**53
    n_new = mcl.resample_adaptive_with_uniforms(r)
    cnt_d, idx_d = det_adaptive(det, np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(yaw), np.ascontiguousarray(w), r, lo, hi)
    assert n_new == cnt_d and lo < n_new <= hi
    assert np.array_equal(mcl.last_resample_indices(), idx_d)
    # and a second resample from the new (uniform-weight) set, Philox draws this time
    before = mcl.get_particles_array()
    mcl.resample()
    after = mcl.get_particles_array()
    assert lo <= mcl.particle_count() <= hi
    assert set(map(tuple, np.round(after[:50, :2], 12))) <= set(map(tuple, np.round(before[:, :2], 12)))


def test_asynchronous_steps_need_no_host_count(loc):
    """Round 3: an adaptive filter's particle count lives on the device (Ctl.n_active, read by every kernel of such a filter),
    so `step_async` does not wait for it.  40 asynchronous steps in a row -- the host never learns a count in between -- must
    leave exactly the particle set and the count of 40 synchronous steps (which read the count back every step), and the
    estimate / covariance accessors must agree."""
    lo, hi, sig = 80, 3000, 0.4
    cfg = loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi, range_noise=sig, velocity_noise=0.4, yaw_rate_noise=math.radians(8.0))
    a = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=23)
    b = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=23)
    rng = np.random.default_rng(5)
    counts = []
    for t in range(40):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), sig, rng)
        a.step_async([1.0, 0.1], obs)
        b.try_step([1.0, 0.1], obs)
        counts.append(b.particle_count())
    assert len(set(counts)) > 3, counts
    assert a.particle_count() == b.particle_count()
    pa, pb = a.get_particles_array(), b.get_particles_array()
    assert pa.shape == pb.shape and np.array_equal(pa.view(np.uint64), pb.view(np.uint64))
    np.testing.assert_allclose(a.estimate(), b.estimate(), rtol=0, atol=0)
    np.testing.assert_allclose(a.calc_covariance(), b.calc_covariance(), rtol=0, atol=0)
    # and on from there, mixing the two forms
    for t in range(40, 50):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), sig, rng)
        (a.step_async if t % 2 else a.try_step)([1.0, 0.1], obs)
        b.try_step([1.0, 0.1], obs)
    assert a.particle_count() == b.particle_count()
    assert np.array_equal(a.get_particles_array().view(np.uint64), b.get_particles_array().view(np.uint64))


@pytest.mark.parametrize("lo,hi", [(100, 5000), (60, 800)])
def test_fused_adaptive_step_equals_the_separate_launches(lo, hi):
    """The adaptive step of a small filter: k_mcl_adaptive_small does propagate + weight + integer image + CDF + plan in one
    workgroup -- and, up to 1 024 candidate draws, the draws, the bin table, the stop rule and the gather as well; beyond that
    those follow as three wide launches.  RR_MCL_SMALL=0 (read when the filter is created) keeps the six separate launches.
    Same particles, weights and counts after every step, bit for bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, math, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "import rust_robotics_amd.localization as loc\n"
        "cfg = loc.MonteCarloLocalizationConfig(min_particles=%d, max_particles=%d)\n"
        "mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=7)\n"
        "lms = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]\n"
        "truth = np.zeros(3); h = hashlib.sha256(); counts = []\n"
        "for t in range(60):\n"
        "    truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]\n"
        "    mcl.step_async([1.0, 0.1], [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms])\n"
        "    if t %% 7 == 6:\n"
        "        h.update(np.ascontiguousarray(mcl.get_particles_array()).tobytes()); counts.append(mcl.particle_count())\n"
        "print(h.hexdigest(), counts)\n") % (root, lo, hi)
    outs = []
    for small in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RR_MCL_SMALL=small), capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


@pytest.mark.parametrize("lo,hi", [(100, 5000), (100, 700), (1500, 4000)])
def test_try_step_returns_the_estimate_of_the_new_set(loc, lo, hi):
    """monte_carlo_localization.rs:291-300: try_step returns the mean of the resampled set.  For up to 1 024 particles the
    one-launch adaptive step forms it itself (as k_moments + k_moments_final would, reduction order included) and hands it
    over through the host mailbox; beyond that, and on the separate-launch route, rr_pf_estimate's kernels run.  The same
    bits either way."""
    mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi), seed=9)
    lms = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    truth = np.zeros(3)
    for t in range(40):
        truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
        e = np.asarray(mcl.try_step([1.0, 0.1], [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms]))
        e2 = np.asarray(mcl.estimate())
        assert np.array_equal(e.view(np.uint64), e2.view(np.uint64)), f"step {t}"
        assert lo <= mcl.particle_count() <= hi


def test_try_step_without_accessors_uses_the_new_count(loc):
    """rr_pf_step of an adaptive filter whose one-launch step hands over to the moment kernels (more than 1 024 particles
    after the resample): the host's particle count is stale at that point (the step left it on the device) and has to be
    refreshed before those kernels are sized.  Filter `a` calls nothing but try_step; filter `b` refreshes the host's copy
    with an accessor after every step.  Same estimates, bit for bit, with a count that moves above 1 024."""
    cfg = loc.MonteCarloLocalizationConfig(min_particles=100, max_particles=5000, range_noise=3.0)
    a = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=31)
    b = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=31)
    lms = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    truth = np.zeros(3)
    counts = []
    for t in range(60):
        truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
        obs = [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms]
        ea = np.asarray(a.try_step([1.0, 0.1], obs))
        eb = np.asarray(b.try_step([1.0, 0.1], obs))
        counts.append(b.particle_count())
        e2 = np.asarray(b.estimate())
        assert np.array_equal(eb.view(np.uint64), e2.view(np.uint64)), f"step {t}"
        assert np.array_equal(ea.view(np.uint64), eb.view(np.uint64)), f"step {t}: {ea} vs {eb} (count {counts[-1]})"
    assert max(counts) > 1024 and len(set(counts)) > 3, counts
    assert a.particle_count() == b.particle_count()
