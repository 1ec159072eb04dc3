"""The resident service (rr_pf_set_resident, csrc/resident_core.hpp): the small filter's step kernel stays on the device and is
fed steps through pinned memory.  Same kernel code as the launched step, so everything must be bit-identical to it -- through
idle exits, relaunches, accessors in between, varying observation counts, both resamplers, both gates, every register layout."""
import math
import time

import numpy as np
import pytest

from tests import helpers as H
from tests.test_gpu_small_n import bits, make

pytestmark = pytest.mark.gpu


def scenario(K, L=4, seed=4):
    lms = H.landmarks_grid(max(L, 1), 3)[:L] if L != 4 else H.REF_SCENE_LANDMARKS
    rng = np.random.default_rng(seed)
    obs = [H.observations(lms, H.true_pose(t + 1), 0.5, rng) if L else np.zeros((0, 3)) for t in range(K)]
    u = np.tile([1.0, 0.1], (K, 1))
    u[:, 0] += 0.01 * np.arange(K)
    return u, obs


@pytest.mark.parametrize("n", [1, 100, 120, 512, 1000, 1025, 2048])
@pytest.mark.parametrize("scheme,gated", [(0, True), (0, False), (1, True), (1, False)])
def test_resident_steps_equal_launched_steps(n, scheme, gated):
    import rust_robotics_amd.localization as loc

    K = 40
    u, obs = scenario(K)
    a = make(loc, n, scheme, gated)
    b = make(loc, n, scheme, gated)
    a.set_resident(5000.0)
    for t in range(K):
        ea = a.step(u[t], obs[t])
        eb = b.step(u[t], obs[t])
        assert np.array_equal(bits(ea), bits(eb)), f"estimate differs at step {t}: {ea} vs {eb}"
        if t % 13 == 12:  # an accessor parks the kernel; the next step starts it again
            assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array())), f"particles differ at step {t}"
            assert a.last_resample_fired() == b.last_resample_fired()
    launches, steps = a.resident_stats()
    assert steps == K and 1 <= launches <= 2 + K // 13 + 2, (launches, steps)
    assert a.counters() == b.counters() == (K, K)
    assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array()))
    assert np.array_equal(bits(a.raw_weights()), bits(b.raw_weights()))
    if a.last_resample_fired():
        assert np.array_equal(a.last_resample_indices(), b.last_resample_indices())
    np.testing.assert_array_equal(a.estimate(), b.estimate())
    np.testing.assert_array_equal(a.calc_covariance(), b.calc_covariance())
    assert a.n_eff() == b.n_eff()


def test_idle_exit_and_relaunch_lose_nothing():
    """The kernel leaves after idle_us without a step; the next step finds it gone (EXIT marker), starts a new incarnation and
    is served by it.  Asynchronous steps ride the same ring (one command in flight)."""
    import rust_robotics_amd.localization as loc

    K = 30
    u, obs = scenario(K)
    a = make(loc, 1000, 1, False)
    b = make(loc, 1000, 1, False)
    a.set_resident(300.0)  # 0.3 ms
    for t in range(K):
        if t % 3 == 1:
            a.step_async(u[t], obs[t])
            b.step_async(u[t], obs[t])
        else:
            assert np.array_equal(bits(a.step(u[t], obs[t])), bits(b.step(u[t], obs[t]))), t
        if t % 5 == 4:
            time.sleep(0.01)  # >> idle_us: the incarnation is gone when the next step comes
    launches, steps = a.resident_stats()
    assert steps == K and launches >= K // 5, (launches, steps)
    assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array()))
    a.set_resident(0.0)
    assert np.array_equal(bits(a.step(u[0], obs[0])), bits(b.step(u[0], obs[0])))
    assert a.resident_stats()[1] == K  # served by a launch, not by the service


def test_observation_count_may_change_from_step_to_step():
    import rust_robotics_amd.localization as loc

    a = make(loc, 700, 0, True)
    b = make(loc, 700, 0, True)
    a.set_resident(5000.0)
    rng = np.random.default_rng(2)
    for t, L in enumerate([4, 4, 0, 30, 30, 128, 3, 128, 1, 4, 200, 4]):  # 200 > the service's limit: that step is launched
        lms = H.landmarks_grid(max(L, 1), 3)[:L]
        o = H.observations(lms, H.true_pose(t + 1), 0.5, rng) if L else np.zeros((0, 3))
        assert np.array_equal(bits(a.step([1.0, 0.1], o)), bits(b.step([1.0, 0.1], o))), (t, L)
    assert a.resident_stats()[1] == 11
    assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array()))


def test_two_resident_filters_side_by_side_and_teardown():
    """Two handles with live resident kernels (each on its own stream) serve interleaved steps; destroying a handle with a live
    kernel asks it to leave first."""
    import rust_robotics_amd.localization as loc

    u, obs = scenario(20)
    pair = [make(loc, 150, 0, True, seed=s) for s in (3, 4)]
    ref = [make(loc, 150, 0, True, seed=s) for s in (3, 4)]
    for f in pair:
        f.set_resident(20000.0)
    for t in range(20):
        for f, r in zip(pair, ref):
            assert np.array_equal(bits(f.step(u[t], obs[t])), bits(r.step(u[t], obs[t])))
    del pair  # (live kernels)
    import gc

    gc.collect()
    c = make(loc, 150, 0, True, seed=3)
    assert np.all(np.isfinite(c.step(u[0], obs[0])))


def test_invalid_inputs_leave_the_service_untouched():
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd.core import RoboticsError

    a = make(loc, 100, 0, True)
    a.set_resident(5000.0)
    u, obs = scenario(3)
    a.step(u[0], obs[0])
    with pytest.raises(RoboticsError):
        a.step([math.nan, 0.0], obs[1])
    bad = obs[1].copy()
    bad[0, 0] = -1.0
    with pytest.raises(RoboticsError):
        a.step(u[1], bad)
    with pytest.raises(RoboticsError):
        a.set_resident(-1.0)
    assert a.counters() == (1, 1)
    b = make(loc, 100, 0, True)
    b.step(u[0], obs[0])
    assert np.array_equal(bits(a.step(u[1], obs[1])), bits(b.step(u[1], obs[1])))


@pytest.mark.parametrize("lo,hi,sig", [(100, 5000, 0.2), (100, 5000, 3.0), (60, 800, 0.2), (1500, 4000, 0.2)])
def test_resident_adaptive_mcl_equals_launched_steps(lo, hi, sig):
    """The MonteCarloLocalizer with the KLD-adaptive particle count (monte_carlo_localization.rs:322-385) through the resident
    service: k_mcl_adaptive_small stays on the device, the particle count never visits the host between steps, the next step's
    motion noise and candidate-draw uniforms are drawn while it waits.  Same estimates, counts and particles as launched steps."""
    import rust_robotics_amd.localization as loc

    cfg = loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi, range_noise=sig)
    a = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=31)
    b = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=31)
    a.set_resident(5000.0)
    lms = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    truth = np.zeros(3)
    counts = []
    for t in range(60):
        truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
        obs = [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms]
        if t % 4 == 3:
            a.step_async([1.0, 0.1], obs)
            b.step_async([1.0, 0.1], obs)
        else:
            ea, eb = np.asarray(a.try_step([1.0, 0.1], obs)), np.asarray(b.try_step([1.0, 0.1], obs))
            assert np.array_equal(bits(ea), bits(eb)), f"step {t}: {ea} vs {eb}"
        if t % 17 == 16:
            counts.append(a.particle_count())
            assert counts[-1] == b.particle_count()
            assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array())), f"particles differ at step {t}"
    launches, steps = a.resident_stats()
    assert steps == 60 and launches <= 8, (launches, steps)
    assert a.particle_count() == b.particle_count() and lo <= a.particle_count() <= hi
    assert np.array_equal(bits(a.get_particles_array()), bits(b.get_particles_array()))
    np.testing.assert_array_equal(a.estimate(), b.estimate())


def test_more_resident_filters_than_hardware_queues_still_make_progress():
    """HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): with more live resident kernels than
    that, a launch can land behind somebody else's resident kernel and only starts when that one has idled out.  Slower (one idle
    time per such step), never stuck, same bits."""
    import rust_robotics_amd.localization as loc

    u, obs = scenario(6)
    many = [make(loc, 100, 0, True, seed=s) for s in range(7)]
    ref = [make(loc, 100, 0, True, seed=s) for s in range(7)]
    for f in many:
        f.set_resident(400.0)
    t0 = time.time()
    for t in range(6):
        for f, r in zip(many, ref):
            assert np.array_equal(bits(f.step(u[t], obs[t])), bits(r.step(u[t], obs[t])))
    assert time.time() - t0 < 20.0
    for f, r in zip(many, ref):
        assert np.array_equal(bits(f.get_particles_array()), bits(r.get_particles_array()))
