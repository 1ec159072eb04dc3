"""worker of tests/test_gpu_plan_degrade.py.

  serial <kind>            RR_PF_PLAN_TIMEOUT_US=0 is set by the parent: every workgroup of the one-launch resample plan that
                           does not find the sums ready at its first look gives up, so the launch degrades to the serial plan
                           on an idle device.  The run must equal, bit for bit, the multi-launch plan (RR_PF_FUSED_PLAN=0).
  contend <dir> <who>      one of two processes that step a 1e6-particle MCL filter on the SAME GPU at the same time (each
                           one's plan kernels spin and hold CU slots the other one's missing workgroups need); prints a
                           digest of its final particle set and its plan statistics.
"""
import hashlib
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import helpers as H  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def mcl_run(n, L, T, fused, gated, estimates=True):
    import rust_robotics_amd.localization as loc

    os.environ["RR_PF_FUSED_PLAN"] = "1" if fused else "0"
    kw = dict(seed=11, resample_scheme=1, record_indices=True)
    if gated:
        pf = loc.ParticleFilterLocalizer(loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, resample_threshold=0.5), **kw)
    else:
        pf = loc.MonteCarloLocalizer(loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5), **kw)
    lms = H.landmarks_grid(L, 5)
    rng = np.random.default_rng(12)
    out = []
    for phase in range(2):  # phase 0: no host read in between (every launch on the device-side path); phase 1: after the host has noticed
        for t in range(T):
            obs = H.observations(lms, H.true_pose(phase * T + t + 1), 0.5, rng)
            if estimates and t % 2:
                pf.step_async_estimate([1.0, 0.1], obs)
            else:
                pf.step_async([1.0, 0.1], obs)
        est = pf.last_step_estimate() if estimates else None
        stats = pf.plan_stats()
        out.append(dict(particles=pf.get_particles_array(), idx=pf.last_resample_indices(), fired=pf.last_resample_fired(), est=est,
                        stats=stats))
    return out


def serial_pf(gated):
    n, L, T = 300_000, 8, 7  # 147 tiles
    a = mcl_run(n, L, T, fused=True, gated=gated)
    b = mcl_run(n, L, T, fused=False, gated=gated)
    g0, on0 = a[0]["stats"]
    assert g0 >= 1, f"no launch degraded although nobody may wait: {a[0]['stats']}"
    assert not on0, "the handle should have left the one-launch plan at its first host read"
    assert b[0]["stats"] == (0, False)
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(bits(x["particles"]), bits(y["particles"])), f"phase {k}: particles differ"
        assert x["fired"] == y["fired"] and np.array_equal(x["idx"], y["idx"]), f"phase {k}: resample differs"
        assert np.array_equal(bits(x["est"]), bits(y["est"])), f"phase {k}: in-step estimate differs"
    print("PLAN_SERIAL_OK", g0)


def serial_fs():
    from rust_robotics_amd.slam import fastslam1 as fs

    n, L, T = 300_000, 3, 6
    lms = np.random.default_rng(3).uniform(-8, 8, size=(L, 2))

    def run(fused):
        os.environ["RR_PF_FUSED_PLAN"] = "1" if fused else "0"
        prm = fs.default_params()
        prm.first_obs_cov = 0.5
        prm.nth = n / 1.5
        f = fs.FastSlam1(n, L, params=prm, seed=4)
        out = []
        for phase in range(2):
            for t in range(T):
                z = np.array(fs.get_observations(H.true_pose(phase * T + t + 1, v=0.5), [tuple(p) for p in lms], seed=4, step=phase * T + t)).reshape(-1, 3)
                f.update_async([0.5, 0.1], z)
            stats = f.plan_stats()
            out.append((f.get_state(), f.last_resample_fired(), stats))
        return out

    a, b = run(True), run(False)
    assert a[0][2][0] >= 1 and not a[0][2][1], a[0][2]
    for k, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(bits(x[0][0]), bits(y[0][0])), f"phase {k}: poses / weights differ"
        assert np.array_equal(bits(x[0][1]), bits(y[0][1])), f"phase {k}: maps differ"
        assert x[1] == y[1]
    print("PLAN_SERIAL_OK", a[0][2][0])


def contend(box, who):
    import rust_robotics_amd.localization as loc

    n, L, T = 1_000_000, 32, 1500
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(2)
    obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(T)]
    pf.step_async([1.0, 0.1], obs[0])  # code objects loaded, first launch done
    pf.synchronize()
    open(os.path.join(box, f"ready{who}"), "w").close()
    other = os.path.join(box, f"ready{1 - int(who)}")
    t0 = time.time()
    while not os.path.exists(other):
        if time.time() - t0 > 120:
            raise SystemExit("the other process never got ready")
        time.sleep(0.001)
    t0 = time.time()
    for t in range(1, T):
        pf.step_async([1.0, 0.1], obs[t])
    pf.synchronize()  # raises if the engine reports an error
    dt = time.time() - t0
    giveups, on = pf.plan_stats()
    p = pf.get_particles_array()
    print("CONTEND", who, hashlib.sha256(bits(p).tobytes()).hexdigest(), giveups, int(on), f"{dt:.3f}")


def contend_reference():
    """the digest both contenders must produce: the same trajectory, alone on the device, multi-launch plan"""
    import rust_robotics_amd.localization as loc

    os.environ["RR_PF_FUSED_PLAN"] = "0"
    n, L, T = 1_000_000, 32, 1500
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(2)
    for t in range(T):
        pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
    p = pf.get_particles_array()
    print("REFERENCE", hashlib.sha256(bits(p).tobytes()).hexdigest())


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "serial":
        {"pf": lambda: serial_pf(True), "mcl": lambda: serial_pf(False), "fs": serial_fs}[sys.argv[2]]()
    elif what == "contend":
        contend(sys.argv[2], sys.argv[3])
    elif what == "reference":
        contend_reference()
