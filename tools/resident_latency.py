"""Per-step latency of the synchronous try_step of a small filter: launched (one kernel launch + mailbox) vs the resident
service (rr_pf_set_resident).  Prints one JSON line."""
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_robotics_amd.localization as loc  # noqa: E402
from rust_robotics_amd import _ffi  # noqa: E402


def run(n, L, resident_us, steps=3000, scheme=0, gated=True):
    import ctypes as C

    if gated:
        cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        pf = loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=scheme)
    else:
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=scheme)
    if resident_us:
        pf.set_resident(resident_us)
    rng = np.random.default_rng(0)
    lms = rng.uniform(-20, 20, (L, 2))
    obs = np.ascontiguousarray(np.column_stack([np.hypot(lms[:, 0], lms[:, 1]), lms]))
    u = np.array([1.0, 0.1])
    out = np.empty(4)
    L_ = _ffi.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    h = pf._h
    up, op, outp = dp(u), dp(obs), dp(out)
    for _ in range(200):
        L_.rr_pf_step(h, up, op, L, outp)
    ts = np.empty(steps)
    for i in range(steps):
        t0 = time.perf_counter_ns()
        L_.rr_pf_step(h, up, op, L, outp)
        ts[i] = time.perf_counter_ns() - t0
    ts /= 1e3
    return {"n": n, "L": L, "resident_us": resident_us, "scheme": scheme, "gated": gated, "mean_us": round(float(ts.mean()), 2),
            "p50_us": round(float(np.median(ts)), 2), "p99_us": round(float(np.percentile(ts, 99)), 2), "min_us": round(float(ts.min()), 2),
            "stats": pf.resident_stats()}


def run_adaptive(lo, hi, resident_us, steps=2000):
    import ctypes as C

    mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi), seed=5)
    if resident_us:
        mcl.set_resident(resident_us)
    lms4 = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    truth, obs = np.zeros(3), []
    for _ in range(steps + 200):
        truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
        obs.append(np.ascontiguousarray([(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms4]))
    u, out = np.array([1.0, 0.1]), np.empty(4)
    L_ = _ffi.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    for t in range(200):
        L_.rr_pf_step(mcl._h, dp(u), dp(obs[t]), 4, dp(out))
    ts = np.empty(steps)
    for i in range(steps):
        o = dp(obs[200 + i])
        t0 = time.perf_counter_ns()
        L_.rr_pf_step(mcl._h, dp(u), o, 4, dp(out))
        ts[i] = time.perf_counter_ns() - t0
    ts /= 1e3
    return {"adaptive": [lo, hi], "resident_us": resident_us, "mean_us": round(float(ts.mean()), 2), "p50_us": round(float(np.median(ts)), 2),
            "p99_us": round(float(np.percentile(ts, 99)), 2), "particles_at_end": int(mcl.particle_count()), "stats": mcl.resident_stats()}


if __name__ == "__main__":
    rows = []
    for n, L in ((100, 3), (120, 4), (150, 5), (1000, 4), (2048, 4)):
        for res in (0.0, 5000.0):
            rows.append(run(n, L, res))
    rows.append(run(1000, 4, 5000.0, scheme=1, gated=False))
    rows.append(run(1000, 4, 0.0, scheme=1, gated=False))
    for lo, hi in ((100, 5000), (100, 1000)):
        for res in (0.0, 5000.0):
            rows.append(run_adaptive(lo, hi, res))
    print(json.dumps(rows))
