#!/usr/bin/env python3
"""In-kernel timeline of a resident step (csrc/resident_core.hpp, k_step_small resident mode), from the instrumented build:
    make -C rust_robotics_amd/csrc timeline
    RR_AMD_LIBRARY=rust_robotics_amd/librust_robotics_amd_timeline.so python tools/resident_timeline.py
Stamps are the device's 100 MHz wall clock (10 ns); host time is perf_counter around the same rr_pf_step call.  Prints JSON."""
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rust_robotics_amd.localization as loc  # noqa: E402
from rust_robotics_amd import _ffi  # noqa: E402

PHASES = ["wait for the command (host turnaround + link)", "propagate + weight", "block maximum", "integer image, scans, gate",
          "resample + gather", "estimate sums + answer issued"]


def run(n, L, scheme=0, gated=True, steps=4000):
    lib = _ffi.lib()
    lib.rr_pf_debug_resident_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.rr_pf_debug_resident_timeline.restype = C.c_int
    if gated:
        cfg = loc.ParticleFilterConfig(n_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        pf = loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=scheme)
    else:
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=scheme)
    pf.set_resident(5000.0)
    rng = np.random.default_rng(0)
    lms = rng.uniform(-20, 20, (L, 2))
    obs = np.ascontiguousarray(np.column_stack([np.hypot(lms[:, 0], lms[:, 1]), lms]))
    u, out = np.array([1.0, 0.1]), np.empty(4)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    tl = (C.c_uint64 * 8)()
    rows, host = [], []
    for i in range(steps + 200):
        t0 = time.perf_counter_ns()
        lib.rr_pf_step(pf._h, dp(u), dp(obs), L, dp(out))
        dt = time.perf_counter_ns() - t0
        lib.rr_pf_debug_resident_timeline(pf._h, tl)
        if i >= 200:
            rows.append(list(tl))
            host.append(dt / 1e3)
    a = np.array(rows, dtype=np.float64)
    d = np.diff(a[:, :7], axis=1) / 100.0  # us
    fired = a[:, 7] != 0
    res = {"n": n, "L": L, "scheme": scheme, "gated": gated, "host_us_per_step_mean": round(float(np.mean(host)), 2),
           "fired_fraction": round(float(fired.mean()), 3)}
    for name, sel in (("all", np.ones(len(a), bool)), ("fired", fired), ("not_fired", ~fired)):
        if sel.any():
            res[name] = {ph: round(float(d[sel, k].mean()), 2) for k, ph in enumerate(PHASES)}
            res[name]["device total after the command"] = round(float(d[sel, 1:].sum(axis=1).mean()), 2)
    # time from one answer to the next step's random numbers being ready = the precompute
    res["precompute_us (answer issued -> next step's random numbers drawn)"] = round(float(((a[1:, 0] - a[:-1, 6]) / 100.0).mean()), 2)
    return res


APHASES = ["wait for the command (host turnaround + link)", "propagate + weight + maximum", "integer image, CDF, plan", "draws, bin table, stop rule",
           "table wipe + gather", "estimate + answer issued"]


def run_adaptive(lo, hi, steps=3000):
    lib = _ffi.lib()
    lib.rr_pf_debug_resident_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.rr_pf_debug_resident_timeline.restype = C.c_int
    mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi), seed=5)
    mcl.set_resident(5000.0)
    lms4 = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]
    truth = np.zeros(3)
    u, out = np.array([1.0, 0.1]), np.empty(4)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    tl = (C.c_uint64 * 8)()
    rows = []
    for i in range(steps + 200):
        truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
        obs = np.ascontiguousarray([(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms4])
        lib.rr_pf_step(mcl._h, dp(u), dp(obs), 4, dp(out))
        lib.rr_pf_debug_resident_timeline(mcl._h, tl)
        if i >= 200:
            rows.append(list(tl))
    a = np.array(rows, dtype=np.float64)
    d = np.diff(a[:, :7], axis=1) / 100.0
    res = {"adaptive": [lo, hi], "particles_mean": round(float(a[:, 7].mean()), 1)}
    res["phases_us"] = {ph: round(float(d[:, k].mean()), 2) for k, ph in enumerate(APHASES)}
    res["device total after the command"] = round(float(d[:, 1:].sum(axis=1).mean()), 2)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "adaptive":
        print(json.dumps([run_adaptive(100, 5000), run_adaptive(100, 1000)], indent=1))
        sys.exit(0)
    print(json.dumps([run(100, 3), run(1000, 4), run(1000, 4, scheme=1, gated=False), run(2048, 4)], indent=1))
