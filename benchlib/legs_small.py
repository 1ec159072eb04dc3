"""`small_n`: the synchronous try_step at the sizes the reference's own callers run (100 - 1 000 particles), launched and resident,
with the reference's loop on one CPU core beside it."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import BENCH_PY, ROOT  # noqa: F401
from .common import host_cpu, pin_to_gpu_numa_node


SMALL_ROWS = [
    # (key, particles, landmarks [x, y], config overrides, initial state, exact observations?, where the reference runs this size)
    ("100x3", 100, [(5.0, 0.0), (0.0, 5.0), (5.0, 5.0)], {}, (0.0, 0.0, 0.0, 0.0), True,
     "headless_localizers.rs:29-56: ParticleFilterConfig::default() -- 100 particles, 3 landmarks, exact ranges, try_step_state every step"),
    ("120x5", 120, [(2.0, 2.0), (10.0, 2.0), (2.0, 8.0), (10.0, 8.0), (6.0, 5.0)], {"range_noise": 0.25}, (5.0, 5.0, 0.0, 0.0), False,
     "rust_robotics_playground/src/localization.rs:50-66: 120 particles, 5 landmarks, range_noise 0.25"),
    ("150x5", 150, [(2.0, 2.0), (10.0, 2.0), (2.0, 8.0), (10.0, 8.0), (6.0, 5.0)], {"range_noise": 0.25}, (5.0, 5.0, 0.0, 0.0), False,
     "render_gif_particle_filter.rs:25-40: 150 particles, 5 landmarks, range_noise 0.25"),
    ("1000x4", 1000, [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)], {"range_noise": 0.5, "velocity_noise": 0.3, "yaw_rate_noise": math.radians(5.0)},
     (0.0, 0.0, 0.0, 0.0), False, "BASELINE.json configs[0] / tests/unified_filter_comparison.rs:43,278-285: 1 000 particles, 4 landmarks"),
]


def leg_small_n(with_cpu):
    """The sizes the reference's own callers run (SMALL_ROWS: 100 - 1 000 particles, 3 - 5 landmarks), ParticleFilterLocalizer
    semantics (multinomial resample behind the N_eff gate, particle_filter.rs:337-345,441-473), through the entry point those
    callers use -- the SYNCHRONOUS try_step -- in both of its forms: one launch of one workgroup per step (k_step_small + host
    mailbox) and the resident service (rr_pf_set_resident: the kernel stays, steps travel through pinned memory); beside them
    the asynchronous and the batched forms, and the literal restatement of the reference's try_step loop (cache refreshes
    included) on ONE host core, timed inside C.  Calls go through ctypes with prebuilt argument pointers (~1 us of call overhead
    stays in every GPU number)."""
    import ctypes as C

    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi

    lib = _ffi.lib()
    K = 2000
    pinned = pin_to_gpu_numa_node(0)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    out = {"steps": K, "unit_rows": "microseconds per step", "host_pinned_to_cpus": pinned, "rows": {}}

    for key, n, lms, over, init, exact, where in SMALL_ROWS:
        L = len(lms)
        cfg = loc.ParticleFilterConfig(n_particles=n, **over)
        rng = np.random.default_rng(42)
        truth = np.array(init[:3], dtype=np.float64)
        obs = np.empty((K, L, 3))
        for t in range(K):
            truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
            for q, (lx, ly) in enumerate(lms):
                d = math.hypot(truth[0] - lx, truth[1] - ly)
                obs[t, q] = (d if exact else max(d + rng.normal(0.0, cfg.range_noise), 0.0), lx, ly)
        u = np.tile([1.0, 0.1], (K, 1))
        est = np.empty(4)

        def fresh(resident_us=0.0):
            pf = loc.ParticleFilterLocalizer.with_initial_state(list(init), cfg, seed=42)
            if resident_us:
                pf.set_resident(resident_us)
            return pf

        def best_of(make, body, reps=3):
            """body(pf) -> microseconds per step, measured around its own loop (argument pointers are built outside it)"""
            best = None
            for _ in range(reps):
                pf = make()
                for t in range(300):  # warm: clocks, code, the resident incarnation
                    lib.rr_pf_step(pf._h, dp(u[t]), dp(obs[t]), L, dp(est))
                dt = body(pf)
                best = dt if best is None else min(best, dt)
                del pf
            return best

        def sync_loop(pf):
            h, e = pf._h, dp(est)
            ptrs = [(dp(u[t]), dp(obs[t])) for t in range(K)]
            t0 = time.perf_counter()
            for up, op in ptrs:
                lib.rr_pf_step(h, up, op, L, e)
            return (time.perf_counter() - t0) / K * 1e6

        def async_loop(pf):
            h = pf._h
            ptrs = [(dp(u[t]), dp(obs[t])) for t in range(K)]
            t0 = time.perf_counter()
            for up, op in ptrs:
                lib.rr_pf_step_async(h, up, op, L)
            pf.synchronize()
            return (time.perf_counter() - t0) / K * 1e6

        def many_loop(pf):
            t0 = time.perf_counter()
            pf.step_many(u, obs)
            return (time.perf_counter() - t0) / K * 1e6

        row = {"config": where, "particles": n, "landmarks": L}
        row["try_step, launched (one launch + mailbox per step)"] = best_of(fresh, sync_loop)
        row["try_step, resident service"] = best_of(lambda: fresh(5000.0), sync_loop)
        row["step_async, launched"] = best_of(fresh, async_loop)
        row["step_many (one launch for all steps, estimates read back at the end)"] = best_of(fresh, many_loop)
        if with_cpu:
            import oracle
            from oracle import dp as odp, u32p

            ref, det = oracle.ref(), oracle.det()
            ref.ref_set_threads(1)
            x, y, yaw, v = (np.full(n, init[k]) for k in range(4))
            w = np.full(n, 1.0 / n)
            idx, est_k = np.empty(n, np.uint32), np.empty((K, 4))
            nv, nw, r = np.empty((K, n)), np.empty((K, n)), np.empty((K, n))
            z0, z1, r2 = np.empty(n), np.empty(n), np.empty(n)
            for t in range(K):
                det.det_normal2_v(42, 3, t, 0, n, odp(z0), odp(z1))
                det.det_uniform2_v(42, 4, t, 0, n, odp(r[t]), odp(r2))
                nv[t], nw[t] = cfg.velocity_noise * z0, cfg.yaw_rate_noise * z1
            best = None
            for _ in range(3):
                for arr, k in ((x, 0), (y, 1), (yaw, 2), (v, 3)):
                    arr[:] = init[k]
                w[:] = 1.0 / n
                sec = ref.ref_pf_try_step_loop(n, odp(x), odp(y), odp(yaw), odp(v), odp(w), odp(u), cfg.dt, odp(nv), odp(nw), odp(obs), L, cfg.range_noise,
                                               cfg.resample_threshold, 0, odp(r), u32p(idx), K, odp(est_k), 1, 1)
                best = sec if best is None else min(best, sec)
            row["cpu: the reference's try_step loop, one core"] = best / K * 1e6
        out["rows"][key] = row

    head = out["rows"]["1000x4"]
    batched = head["step_many (one launch for all steps, estimates read back at the end)"]
    out["config"] = {"workload": "particle filter at the reference's own sizes (BASELINE.json configs[0] = row 1000x4): multinomial resample behind "
                                 "the N_eff gate, synchronous try_step", "particles": 1000, "landmarks": 4}
    out["value"] = 1000 * 4 / (batched * 1e-6)
    out["unit"] = "particle-landmark updates/s"
    out["note"] = ("value = rr_pf_step_many at 1000 x 4 with the per-step estimates (a single workgroup on one of 256 CUs: the work of a step does not "
                   "fill more); the rows are what a caller of try_step sees per step")
    if with_cpu:
        model, nproc = host_cpu()
        cpu_us = head["cpu: the reference's try_step loop, one core"]
        out["cpu_baseline"] = {"value": 1000 * 4 / (cpu_us * 1e-6), "unit": "particle-landmark updates/s", "cores": 1, "kind": "port", "us_per_step": cpu_us,
                               "sample": f"oracle/ref_literal.c ref_pf_try_step_loop: try_step as the reference runs it -- predict, update, its O(N^2) linear-scan "
                                         f"resample behind the N_eff gate (particle_filter.rs:455-470) and refresh_cache (mean + covariance) after predict, update and "
                                         f"resample (:299,332,343) -- {K} steps per row on one core ({model}), timed inside C, noise samples pre-drawn (the reference's "
                                         f"RNG is not in the loop: a lower bound of its cost)"}
    return out
