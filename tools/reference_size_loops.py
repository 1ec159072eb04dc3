#!/usr/bin/env python3
"""What the reference's own loops cost per iteration on this engine, at the reference's own sizes (DESIGN.md section 4,
INTEGRATION.md section 4): the particle filter's try_step at 1 000 x 4, the MonteCarloLocalizer's adaptive try_step with the
default configuration (100 - 5 000 particles), FastSLAM 1.0's update followed by the best particle at 100 x 8, and the
synchronous step of larger filters.  Prints one JSON object; `python tools/reference_size_loops.py > out.json` on a GPU box."""
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, k):
    t0 = time.perf_counter()
    for t in range(k):
        fn(t)
    return (time.perf_counter() - t0) / k * 1e6


def main():
    import bench
    import oracle
    import rust_robotics_amd.localization as loc
    from oracle import dp, u32p
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests import helpers as H

    out = {"unit": "microseconds per iteration, Python caller (ctypes)", "clock": "time.perf_counter around 1000 iterations after a warm-up",
           "host_pinned_to_cpus": bench.pin_to_gpu_numa_node(0),
           "columns": "launched = one or more kernel launches per call; resident = rr_pf_set_resident / rr_fs1_set_resident (the kernel stays on the "
                      "device, steps travel through pinned memory); cpu = oracle/ref_literal.c, the reference's loop restated, one host core"}
    # ---- particle filter, tests/unified_filter_comparison.rs:286-295
    lms4 = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]

    def obs_track(n):
        truth, res = np.zeros(3), []
        for _ in range(n):
            truth += [math.cos(truth[2]) * 0.1, math.sin(truth[2]) * 0.1, 0.01]
            res.append([(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in lms4])
        return res

    obs = obs_track(1400)
    cfg = loc.ParticleFilterConfig(n_particles=1000, range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
    pf = loc.ParticleFilterLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=3)
    for t in range(200):
        pf.step([1.0, 0.1], obs[t])
    out["particle filter 1000 x 4: try_step (synchronous, returns the estimate)"] = timed(lambda t: pf.step([1.0, 0.1], obs[200 + t]), 1000)
    pf.set_resident(5000.0)
    for t in range(200):
        pf.step([1.0, 0.1], obs[t])
    out["particle filter 1000 x 4: try_step, resident"] = timed(lambda t: pf.step([1.0, 0.1], obs[200 + t]), 1000)
    small = bench.leg_small_n(True)  # the rows of the default bench line (ctypes with prebuilt pointers, CPU loop timed inside C)
    for key, row in small["rows"].items():
        for name, v in row.items():
            if isinstance(v, float):
                out[f"particle filter {key}: {name}"] = v
    # ---- MonteCarloLocalizer, default configuration (monte_carlo_localization.rs:66-80: 100 .. 5000 particles, KLD-adaptive)
    for lo, hi in ((100, 5000), (100, 1000)):
        mcl = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], loc.MonteCarloLocalizationConfig(min_particles=lo, max_particles=hi), seed=5)
        for t in range(200):
            mcl.step_async([1.0, 0.1], obs[t])
        mcl.synchronize()
        t0 = time.perf_counter()
        for t in range(200, 700):
            mcl.step_async([1.0, 0.1], obs[t])
        mcl.synchronize()
        out[f"adaptive MCL {lo}..{hi}: step_async"] = (time.perf_counter() - t0) / 500 * 1e6
        out[f"adaptive MCL {lo}..{hi}: try_step (synchronous, returns the estimate)"] = timed(lambda t: mcl.try_step([1.0, 0.1], obs[700 + t]), 500)
        mcl.set_resident(5000.0)
        for t in range(100):
            mcl.try_step([1.0, 0.1], obs[t])
        out[f"adaptive MCL {lo}..{hi}: try_step, resident"] = timed(lambda t: mcl.try_step([1.0, 0.1], obs[800 + t]), 500)
        out[f"adaptive MCL {lo}..{hi}: particles at the end"] = int(mcl.particle_count())
    # ---- FastSLAM 1.0, fastslam1.rs:237-274 / render_gif_slam.rs:172-178
    for n, L in ((100, 8), (1000, 8)):
        lms = np.random.default_rng(3).uniform(-13, 13, size=(L, 2))
        f = fs.FastSlam1(n, L, seed=5)
        zs = [np.ascontiguousarray(np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=5, step=t)).reshape(-1, 3)) for t in range(64)]
        for t in range(200):
            f.update_async([0.5, 0.1], zs[t % 64])
        f.synchronize()
        t0 = time.perf_counter()
        for t in range(1000):
            f.update_async([0.5, 0.1], zs[t % 64])
        f.synchronize()
        out[f"FastSLAM 1.0 {n} x {L}: update_async"] = (time.perf_counter() - t0) / 1000 * 1e6
        out[f"FastSLAM 1.0 {n} x {L}: update (synchronous)"] = timed(lambda t: f.update([0.5, 0.1], zs[t % 64]), 1000)

        def loop(t):
            f.update([0.5, 0.1], zs[t % 64])
            f.best_particle()

        out[f"FastSLAM 1.0 {n} x {L}: update + best_particle (the reference's loop)"] = timed(loop, 1000)

        def loop_async(t):
            f.update_async([0.5, 0.1], zs[t % 64])
            f.best_particle()

        out[f"FastSLAM 1.0 {n} x {L}: update_async + best_particle"] = timed(loop_async, 1000)
        f.set_resident(5000.0)
        for t in range(200):
            loop(t)
        out[f"FastSLAM 1.0 {n} x {L}: update + best_particle, resident"] = timed(loop, 1000)
        # the reference's loop on one core: fastslam_update + get_best_particle (noise samples pre-drawn; maps with the EKF branch live)
        ref = oracle.ref()
        ref.ref_set_threads(1)
        m = oracle.RefFs1Model()
        if m is not None:
            ref.ref_fs1_model_default(C.byref(m))
            rng = np.random.default_rng(1)
            px, py, pyaw, pw = (np.zeros(n) for _ in range(4))
            lm = np.zeros((n, L, 6))
            ref.ref_fs1_create(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm))
            lm[:, :, 0:2] = lms[None, :, :] + rng.normal(0, 0.5, (n, L, 2))
            lm[:, :, 2], lm[:, :, 5] = 0.5, 0.5
            idx = np.empty(n, np.uint32)
            z0s, z1s = rng.normal(size=(1200, n)), rng.normal(size=(1200, n))

            def cpu_loop(t):
                ref.ref_fs1_update(n, L, dp(px), dp(py), dp(pyaw), dp(pw), dp(lm), 0.5, 0.1, dp(z0s[t]), dp(z1s[t]), dp(zs[t % 64]), len(zs[t % 64]), C.byref(m), 100.0 / 1.5,
                                   0.37 / n, u32p(idx))
                ref.ref_fs1_best_particle(n, dp(pw))

            for t in range(100):
                cpu_loop(t)
            out[f"FastSLAM 1.0 {n} x {L}: cpu, the reference's loop (fastslam_update + get_best_particle), one core"] = timed(lambda t: cpu_loop(100 + t), 1000)
    # ---- the synchronous step of filters beyond the one-workgroup kernel
    for n, L in ((10_000, 4), (100_000, 32), (1_000_000, 32)):
        lm = H.landmarks_grid(L, 1)
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        p = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
        rng = np.random.default_rng(2)
        ob = [H.observations(lm, H.true_pose(t + 1), 0.2, rng) for t in range(64)]
        for t in range(300):
            p.step_async_estimate([1.0, 0.1], ob[t % 64])
        p.synchronize()
        out[f"MCL {n} x {L} systematic: step (synchronous, returns the estimate)"] = timed(lambda t: p.step([1.0, 0.1], ob[t % 64]), 500)
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
