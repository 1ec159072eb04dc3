#!/usr/bin/env python3
"""How far does one 288 GB MI355X take the unsharded MCL filter?  Particle sets from BASELINE configs[4]'s 1.6e7 up to the ABI's
limit (n_particles < 2^31: 32-bit resample indices and markers), L = 4 landmarks, a few steps each with the systematic resampler
every step, checked by what does not need an N-sized copy to the host: the estimate tracks the truth, N_eff lies in (0, N], the
covariance is finite and its diagonal non-negative, the counters count.  Prints one JSON line per size (stops at the first size
the device refuses; an allocation failure is an error message, not a crash).
    python tools/max_size_probe.py [max_particles [min_particles [scheme]]] > profiles/r06z5_max_size_probe.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402  (scene builders only)
import rust_robotics_amd.localization as loc  # noqa: E402


def main():
    top = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2**31 - 1
    low = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0
    scheme = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # 1: systematic (fastslam1.rs:205-234), 0: multinomial (monte_carlo_localization.rs:322-365)
    L, steps = 4, 4
    lms = H.REF_SCENE_LANDMARKS
    for n in (16_000_000, 100_000_000, 400_000_000, 1_000_000_000, 1_600_000_000, 2**31 - 1):
        if n > top:
            break
        if n < low:
            continue
        row = {"particles": n, "landmarks": L, "resample": "systematic" if scheme == 1 else "multinomial"}
        t0 = time.perf_counter()
        try:
            cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
            pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=5, resample_scheme=scheme)
        except Exception as e:  # the engine's message (hipMalloc, a limit of the ABI)
            row.update(ok=False, error=str(e)[:300])
            print(json.dumps(row), flush=True)
            break
        row["create_s"] = round(time.perf_counter() - t0, 3)
        rng = np.random.default_rng(6)
        try:
            for t in range(2):
                pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
            pf.synchronize()
            t1 = time.perf_counter()
            for t in range(2, 2 + steps):
                pf.step_async([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
            pf.synchronize()
            dt = (time.perf_counter() - t1) / steps
            est = pf.estimate()
            truth = H.true_pose(2 + steps)
            cov = pf.calc_covariance()
            pf.predict_with_control([1.0, 0.1])
            pf.update_with_observations(H.observations(lms, H.true_pose(3 + steps), 0.2, rng))
            neff = pf.n_eff()
            row.update(ok=bool(np.all(np.isfinite(est)) and np.hypot(est[0] - truth[0], est[1] - truth[1]) < 0.5 and 0.0 < neff <= n
                               and np.all(np.isfinite(cov)) and np.all(np.diag(cov) >= 0.0)),
                       ms_per_step=round(dt * 1e3, 3), updates_per_s=round(n * L / dt, 1), particle_steps_per_s=round(n / dt, 1),
                       whole_step_algorithmic_GBps=round(200.0 * n / dt / 1e9, 1), estimate=[round(float(v), 5) for v in est],
                       truth=[round(float(v), 5) for v in truth[:2]], n_eff=neff, counters=list(pf.counters()))
        except Exception as e:
            row.update(ok=False, error=str(e)[:300])
        print(json.dumps(row), flush=True)
        del pf
        if not row["ok"]:
            break


if __name__ == "__main__":
    main()
