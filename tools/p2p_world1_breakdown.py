#!/usr/bin/env python3
"""Where the one-rank peer-to-peer step's time goes against the plain step (10^6 x 32): host enqueue time per step (no
synchronisation inside), wall time per step, for both.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd.sharded import P2PShard
    from tests import helpers as H

    n, L, K = 1_000_000, 32, 200
    lms = H.landmarks_grid(L, 1)
    rng = np.random.default_rng(2)
    obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(1200 + 3 * K)]
    out = {}
    sh = P2PShard(0, 1, 0, n, seed=1)
    P2PShard.link_local([sh])
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
    for name, step, sync in (("p2p world 1", lambda o: sh.step([1.0, 0.1], o), sh.synchronize), ("plain", lambda o: pf.step_async([1.0, 0.1], o), pf.synchronize)):
        for t in range(1200):
            step(obs[t])
            if t % 50 == 49:
                sync()
        sync()
        rows = []
        for rep in range(3):
            t0 = time.perf_counter()
            for t in range(K):
                step(obs[1200 + rep * K + t])
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            rows.append({"enqueue_us_per_step": (t1 - t0) / K * 1e6, "wall_us_per_step": (t2 - t0) / K * 1e6})
        out[name] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
