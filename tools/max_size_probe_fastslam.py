#!/usr/bin/env python3
"""How large a FastSLAM 1.0 filter does one 288 GB MI355X hold?  200 landmarks per particle (BASELINE's map size: 9.6 KB of maps per
particle and buffer set, two sets), particle counts from configs[3]'s 10^6 upwards, every landmark observed in every update, checked
by what needs no map-sized copy to the host: the best particle is finite and near the truth, its map lies around the landmarks, N_eff
in (0, N].  One JSON line per size; stops at the first size the device refuses (an error message, not a crash).
    python tools/max_size_probe_fastslam.py [max_particles [min_particles [1 | 2]]] > profiles/r06z5_max_size_probe_fastslam.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from rust_robotics_amd.slam import fastslam1 as fs  # noqa: E402
from rust_robotics_amd.slam import fastslam2 as fs2  # noqa: E402


def main():
    top = int(float(sys.argv[1])) if len(sys.argv) > 1 else 14_000_000
    low = int(float(sys.argv[2])) if len(sys.argv) > 2 else 0
    variant = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # 2: FastSLAM 2.0 (fastslam2.rs:173-374) on the same planes
    L, steps = 200, 3
    lms = np.random.default_rng(61).uniform(-13.0, 13.0, size=(L, 2))
    for n in (1_000_000, 4_000_000, 8_000_000, 10_000_000, 12_000_000, 14_000_000):
        if n > top:
            break
        if n < low:
            continue
        row = {"filter": f"FastSLAM {variant}.0", "particles": n, "landmarks": L, "map_bytes_both_sets_GB": round(2 * (3 + 6 * L) * 8 * n / 1e9, 1)}
        prm = fs2.default_params() if variant == 2 else fs.default_params()
        base = prm.base if variant == 2 else prm
        base.first_obs_cov = 0.5
        base.nth = n / 1.5
        t0 = time.perf_counter()
        try:
            f = (fs2.FastSlam2 if variant == 2 else fs.FastSlam1)(n, L, params=prm, seed=8)
        except Exception as e:
            row.update(ok=False, error=str(e)[:300])
            print(json.dumps(row), flush=True)
            break
        row["create_s"] = round(time.perf_counter() - t0, 3)
        try:
            z = [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=8, step=t)).reshape(-1, 3) for t in range(steps + 1)]
            f.update([0.5, 0.1], z[0])  # the first update initialises the maps
            t1 = time.perf_counter()
            for t in range(1, steps + 1):
                f.update_async([0.5, 0.1], z[t])
            f.synchronize()
            dt = (time.perf_counter() - t1) / steps
            pose, w, i = f.best_particle()
            lm_best = f.landmarks_of(i)
            truth = H.true_pose(steps + 1, v=0.5)
            neff = f.n_eff()
            row.update(ok=bool(np.all(np.isfinite(pose)) and np.hypot(pose[0] - truth[0], pose[1] - truth[1]) < 1.0 and np.all(np.isfinite(lm_best))
                               and np.median(np.hypot(lm_best[:, 0] - lms[:, 0], lm_best[:, 1] - lms[:, 1])) < 1.5 and 0.0 < neff <= n),
                       ms_per_update=round(dt * 1e3, 3), updates_per_s=round(n * L / dt, 1), algorithmic_GBps=round(96.0 * n * L / dt / 1e9, 1),
                       best_particle=int(i), best_pose=[round(float(v), 4) for v in pose], truth=[round(float(v), 4) for v in truth], n_eff=neff)
        except Exception as e:
            row.update(ok=False, error=str(e)[:300])
        print(json.dumps(row), flush=True)
        try:
            f.close()
        except Exception:
            pass
        del f
        if not row["ok"]:
            break


if __name__ == "__main__":
    main()
