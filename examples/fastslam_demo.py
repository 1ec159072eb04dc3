#!/usr/bin/env python3
"""The FastSLAM call pattern of the reference's examples (crates/rust_robotics/examples/render_gif_slam.rs:166-205:
60 particles, its 6-landmark scene and control schedule, 72 steps) on the GPU engine, through the free functions that
keep the reference's signatures -- `create_particles`, `fastslam_update` / `fastslam2_update`, `get_best_particle`.

    python examples/fastslam_demo.py [--algorithm 1|2] [--particles N] [--steps K] [--seed S]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rust_robotics_amd.slam import fastslam1, fastslam2  # noqa: E402

LANDMARKS = [(2.5, 1.5), (6.0, 1.5), (9.5, 1.5), (2.5, 6.5), (6.0, 6.5), (9.5, 6.5)]  # render_gif_slam.rs:25-32
MAX_RANGE, R_DIST, R_ANGLE, DT = 18.0, 0.3, math.radians(5.0), 0.1


def control(step):  # render_gif_slam.rs:56-67
    phase = step % 36
    if phase < 14:
        return (1.0, 0.0)
    if phase < 20:
        return (0.35, 0.55)
    if phase < 28:
        return (0.9, 0.0)
    return (0.35, 0.55)


def wrap(a):
    return (a + math.pi) % (2 * math.pi) - math.pi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algorithm", type=int, choices=[1, 2], default=1)
    ap.add_argument("--particles", type=int, default=60)
    ap.add_argument("--steps", type=int, default=72)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    mod = fastslam1 if args.algorithm == 1 else fastslam2
    update = fastslam1.fastslam_update if args.algorithm == 1 else fastslam2.fastslam2_update
    rng = np.random.default_rng(args.seed)
    particles = mod.create_particles(args.particles, len(LANDMARKS))
    pose = np.zeros(3)
    err = []
    for step in range(args.steps):
        u = control(step)
        obs = []
        for k, (lx, ly) in enumerate(LANDMARKS):  # noisy_observations, render_gif_slam.rs:69-86
            dx, dy = lx - pose[0], ly - pose[1]
            d = math.hypot(dx, dy)
            if d <= MAX_RANGE:
                obs.append((d + rng.normal(0, R_DIST), wrap(math.atan2(dy, dx) - pose[2] + rng.normal(0, R_ANGLE)), k))
        update(particles, u, obs, seed=args.seed)
        pose = np.array([pose[0] + u[0] * DT * math.cos(pose[2]), pose[1] + u[0] * DT * math.sin(pose[2]), wrap(pose[2] + u[1] * DT)])
        best = mod.get_best_particle(particles)
        err.append(math.hypot(best.x - pose[0], best.y - pose[1]))
    seen = [lm for lm in best.landmarks if lm.cov[0, 0] < 100.0]
    lm_err = [math.hypot(lm.x - LANDMARKS[k][0], lm.y - LANDMARKS[k][1]) for k, lm in enumerate(best.landmarks) if lm.cov[0, 0] < 100.0]
    print(f"FastSLAM {args.algorithm}.0  N={args.particles}  steps={args.steps}  best pose=({best.x:.3f}, {best.y:.3f}, {best.yaw:.3f})  "
          f"truth=({pose[0]:.3f}, {pose[1]:.3f}, {pose[2]:.3f})  pose RMSE={math.sqrt(np.mean(np.square(err))):.3f} m  "
          f"landmarks mapped={len(seen)}/{len(LANDMARKS)}  mean landmark error={np.mean(lm_err) if lm_err else float('nan'):.3f} m")


if __name__ == "__main__":
    main()
