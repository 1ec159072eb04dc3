#!/usr/bin/env python3
"""BASELINE.json configs[0]: the particle-filter localization demo (1 000 particles, the
reference's own 4-landmark scene of tests/unified_filter_comparison.rs:29-43,277-285) on the GPU
engine, driven exactly like examples/headless_localizers.rs:39-56 drives the CPU filter.

    python examples/particle_filter_localization.py [--particles N] [--steps K] [--seed S]
"""
import argparse
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rust_robotics_amd.core import ControlInput, Point2D  # noqa: E402
from rust_robotics_amd.localization import ParticleFilterConfig, ParticleFilterLocalizer  # noqa: E402

LANDMARKS = [(10.0, 0.0), (0.0, 15.0), (-5.0, 20.0), (10.0, 10.0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    cfg = ParticleFilterConfig(n_particles=args.particles, range_noise=0.5, velocity_noise=0.3,
                               yaw_rate_noise=math.radians(5.0), dt=0.1)
    pf = ParticleFilterLocalizer.try_new(cfg, seed=args.seed)
    pf.set_landmarks([Point2D(x, y) for x, y in LANDMARKS])
    rng = np.random.default_rng(args.seed)
    truth = np.zeros(3)
    u = ControlInput(1.0, 0.1)
    err = []
    for t in range(args.steps):
        truth[0] += u.v * math.cos(truth[2]) * cfg.dt
        truth[1] += u.v * math.sin(truth[2]) * cfg.dt
        truth[2] += u.omega * cfg.dt
        obs = [(max(0.0, math.hypot(lx - truth[0], ly - truth[1]) + rng.normal(0, 0.5)), lx, ly) for lx, ly in LANDMARKS]
        est = pf.try_step_state(u, obs)
        err.append(math.hypot(est.x - truth[0], est.y - truth[1]))
    print(f"PF  N={args.particles}  steps={args.steps}  final estimate=({est.x:.3f}, {est.y:.3f}, {est.yaw:.3f})  "
          f"truth=({truth[0]:.3f}, {truth[1]:.3f}, {truth[2]:.3f})  RMSE={math.sqrt(np.mean(np.square(err))):.3f} m")


if __name__ == "__main__":
    main()
