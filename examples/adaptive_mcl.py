#!/usr/bin/env python3
"""KLD-adaptive Monte Carlo localization on the GPU engine, driven like the reference's own test
monte_carlo_localization.rs:489-516 (250..1200 particles, 3 landmarks, 60 steps): the particle count follows the
KLD bound over the occupied 0.5 m x 0.5 m x 15 deg bins.

    python examples/adaptive_mcl.py [--min N] [--max N] [--steps K] [--seed S]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rust_robotics_amd.localization import MonteCarloLocalizationConfig, MonteCarloLocalizer  # noqa: E402

LANDMARKS = [(0.0, 0.0), (10.0, 0.0), (5.0, 8.0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=250)
    ap.add_argument("--max", type=int, default=1200)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--seed", type=int, default=7)
    args = ap.parse_args()
    cfg = MonteCarloLocalizationConfig(min_particles=args.min, max_particles=args.max, range_noise=0.25, velocity_noise=0.05,
                                       yaw_rate_noise=0.02, dt=0.1)
    mcl = MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 0.0], cfg, seed=args.seed)
    truth = [0.0, 0.0, 0.0]
    counts = []
    for _ in range(args.steps):
        truth = [truth[0] + math.cos(truth[2]) * 0.1, truth[1] + math.sin(truth[2]) * 0.1, truth[2] + 0.03 * 0.1]
        obs = [(math.hypot(truth[0] - lx, truth[1] - ly), lx, ly) for lx, ly in LANDMARKS]
        est = mcl.try_step([1.0, 0.03], obs)
        counts.append(mcl.particle_count())
    print(f"adaptive MCL  particles {args.min}..{args.max}: count per step min={min(counts)} max={max(counts)} last={counts[-1]}  "
          f"estimate=({est[0]:.3f}, {est[1]:.3f})  truth=({truth[0]:.3f}, {truth[1]:.3f})  "
          f"error={math.hypot(est[0] - truth[0], est[1] - truth[1]):.3f} m")


if __name__ == "__main__":
    main()
