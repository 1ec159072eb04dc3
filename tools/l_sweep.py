#!/usr/bin/env python3
"""Where does the MCL step kernel turn from HBM-bound to FP64-bound?  k_step_lazy (propagate + weight + folded resample gather)
at a size far beyond the Infinity Cache (default 1.6e7 particles: 1.15 GB per launch) for L = 1 .. 64 landmarks: kernel time from
the dispatch timestamps of its own launches, HBM fraction = 72 B/particle / time / 8 TB/s, FP64 fraction = (per-pair x L +
per-particle) lane-instructions / time / the issue peak (csrc/INSTRUCTION_BUDGET.json).  Prints one JSON object.
    python tools/l_sweep.py [particles] > profiles/r04_mcl_L_sweep.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rust_robotics_amd.localization as loc  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
    pair_i, part_i = bench.mcl_instruction_budget()
    rows = []
    for L in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64):
        K, W = 12, 6
        obs = bench.make_scene(L, W + K + 2, seed=1)
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
        u = [1.0, 0.1]
        for t in range(W):
            pf.step_async(u, obs[t])
        pf.synchronize()
        pf.profile_enable(2)  # dispatch timestamps of k_step_lazy only
        pf.profile_reset()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            pf.step_async(u, obs[t])
        pf.synchronize()
        dt = (time.perf_counter() - t0) / K
        cnt, ms = pf.profile_read()["k_propagate_weight"]
        pf.profile_enable(0)
        del pf
        k_s = ms / max(cnt, 1) * 1e-3
        hbm = 72.0 * n / k_s
        valu = (pair_i * L + part_i) * n / k_s
        rows.append({"landmarks": L, "step_ms": round(dt * 1e3, 4), "k_step_lazy_ms": round(k_s * 1e3, 4), "launches": int(cnt),
                     "hbm_GBps": round(hbm / 1e9, 1), "hbm_frac": round(hbm / bench.HBM_PEAK, 4),
                     "fp64_lane_instr_per_particle": pair_i * L + part_i, "fp64_frac": round(valu / bench.FP64_VALU_PEAK, 4),
                     "bound": "fp64_valu" if valu / bench.FP64_VALU_PEAK > hbm / bench.HBM_PEAK else "hbm",
                     "updates_per_s": round(n * L / dt, 1)})
    print(json.dumps({"particles": n, "kernel": "k_step_lazy<kSrcMarkers> (systematic, fused likelihood)",
                      "algorithmic_bytes_per_particle": 72, "hbm_peak_GBps": bench.HBM_PEAK / 1e9,
                      "fp64_issue_peak_lane_instr_per_s": bench.FP64_VALU_PEAK, "instruction_budget": {"per_pair": pair_i, "per_particle": part_i},
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
