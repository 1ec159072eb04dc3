#!/usr/bin/env python3
"""How much of k_step_lazy's idle issue time at 1e6 particles is wave quantisation?  A 1e6-particle launch is 15 625 rows of 64
slots on 1 024 SIMDs: 15.26 per SIMD on average, 16 on the fullest (1 954 workgroups of 512 slots on 256 CUs: 8 on some, 7 on the
others) -- the launch takes the time of 16 rows per SIMD whatever the kernel does, 4.6 % more than its work.  This sweep times the
kernel (dispatch timestamps of its own launches) at particle counts around whole numbers of workgroups per CU: 7 per CU
(917 504), 1e6, 8 per CU (1 048 576), 8 per CU + 1 workgroup, 9 per CU.  If the kernel's time is a staircase in N -- flat from
1e6 to 1 048 576 -- the missing 4.6 % at 1e6 is geometry, not the kernel.  L = 32, systematic, the headline's step.
    python tools/n_sweep.py > profiles/r06z_mcl_N_sweep.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rust_robotics_amd.localization as loc  # noqa: E402


def main():
    L = 32
    pair_i, part_i = bench.mcl_instruction_budget()
    tiles_per_cu = 256  # CUs
    sizes = [(7 * tiles_per_cu) * 512, 7 * tiles_per_cu * 512 + 512, 1_000_000, 8 * tiles_per_cu * 512, 8 * tiles_per_cu * 512 + 512,
             9 * tiles_per_cu * 512, 1_000_000]
    rows = []
    for n in sizes:
        K, W = 200, 300
        obs = bench.make_scene(L, 64, seed=1)
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
        u = [1.0, 0.1]
        for t in range(W):
            pf.step_async(u, obs[t % 64])
        pf.synchronize()
        pf.profile_enable(2)  # dispatch timestamps of k_step_lazy only
        pf.profile_reset()
        t0 = time.perf_counter()
        for t in range(K):
            pf.step_async(u, obs[t % 64])
        pf.synchronize()
        dt = (time.perf_counter() - t0) / K
        cnt, ms = pf.profile_read()["k_propagate_weight"]
        pf.profile_enable(0)
        del pf
        k_us = ms / max(cnt, 1) * 1e3
        workgroups = (n + 511) // 512
        rows.append({"particles": n, "workgroups": workgroups, "workgroups_per_cu": round(workgroups / 256.0, 3),
                     "rows_on_the_fullest_simd": 2 * ((workgroups + 255) // 256), "k_step_lazy_us": round(k_us, 3), "launches": int(cnt),
                     "step_us": round(dt * 1e6, 3), "ns_per_1000_particles": round(k_us * 1e3 / (n / 1000.0), 3),
                     "fp64_issue_frac": round((pair_i * L + part_i) * n / (k_us * 1e-6) / bench.FP64_VALU_PEAK, 4)})
    print(json.dumps({"kernel": "k_step_lazy<kSrcMarkers> (systematic, fused likelihood, L = 32, plain step)", "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
