#!/usr/bin/env python3
"""The systematic resample AT the ABI limit of 2^31 - 1 particles (tools/max_size_probe.py checks the estimate there; this checks the indices):
the recorded source indices of one resample are non-decreasing, inside [0, n), and over the first and the last 2^24 sources every offspring count is
within 1 of n w / sum w (the property of a systematic walk) -- 17 GB of weights and 8.6 GB of indices cross to the host for it.
    python tools/max_size_indices_probe.py [particles] > profiles/r06z12_resample_indices_at_the_abi_limit.json"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import rust_robotics_amd.localization as loc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2**31 - 1
cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=5, resample_scheme=1, record_indices=True)
rng = np.random.default_rng(6)
for t in range(2):
    pf.step_async([1.0, 0.1], H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.2, rng))
pf.predict_with_control([1.0, 0.1])
pf.update_with_observations(H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(3), 0.2, rng))
w = pf.raw_weights()            # 17 GB on the host
wsum = float(w.sum(dtype=np.float64))
pf.resample()
idx = pf.last_resample_indices()  # 8.6 GB
ok_sorted, ok_range, prev = True, True, 0
counts_err = 0.0
CH = 1 << 27
for a in range(0, n, CH):
    c = idx[a:a + CH].astype(np.int64)
    ok_sorted &= bool(c[0] >= prev and np.all(np.diff(c) >= 0)); prev = int(c[-1])
    ok_range &= bool(c.max() < n)
# systematic offspring within 1 of n w / sum(w), on the first and the last 2^24 sources (a full bincount would need 17 GB more)
for lo, hi in ((0, 1 << 24), (n - (1 << 24), n)):
    a, b = np.searchsorted(idx, [lo, hi])
    cnt = np.bincount(idx[a:b].astype(np.int64) - lo, minlength=hi - lo)
    counts_err = max(counts_err, float(np.max(np.abs(cnt - n * w[lo:hi] / wsum))))
print(json.dumps({"particles": n, "indices_sorted": ok_sorted, "indices_in_range": ok_range, "first": int(idx[0]), "last": int(idx[-1]),
                  "max_offspring_minus_n_w_over_both_ends": counts_err, "n_eff_ok": 0 < pf.n_eff() <= n}))
