#!/usr/bin/env python3
"""How many distinct cache lines do the iid draws of one 512-slot tile of the multinomial step touch?  ("sort the
tile's 512 targets in LDS so that neighbouring targets share guide lines and 128-byte record lines" -- or show that tile-local
reuse does not exist at 10^6.)  CPU only, numpy: the bench's own scene and configuration (bench.py make_scene, MCL defaults of
monte_carlo_localization.rs:50-82), the reference's arithmetic in float64, multinomial resampling every step
(particle_filter.rs:441-473); at a chosen step the draws of every slot tile are looked at:

  records   the source records k_step_lazy<kSrcDraw, PACKED> reads: 32 B each, 4 per 128-byte line      -> distinct j // 4 per tile
  guide     the guide pairs mn_guide_search reads: 8 B each, 16 per line, bucket = target >> s        -> distinct bucket // 16 per tile
  (the CDF probes inside a bracket are left out: 1.5 us of 56 in the knock-out builds, profiles/r04_multinomial_ab.md)

A set of lines does not depend on the order its members are asked for in, so SORTING a tile's targets cannot lower these counts; it
could only help if several draws of a tile fell into the same line.  The table says how often they do.

    python tools/multinomial_line_reuse.py [n=1000000] [L=32] [steps=12]  ->  JSON on stdout"""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
TILE, REC_PER_LINE, PAIRS_PER_LINE, LINE = 512, 4, 16, 128
sigma, sv, sw, dt = 0.2, 2.0, math.radians(40.0), 0.1  # MonteCarloLocalizationConfig::default()
lms = H.landmarks_grid(L, 1)
rng_obs = np.random.default_rng(2)
rng = np.random.default_rng(3)
x, y, yaw, v = np.zeros(n), np.zeros(n), np.zeros(n), np.ones(n)
rows = []
for t in range(steps):
    obs = H.observations(lms, H.true_pose(t + 1), sigma, rng_obs)
    ua = 1.0 + rng.normal(0, 1, n) * sv
    uw = 0.1 + rng.normal(0, 1, n) * sw
    x, y, yaw, v = x + dt * np.cos(yaw) * ua, y + dt * np.sin(yaw) * ua, yaw + dt * uw, ua  # motion_model, particle_filter.rs:255-301
    ss = np.zeros(n)
    for d, lx, ly in obs:
        ss += (d - np.hypot(x - lx, y - ly)) ** 2
    w = np.exp(-(ss - ss.min()) / (2 * sigma * sigma))  # (the common factor cancels in the normalisation, :426-439)
    cdf = np.cumsum(w / w.sum())
    r = rng.random(n)  # one iid draw per output slot, in slot order (:455-470)
    j = np.minimum(np.searchsorted(cdf, r, side="left"), n - 1)
    n_eff = 1.0 / np.sum((w / w.sum()) ** 2)
    # guide table over the TARGET space: 2^g buckets, g = ceil(log2 n) (resample_core.hpp guide_shift): bucket of a draw = floor(r * 2^g)
    g = int(math.ceil(math.log2(n)))
    bucket = np.minimum((r * (1 << g)).astype(np.int64), (1 << g) - 1)
    n_tiles = (n + TILE - 1) // TILE
    pad = n_tiles * TILE - n
    jt = np.concatenate([j, np.full(pad, -1)]).reshape(n_tiles, TILE)
    bt = np.concatenate([bucket, np.full(pad, -1)]).reshape(n_tiles, TILE)

    def distinct_per_tile(a, per_line):
        q = np.sort(np.where(a >= 0, a // per_line, -1), axis=1)
        return ((q[:, 1:] != q[:, :-1]) & (q[:, 1:] >= 0)).sum(axis=1) + (q[:, 0] >= 0)

    rec_lines = distinct_per_tile(jt, REC_PER_LINE)
    rec_sources = distinct_per_tile(jt, 1)
    guide_lines = distinct_per_tile(bt, PAIRS_PER_LINE)
    slots = (jt >= 0).sum(axis=1)
    rows.append({
        "step": t + 1, "n_eff_over_n": round(float(n_eff / n), 4),
        "distinct_sources_over_all_slots": round(float(len(np.unique(j)) / n), 4),
        "per_tile_of_512_draws": {
            "distinct_source_records_mean": round(float(rec_sources.mean()), 2),
            "distinct_record_lines_mean": round(float(rec_lines.mean()), 2), "distinct_record_lines_min": int(rec_lines.min()),
            "distinct_guide_lines_mean": round(float(guide_lines.mean()), 2), "distinct_guide_lines_min": int(guide_lines.min()),
            "draws_that_share_a_record_line_with_another_draw_of_the_tile": round(float(1.0 - rec_lines.sum() / slots.sum()), 5),
            "draws_that_share_a_guide_line_with_another_draw_of_the_tile": round(float(1.0 - guide_lines.sum() / slots.sum()), 5)},
        "bytes_if_every_tile_fetches_its_lines_once_MB": round(float((rec_lines.sum() + guide_lines.sum()) * LINE / 1e6), 1),
        "bytes_if_every_distinct_line_of_the_step_is_fetched_once_MB": round(
            float((len(np.unique(j // REC_PER_LINE)) + len(np.unique(bucket // PAIRS_PER_LINE))) * LINE / 1e6), 1)})
    x, y, yaw, v = x[j], y[j], yaw[j], v[j]
print(json.dumps({"n": n, "L": L, "tile": TILE, "line_bytes": LINE, "algorithmic_bytes_of_the_two_reads_MB": round(n * (32 + 8) / 1e6, 1),
                  "measured_read_traffic_of_the_kernel_MB": {"value": 216, "source": "profiles/r04d_pmc_hbm_traffic.csv (FETCH_SIZE x2)"},
                  "steps": rows}, indent=1))
