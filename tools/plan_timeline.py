#!/usr/bin/env python3
"""In-kernel timeline of the one-launch resample plan (k_quantize_plan_mark) at the headline size, from wall-clock stamps the
instrumented build writes (make -C rust_robotics_amd/csrc timeline; 100 MHz clock, 10 ns resolution):

    RR_AMD_LIBRARY=rust_robotics_amd/librust_robotics_amd_timeline.so python tools/plan_timeline.py [out.json]

Stations per workgroup (thread 0): 0 start, 1 tile record stored (image of the tile done, acknowledged at device scope),
2 ticket taken, 3 the launch's state word seen (all sums out), 4 own prefix + totals read, 5 markers written, 6 estimate partial
stored.  Reported in microseconds after the FIRST workgroup's start: min / median / max over the workgroups, median over the
sampled launches."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from tests import helpers as H

    L_ = _ffi.lib()
    if not hasattr(L_, "rr_pf_debug_plan_timeline"):
        sys.exit("load the instrumented library: RR_AMD_LIBRARY=rust_robotics_amd/librust_robotics_amd_timeline.so")
    L_.rr_pf_debug_plan_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    n, L = 1_000_000, 32
    n_tiles, words = (n + 2047) // 2048, 8
    lms = H.landmarks_grid(L, 1)
    out = {"workload": f"fixed-N MCL {n} x {L}, systematic, resample every step", "n_tiles": n_tiles, "clock": "wall_clock64, 100 MHz",
           "stations": ["start", "record stored", "ticket taken", "state word seen", "sums read", "markers written", "estimate partial stored"]}
    for name, est in (("plain step (rr_pf_step_async)", False), ("estimate-producing step (rr_pf_step_async_estimate)", True)):
        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, resample_scheme=1)
        rng = np.random.default_rng(2)
        step = pf.step_async_estimate if est else pf.step_async
        for t in range(1500):
            step([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
        pf.synchronize()
        samples = []
        for t in range(40):
            step([1.0, 0.1], H.observations(lms, H.true_pose(1501 + t), 0.2, rng))
            buf = np.zeros(n_tiles * words, dtype=np.uint64)
            rc = L_.rr_pf_debug_plan_timeline(pf._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size)
            assert rc == 0
            st = buf.reshape(n_tiles, words).astype(np.int64)
            t0 = st[:, 0].min()
            samples.append((st[:, :7] - t0) / 100.0)  # us
        a = np.stack(samples)  # [launch][tile][station]
        n_st = 7 if est else 6
        rows = {}
        for k in range(n_st):
            v = a[:, :, k]
            rows[out["stations"][k]] = {"min_us": float(np.median(v.min(axis=1))), "median_us": float(np.median(np.median(v, axis=1))),
                                        "max_us": float(np.median(v.max(axis=1)))}
        out[name] = rows
        assert pf.plan_stats() == (0, True)
        del pf
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
